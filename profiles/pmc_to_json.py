"""Turn the two rocprofv3 PMC passes of collect_pmc.sh into profiles/pmc_traffic.json:
{workload: {kernel: corrected HBM bytes per launch}, workload+"_raw": {kernel: {FETCH_SIZE_KiB, WRITE_SIZE_KiB}}}."""
import collections
import csv
import glob
import json
import os
import sys

wl, out = sys.argv[1], sys.argv[2]
raw = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(os.path.join(out, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            raw[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res, rawout = {}, {}
for k, v in raw.items():
    if not k.startswith("k_"):
        continue
    fetch = sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1)
    write = sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1)
    rawout[k] = {"FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1)}
    res[k] = int((2.0 * fetch + write) * 1024)          # gfx950 correction: reads x2
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import _native  # noqa: E402
digest = _native.source_digest()
if data.get("_source_digest") != digest:        # counters of another build of the kernels are dropped, not mixed in
    data = {}
data["_source_digest"] = digest
data[wl] = res
data[wl + "_raw"] = rawout
data["_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE is doubled per the gfx950 note in "
                 "MI355X_MICROARCH.md (exact for wide coalesced reads, uncalibrated for gathers); averages over the launches of "
                 "`bench.py --steps 4 --warmup 2 --breakdown-steps 1`")
json.dump(data, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1))
