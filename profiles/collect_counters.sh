#!/bin/bash
# Arbitrary SQ/TCP/... counters for every k_* kernel of the bench, one rocprofv3 --pmc pass per call (counters in their
# own run, --kernel-trace only).  Run on the GPU box from the repo root:
#   bash profiles/collect_counters.sh <out.json> "<COUNTER> <COUNTER> ..." [workload]
set -e
DST=$1
CTRS=$2
WL=${3:-cfg3_headline}
export TMPDIR=/tmp
OUT=gpurun_out/ctr_$$
rm -rf $OUT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 bench.py --workload $WL --steps 4 --warmup 2 --breakdown-steps 1 --no-cpu-baseline > $OUT.log 2>&1
python3 - "$OUT" "$DST" "$WL" <<'PY'
import collections, csv, glob, json, os, sys
out, dst, wl = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(x) / len(x), 1) for c, x in v.items()} for k, v in acc.items()}
old = json.load(open(dst)) if os.path.exists(dst) else {"workload": wl, "note": "per-launch averages over the launches of bench.py --steps 4 --warmup 2", "kernels": {}}
for k, v in res.items():
    old["kernels"].setdefault(k, {}).update(v)
json.dump(old, open(dst, "w"), indent=1, sort_keys=True)
for k in ("k_blend_bwd_tile", "k_blend_fwd"):
    if k in res: print(k, res[k])
PY
rm -rf $OUT $OUT.log
