#!/bin/bash
# SQ counters of every kernel of the bench (one rocprofv3 PMC pass, --kernel-trace only), summarised per kernel into
# profiles/<tag>_sq_counters.json.  Run on the GPU box from the repo root:  bash profiles/collect_sq.sh r01_e [workload]
# SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles per wave (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -e
TAG=${1:-r01}
WL=${2:-cfg3_headline}
export TMPDIR=/tmp
OUT=gpurun_out/sq_$WL
rm -rf $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT -- python3 bench.py --workload $WL --steps 4 --warmup 2 --breakdown-steps 1 --no-cpu-baseline > $OUT.log 2>&1
python3 - "$OUT" "profiles/${TAG}_sq_counters.json" "$WL" <<'PY'
import collections, csv, glob, json, os, sys
out, dst, wl = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    res[k] = {c: round(x, 1) for c, x in m.items()}
    res[k]["valu_active_over_wave_cycles"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
    res[k]["wait_any_over_wave_cycles"] = round(m.get("SQ_WAIT_ANY", 0.0) / wc, 4)
    res[k]["issue_stall_over_wave_cycles"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
    idx = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    res[k]["lds_bank_conflict_fraction"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx, 4) if idx else None
sys.path.insert(0, os.getcwd())
from taichi_3d_gaussian_splatting_amd import _native
doc = {"workload": wl, "source_digest": _native.source_digest(), "note": "per-launch averages; ratios are per wave (quad-cycle units cancel)", "kernels": res}
json.dump(doc, open(dst, "w"), indent=1, sort_keys=True)
json.dump(doc, open(os.path.join(os.path.dirname(dst), "sq_counters.json"), "w"), indent=1, sort_keys=True)   # the copy bench.py reads
for k in ("k_blend_bwd_tile", "k_blend_fwd", "k_sort_scatter", "k_bwd_points"):
    if k in res:
        print(k, {c: res[k][c] for c in res[k] if c.endswith("cycles") or c.endswith("fraction")})
PY
rm -rf $OUT
