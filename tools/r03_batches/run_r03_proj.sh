#!/bin/bash
# k_project: feature rows loaded as 224-byte runs through LDS instead of one row per lane
set -e
O=gpurun_out/r03_proj; mkdir -p $O
GSRAST_LIB=$PWD/build_ab/libgsrast_proj.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or cfg2 or cfg3_headline or cfg5 or edge or random or counts" > $O/parity.txt 2>&1 || { tail -20 $O/parity.txt; exit 1; }
tail -1 $O/parity.txt
for r in 1 2; do for v in base proj; do for wl in cfg3_headline cfg5_infer2e6 cfg2_truck7k; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_proj/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "project", k.get("k_project"), "filter", k.get("k_filter"))
PY
