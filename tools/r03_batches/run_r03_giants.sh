#!/bin/bash
# splats that cover thousands of tiles: k_keygen walks a block's pairs with all four waves, k_sum_rows keeps four rows per lane in flight
set -e
O=gpurun_out/r03_giants; mkdir -p $O
GSRAST_LIB=$PWD/build_ab/libgsrast_giants.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -k "splats or clustered or cfg2 or predicted or stage" > $O/parity.txt 2>&1 || { tail -20 $O/parity.txt; exit 1; }
tail -1 $O/parity.txt
for r in 1 2; do for v in base giants; do for wl in cfg3_headline cfg3_clustered cfg2_clustered; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_giants/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "keygen", k.get("k_keygen"), "sum_rows", k.get("k_sum_rows"), "bwd_points", k.get("k_bwd_points"))
PY
