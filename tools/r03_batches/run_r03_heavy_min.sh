#!/bin/bash
# absolute floor of the heavy-tile threshold (work units): 1024 (default), 512, 256
set -e
O=gpurun_out/r03_hm; mkdir -p $O
for r in 1 2; do for v in hm1024 hm512 hm256; do for wl in cfg2_clustered cfg3_clustered cfg2_truck7k cfg3_headline; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_hm/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "bwd", k.get("k_blend_bwd_tile"), "sum_rows", k.get("k_sum_rows"))
PY
