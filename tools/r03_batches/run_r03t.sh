#!/bin/bash
O=gpurun_out/r03t
mkdir -p $O
for wl in cfg3_clustered cfg3_headline cfg2_clustered; do
 for seg in 0 1; do
  for x2 in 3 4 6; do
    GS_BWD_SEGMENTS=$seg GS_BWD_HEAVY_X2=$x2 python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_seg${seg}_x$x2.json 2>> $O/err.log
  done
 done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03t/*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], k["k_blend_bwd_tile"], k["k_blend_fwd"], k["k_sum_rows"])
PY
