#!/bin/bash
O=gpurun_out/r03r
mkdir -p $O
for wl in cfg3_clustered cfg3_headline; do
  for x2 in 2 3 4 5; do
    GS_BWD_HEAVY_X2=$x2 GSRAST_LIB=$PWD/build_ab/libgsrast_wpb8.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_wpb8_x$x2.json 2>> $O/err.log
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03r/*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], k["k_blend_bwd_tile"], k["k_blend_fwd"], k["k_sum_rows"])
PY
GS_BWD_HEAVY_X2=3 python tools/bwd_wave_timeline.py cfg3_clustered > $O/tl_cfg3_clustered_x3.txt 2>>$O/err.log; head -3 $O/tl_cfg3_clustered_x3.txt; tail -1 $O/tl_cfg3_clustered_x3.txt
