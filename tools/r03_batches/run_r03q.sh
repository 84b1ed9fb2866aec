#!/bin/bash
O=gpurun_out/r03q
mkdir -p $O
for wl in cfg2_clustered cfg2_truck7k; do
 for lib in wpb0 wpb8; do
  for x2 in 2 3 4 6; do
    GS_BWD_HEAVY_X2=$x2 GSRAST_LIB=$PWD/build_ab/libgsrast_$lib.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${lib}_x$x2.json 2>> $O/err.log
  done
 done
done
for wl in cfg3_clustered cfg3_headline; do
 for lib in wpb0 wpb8; do
  for x2 in 4 6; do
    GS_BWD_HEAVY_X2=$x2 GSRAST_LIB=$PWD/build_ab/libgsrast_$lib.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${lib}_x$x2.json 2>> $O/err.log
  done
 done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03q/*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], k["k_blend_bwd_tile"], k["k_blend_fwd"], k["k_sum_rows"])
PY
for wl in cfg2_clustered cfg3_clustered; do python tools/bwd_wave_timeline.py $wl > $O/tl_$wl.txt 2>>$O/err.log; tail -1 $O/tl_$wl.txt; done
