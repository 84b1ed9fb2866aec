#!/bin/bash
set -e
O=gpurun_out/r03c
mkdir -p $O
for wl in cfg3_headline cfg3_clustered cfg2_clustered cfg2_truck7k; do
  python tools/bwd_wave_timeline.py $wl > $O/bwd_timeline_$wl.txt 2>> $O/err.log
  python tools/fwd_wave_timeline.py $wl > $O/fwd_timeline_$wl.txt 2>> $O/err.log
done
echo "timelines done" >&2
GSRAST_LIB=$PWD/build_ab/libgsrast_eager.so python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or random_small or heavy_tile or large_and_degenerate or reproducible" > $O/eager_parity.log 2>&1
tail -2 $O/eager_parity.log >&2
bash tools/ab_libs.sh r03c_ab cfg3_headline base eager > $O/ab_sum_rows.txt 2>> $O/err.log
cat $O/ab_sum_rows.txt >&2
