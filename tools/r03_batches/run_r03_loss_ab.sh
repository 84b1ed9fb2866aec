#!/bin/bash
# forward loss kernel: tile height 22 (24.5 KB of LDS, 6 blocks per CU) against 54 (47.5 KB, 3 blocks), both layouts, rocprofv3 averages
set -e
export TMPDIR=/tmp
O=gpurun_out/r03_loss_ab; mkdir -p $O
for v in mty22 mty54; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_trainer_step.py -x -q > $O/tests_$v.txt 2>&1 || { tail -30 $O/tests_$v.txt; exit 1; }
  for m in only-loss-chw only-loss-hwc; do
    rm -rf gpurun_out/prof_ab
    export GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 tools/bench_trainer_step.py --$m > $O/${v}_$m.log 2>&1
    cp $(ls gpurun_out/prof_ab/*/*_kernel_stats.csv | head -1) $O/${v}_${m}_kernel_stats.csv
  done
done
python3 - <<'PY'
import csv
for v in ("mty22","mty54"):
  for m in ("only-loss-chw","only-loss-hwc"):
    rows=list(csv.DictReader(open("gpurun_out/r03_loss_ab/%s_%s_kernel_stats.csv" % (v,m))))
    print(v, m, [(r['Name'][:24], round(float(r['AverageNs'])/1e3,1)) for r in rows[:2]])
PY
