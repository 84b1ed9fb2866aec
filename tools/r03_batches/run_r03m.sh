#!/bin/bash
set -e
O=gpurun_out/r03m
mkdir -p $O
for wl in cfg3_headline cfg2_clustered cfg3_clustered cfg5_infer2e6; do
  bash tools/ab_libs.sh r03m_$wl $wl base semi fpf1 fpf2 > $O/ab_$wl.txt 2>> $O/err.log
  echo "== $wl" >&2; cat $O/ab_$wl.txt >&2
done
GSRAST_LIB=$PWD/build_ab/libgsrast_fpf1.so python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or random_small or heavy or edge_cases or partial_edge" > $O/fpf1_parity.log 2>&1 || { tail -20 $O/fpf1_parity.log >&2; exit 1; }
tail -1 $O/fpf1_parity.log >&2
