#!/bin/bash
set -e
O=gpurun_out/r03o
mkdir -p $O
for wl in cfg2_clustered cfg3_clustered cfg3_headline cfg2_truck7k; do
  bash tools/ab_libs.sh r03o_$wl $wl wpb0 wpb4 wpb8 wpb16 > $O/ab_$wl.txt 2>> $O/err.log
  echo "== $wl" >&2; cat $O/ab_$wl.txt >&2
done
