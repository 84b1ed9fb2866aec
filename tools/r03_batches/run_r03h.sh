#!/bin/bash
set -e
O=gpurun_out/r03h
mkdir -p $O
bash tools/ab_libs.sh r03h_ab cfg3_headline base fwdmin lb6 > $O/ab_cfg3.txt 2>> $O/err.log
cat $O/ab_cfg3.txt >&2
GSRAST_LIB=$PWD/build_ab/libgsrast_fwdmin.so python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or random_small or large_and_degenerate or not_a_number or rgb_only" > $O/fwdmin_parity.log 2>&1 || { tail -20 $O/fwdmin_parity.log >&2; exit 1; }
tail -2 $O/fwdmin_parity.log >&2
