#!/bin/bash
# backward blend: the clean / not-clean choice hoisted out of the per-splat loop (two loops), with and without forcing five waves per SIMD
set -e
O=gpurun_out/r03_hoist; mkdir -p $O
GSRAST_LIB=$PWD/build_ab/libgsrast_hoist.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg2 or cfg3 or not_a_number or heavy" > $O/parity.txt 2>&1 || { tail -20 $O/parity.txt; exit 1; }
tail -1 $O/parity.txt
bash tools/ab_libs.sh r03_hoist cfg3_headline base hoist hoist5
