#!/bin/bash
set -e
O=gpurun_out/r03k
mkdir -p $O
tools/ubench_scatter_runs.bin > $O/scatter_run_length.json 2>> $O/err.log
cat $O/scatter_run_length.json >&2
GSRAST_LIB=$PWD/build_ab/libgsrast_fused.so python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or random_small or heavy_tile_lists" > $O/fused_parity.log 2>&1 || { tail -20 $O/fused_parity.log >&2; exit 1; }
tail -1 $O/fused_parity.log >&2
bash tools/ab_libs.sh r03k_ab cfg3_headline base fused > $O/ab_fused.txt 2>> $O/err.log
cat $O/ab_fused.txt >&2
python - <<PY >&2
import json,glob
for f in sorted(glob.glob("gpurun_out/r03k_ab/*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], "sum_rows", k.get("k_sum_rows"), "bwd_points", k.get("k_bwd_points"))
PY
timeout -k 10 900 bash profiles/collect_round.sh r03_b > $O/collect.log 2>&1 || { tail -30 $O/collect.log >&2; exit 1; }
tail -8 $O/collect.log >&2
