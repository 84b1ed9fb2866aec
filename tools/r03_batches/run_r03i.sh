#!/bin/bash
set -e
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -40 $O/gputest.log >&2; exit 1; }
tail -2 $O/gputest.log >&2
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_cfg3_$i.json 2>> $O/err.log
GS_PREDICT_SIZES=0 python bench.py --no-cpu-baseline > $O/bench_cfg3_exact_$i.json 2>> $O/err.log
done
python bench.py --no-cpu-baseline --workload cfg5_infer2e6 > $O/bench_cfg5.json 2>> $O/err.log
python bench.py --no-cpu-baseline --workload cfg2_truck7k > $O/bench_cfg2.json 2>> $O/err.log
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03i/bench_*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], d["config"]["forward_sizing"], {a:b for a,b in k.items()})
PY
