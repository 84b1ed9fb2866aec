#!/bin/bash
# round-3 measurement batch (GPU box, repo root): bench lines in both sizing modes, fixed vs cycled pose, clustered workloads,
# host timeline, strict-vs-fast record.  Results under gpurun_out/r03b/.
set -e
O=gpurun_out/r03b
mkdir -p $O
python bench.py > $O/bench_cfg3.json 2> $O/err.log
echo "cfg3 done" >&2
GS_PREDICT_SIZES=0 python bench.py --no-cpu-baseline > $O/bench_cfg3_exact_sizing.json 2>> $O/err.log
python bench.py --no-cpu-baseline --fixed-pose > $O/bench_cfg3_fixed_pose.json 2>> $O/err.log
python bench.py --no-cpu-baseline > $O/bench_cfg3_again.json 2>> $O/err.log
echo "cfg3 variants done" >&2
python bench.py --no-cpu-baseline --workload cfg3_clustered > $O/bench_cfg3_clustered.json 2>> $O/err.log
python bench.py --no-cpu-baseline --workload cfg2_clustered > $O/bench_cfg2_clustered.json 2>> $O/err.log
python bench.py --no-cpu-baseline --workload cfg2_truck7k > $O/bench_cfg2.json 2>> $O/err.log
echo "workloads done" >&2
python tools/host_timeline.py > $O/host_timeline_predicted.txt 2>> $O/err.log
GS_PREDICT_SIZES=0 python tools/host_timeline.py > $O/host_timeline_exact.txt 2>> $O/err.log
echo "timeline done" >&2
python tools/strict_vs_fast.py $O/strict_vs_fast.json > $O/strict_vs_fast.log 2>> $O/err.log
echo "all done" >&2
