#!/bin/bash
# Everything the judged numbers of a round come from, in one go on the GPU box (repo root):  bash profiles/collect_round.sh r02_a
# Results land in gpurun_out/round/ (the only directory that travels back from the box); copy them into profiles/ afterwards.
#   <tag>_bench_*.json            bench lines (cfg3 default incl. cpu_baseline; cfg2; cfg5 all outputs and rgb_only; true 1080p)
#   <tag>_bench_cfg3_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the cfg3 command
#   pmc_traffic.json, sq_counters.json (+ <tag>_sq_counters.json)   counters, each in its own --pmc pass
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out/round
bash profiles/collect_pmc.sh cfg3_headline > gpurun_out/${TAG}_pmc.log 2>&1
bash profiles/collect_sq.sh ${TAG} cfg3_headline > gpurun_out/${TAG}_sq.log 2>&1
cp profiles/pmc_traffic.json profiles/sq_counters.json profiles/${TAG}_sq_counters.json gpurun_out/round/
python3 bench.py > gpurun_out/round/${TAG}_bench_cfg3.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --workload cfg2_truck7k --no-cpu-baseline > gpurun_out/round/${TAG}_bench_cfg2.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --workload cfg5_infer2e6 --no-cpu-baseline > gpurun_out/round/${TAG}_bench_cfg5_inference.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --workload cfg5_infer2e6 --rgb-only --no-cpu-baseline > gpurun_out/round/${TAG}_bench_cfg5_inference_rgb_only.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --workload cfg3_1080p --no-cpu-baseline > gpurun_out/round/${TAG}_bench_cfg3_1080p.json 2>> gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1
cp $(ls gpurun_out/prof_${TAG}/*/*_kernel_stats.csv | head -1) gpurun_out/round/${TAG}_bench_cfg3_kernel_stats.csv
python3 tools/bench_trainer_step.py > gpurun_out/round/${TAG}_trainer_step.json 2>> gpurun_out/${TAG}_bench.err || true
head -c 300 gpurun_out/round/${TAG}_bench_cfg3.json; echo; head -5 gpurun_out/round/${TAG}_bench_cfg3_kernel_stats.csv
