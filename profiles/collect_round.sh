#!/bin/bash
# Everything the judged numbers of a round come from, in one go on the GPU box (repo root):  bash profiles/collect_round.sh r03_b
# Results land in gpurun_out/round/ (the only directory that travels back from the box); copy them into profiles/ afterwards.
#   <tag>_bench_*.json            bench lines: cfg3 default (incl. cpu_baseline), cfg3 with exact sizing / a fixed pose, cfg2, cfg5 all
#                                 outputs and rgb_only, true 1080p, the two clustered workloads with and without heavy-tile sharing
#   <tag>_bench_cfg3_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the cfg3 command
#   pmc_traffic.json, sq_counters.json (+ <tag>_sq_counters.json)   counters, each in its own --pmc pass, labelled with the kernel sources' digest
#   <tag>_{bwd,fwd}_wave_timeline_<workload>.txt   wave start/end stamps from the timing-only build (make times)
#   <tag>_host_timeline.txt, <tag>_strict_vs_fast.json, <tag>_trainer_step.json
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
R=gpurun_out/round
mkdir -p $R
bash profiles/collect_pmc.sh cfg3_headline > gpurun_out/${TAG}_pmc.log 2>&1
bash profiles/collect_sq.sh ${TAG} cfg3_headline > gpurun_out/${TAG}_sq.log 2>&1
cp profiles/pmc_traffic.json profiles/sq_counters.json profiles/${TAG}_sq_counters.json $R/
echo "counters done" >&2
E=gpurun_out/${TAG}_bench.err
python3 bench.py > $R/${TAG}_bench_cfg3.json 2> $E
GS_PREDICT_SIZES=0 python3 bench.py --no-cpu-baseline > $R/${TAG}_bench_cfg3_exact_sizing.json 2>> $E
python3 bench.py --no-cpu-baseline --fixed-pose > $R/${TAG}_bench_cfg3_fixed_pose.json 2>> $E
python3 bench.py --workload cfg2_truck7k --no-cpu-baseline > $R/${TAG}_bench_cfg2.json 2>> $E
python3 bench.py --workload cfg5_infer2e6 --no-cpu-baseline > $R/${TAG}_bench_cfg5_inference.json 2>> $E
python3 bench.py --workload cfg5_infer2e6 --rgb-only --no-cpu-baseline > $R/${TAG}_bench_cfg5_inference_rgb_only.json 2>> $E
python3 bench.py --workload cfg3_1080p --no-cpu-baseline > $R/${TAG}_bench_cfg3_1080p.json 2>> $E
for wl in cfg3_clustered cfg2_clustered; do
  python3 bench.py --workload $wl --no-cpu-baseline > $R/${TAG}_bench_$wl.json 2>> $E
  GS_BWD_SPLIT_HEAVY=0 python3 bench.py --workload $wl --no-cpu-baseline > $R/${TAG}_bench_${wl}_no_heavy_sharing.json 2>> $E
  GS_BWD_SEGMENTS=0 python3 bench.py --workload $wl --no-cpu-baseline > $R/${TAG}_bench_${wl}_no_segments.json 2>> $E
  GS_BWD_SEGMENTS=1 python3 bench.py --workload $wl --no-cpu-baseline > $R/${TAG}_bench_${wl}_segments.json 2>> $E
done
echo "bench lines done" >&2
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1
cp $(ls gpurun_out/prof_${TAG}/*/*_kernel_stats.csv | head -1) $R/${TAG}_bench_cfg3_kernel_stats.csv
echo "rocprof done" >&2
for wl in cfg3_headline cfg3_clustered cfg2_clustered; do
  python3 tools/bwd_wave_timeline.py $wl > $R/${TAG}_bwd_wave_timeline_$wl.txt 2>> $E
  python3 tools/fwd_wave_timeline.py $wl > $R/${TAG}_fwd_wave_timeline_$wl.txt 2>> $E
done
python3 tools/blend_stats.py cfg3_headline > $R/${TAG}_blend_event_counters.json 2>> $E
python3 tools/host_timeline.py > $R/${TAG}_host_timeline.txt 2>> $E
GS_PREDICT_SIZES=0 python3 tools/host_timeline.py >> $R/${TAG}_host_timeline.txt 2>> $E
python3 tools/strict_vs_fast.py $R/${TAG}_strict_vs_fast.json > gpurun_out/${TAG}_strict.log 2>> $E
python3 tools/bench_trainer_step.py > $R/${TAG}_trainer_step.json 2>> $E || true
head -c 300 $R/${TAG}_bench_cfg3.json; echo; head -5 $R/${TAG}_bench_cfg3_kernel_stats.csv
