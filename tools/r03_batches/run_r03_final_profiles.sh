#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the final build beyond the headline command: the loss kernels alone (both image
# layouts), a whole training iteration, and the two clustered workloads
set -e
export TMPDIR=/tmp
O=gpurun_out/r03_final; mkdir -p $O
for m in only-loss-chw only-loss-hwc only-fused-iteration; do
  rm -rf gpurun_out/prof_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python3 tools/bench_trainer_step.py --$m > $O/$m.log 2>&1
  cp $(ls gpurun_out/prof_$m/*/*_kernel_stats.csv | head -1) $O/r03_d_trainer_${m}_kernel_stats.csv
done
for wl in cfg3_clustered cfg2_clustered; do
  rm -rf gpurun_out/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$wl -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --workload $wl > $O/$wl.log 2>&1
  cp $(ls gpurun_out/prof_$wl/*/*_kernel_stats.csv | head -1) $O/r03_d_bench_${wl}_kernel_stats.csv
done
python3 - <<'PY'
import csv,glob
for f in sorted(glob.glob("gpurun_out/r03_final/*_kernel_stats.csv")):
    rows=list(csv.DictReader(open(f)))
    print(f.split("/")[-1], [(r['Name'].replace('void ','')[:18], round(float(r['AverageNs'])/1e3,1)) for r in rows[:4]])
PY
