#!/bin/bash
# the loss kernels alone (both layouts) and the whole trainer iteration under rocprofv3, plus the tool's own numbers
set -e
export TMPDIR=/tmp
O=gpurun_out/r03_loss; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_trainer_step.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python3 tools/bench_trainer_step.py > $O/trainer_step.json
cat $O/trainer_step.json
for m in only-loss-chw only-loss-hwc only-fused-iteration; do
  rm -rf gpurun_out/prof_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python3 tools/bench_trainer_step.py --$m > $O/$m.log 2>&1
  cp $(ls gpurun_out/prof_$m/*/*_kernel_stats.csv | head -1) $O/${m}_kernel_stats.csv
done
python3 - <<'PY'
import csv
for m in ("only-loss-chw","only-loss-hwc","only-fused-iteration"):
    rows=list(csv.DictReader(open("gpurun_out/r03_loss/%s_kernel_stats.csv" % m)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(m, "GPU busy per call us:", round(tot/120/1e3,1))
    for r in rows[:12 if m!="only-fused-iteration" else 30]:
        print("   %-64s calls %5s avg %8.1f us" % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3))
PY
