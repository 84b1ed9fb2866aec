#!/bin/bash
set -e
O=gpurun_out/r03n
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -40 $O/gputest.log >&2; exit 1; }
tail -2 $O/gputest.log >&2
for wl in cfg3_headline cfg3_clustered cfg2_clustered cfg2_truck7k; do
  python bench.py --no-cpu-baseline --workload $wl > $O/bench_$wl.json 2>> $O/err.log
  GS_BWD_SEGMENTS=0 python bench.py --no-cpu-baseline --workload $wl > $O/bench_${wl}_nosegments.json 2>> $O/err.log
done
for wl in cfg3_clustered cfg2_clustered; do python tools/bwd_wave_timeline.py $wl > $O/bwd_wave_timeline_$wl.txt 2>> $O/err.log; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r03n/bench_*.json")):
    d=json.load(open(f)); k=d["kernels_ms_per_view"]; print(f.split("/")[-1], d["value"], k["k_blend_bwd_tile"], k["k_blend_fwd"], k["k_sum_rows"], k["k_tile_order"])
PY
