#!/bin/bash
# A/B on one box: the lightest tiles handed out LAST as four-wave items (GS_BWD_TAIL = how many), against the build without.
set -e
O=gpurun_out/r03_tail; mkdir -p $O
L=$PWD/build_ab/libgsrast_tail.so
GSRAST_LIB=$L GS_BWD_TAIL=1024 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg3 or cfg2 or heavy or soak" > $O/parity.txt 2>&1 || { tail -20 $O/parity.txt; exit 1; }
tail -3 $O/parity.txt
for r in 1 2; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_base.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 > $O/cfg3_base_$r.json 2>/dev/null
  for n in 0 512 1024 2048 3072; do
    GSRAST_LIB=$L GS_BWD_TAIL=$n python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 > $O/cfg3_tail${n}_$r.json 2>/dev/null
  done
done
for wl in cfg3_clustered cfg2_truck7k cfg3_1080p; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_base.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_base_1.json 2>/dev/null
  for n in 1024 2048; do
    GSRAST_LIB=$L GS_BWD_TAIL=$n python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_tail${n}_1.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_tail/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], k.get("k_blend_fwd"), k.get("k_blend_bwd_tile"), k.get("k_tile_order"))
PY
