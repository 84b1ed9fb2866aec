#!/bin/bash
# segment length of cut lists / shortest list that is cut: 512/1024 (base), 512/512, 256/256, 256/512
set -e
O=gpurun_out/r03_seg; mkdir -p $O
for v in s512m512 s512m1024; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "heavy or cut or clustered or claim or splats" > $O/parity_$v.txt 2>&1 || { tail -20 $O/parity_$v.txt; exit 1; }
  tail -1 $O/parity_$v.txt
done
for r in 1 2; do for v in base s512m1024 s512m768 s512m512; do for wl in cfg2_clustered cfg2_truck7k; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_seg/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "fwd", k.get("k_blend_fwd"), "bwd", k.get("k_blend_bwd_tile"), "order", k.get("k_tile_order"), "sum_rows", k.get("k_sum_rows"))
PY
