#!/bin/bash
# k_sort_scatter with 4 / 8 / 16 waves per 4096-pair tile (same tile, same LDS staging): parity, then fps and the sort kernels' times
set -e
O=gpurun_out/r03_sort; mkdir -p $O
for v in sort512 sort1024; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg1 or cfg2 or cfg3 or cfg5 or 64 or heavy or stab" > $O/parity_$v.txt 2>&1 || { tail -20 $O/parity_$v.txt; exit 1; }
  tail -1 $O/parity_$v.txt
done
for r in 1 2; do for v in sort256 sort512 sort1024; do for wl in cfg3_headline cfg5_infer2e6 cfg2_truck7k; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_sort/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "scatter", k.get("k_sort_scatter"), "hist", k.get("k_sort_hist"), "keygen", k.get("k_keygen"))
PY
