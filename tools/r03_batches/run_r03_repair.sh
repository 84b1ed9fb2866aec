#!/bin/bash
# the cut policy (every list over 512 entries) with the segments' self-check and the repair launch, against the state before (r03_e sources)
set -e
O=gpurun_out/r03_repair; mkdir -p $O
for r in 1 2; do for v in base new; do for wl in cfg2_clustered cfg2_truck7k cfg3_headline cfg3_clustered; do
  GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_${v}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_repair/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "fwd", k.get("k_blend_fwd"), "bwd", k.get("k_blend_bwd_tile"), "repair", k.get("k_blend_bwd_repair"), "order", k.get("k_tile_order"))
PY
