#!/bin/bash
# heavy-tile threshold (in half-means of tile work) once every list longer than one segment is cut: 4 (default), 3, 2
set -e
O=gpurun_out/r03_heavyx2; mkdir -p $O
for r in 1 2; do for x in 4 3 2; do for wl in cfg2_clustered cfg3_clustered cfg2_truck7k cfg3_headline; do
  GS_BWD_HEAVY_X2=$x python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > $O/${wl}_x${x}_$r.json 2>/dev/null
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_heavyx2/*.json")):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], "fwd", k.get("k_blend_fwd"), "bwd", k.get("k_blend_bwd_tile"))
PY
