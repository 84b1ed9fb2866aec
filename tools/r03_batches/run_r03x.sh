#!/bin/bash
O=gpurun_out/r03x
mkdir -p $O
rc=0
for env in "GS_BWD_SEGMENTS=0" "GS_BWD_SEGMENTS=1" "GS_PREDICT_SIZES=0" "GS_BWD_SPLIT_HEAVY=0" "GS_BWD_HEAVY_X2=3"; do
  tag=$(echo $env | tr '=' '_')
  env $env timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "heavy or soak_seeds or claim_order or partial_edge or point_counts or depth_only or cfg2" > $O/parity_$tag.log 2>&1 || rc=1
  echo "$env: $(tail -1 $O/parity_$tag.log)" >&2
done
exit $rc
