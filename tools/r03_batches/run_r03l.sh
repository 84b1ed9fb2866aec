#!/bin/bash
# the parity suite under the library's diagnostic switches: every path of the blend backward / sizing logic against the oracle
O=gpurun_out/r03l
mkdir -p $O
rc=0
for env in "GS_BWD_SEGMENTS=0" "GS_BWD_SEGMENTS=1" "GS_PREDICT_SIZES=0" "GS_BWD_SPLIT_HEAVY=0" "GS_BWD_WAVES_PER_TILE=1" "GS_BWD_WAVES_PER_TILE=2" "GS_BWD_WAVES_PER_TILE=4" "GS_FWD_ORDER_HINT=0" "GS_COUNTERS_WAIT=stream"; do
  tag=$(echo $env | tr '=' '_')
  env $env timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "not cfg5 and not predicted_sizing and not flag_tags and not cfg3_views and not true_1080p" > $O/parity_$tag.log 2>&1 || rc=1
  echo "$env: $(tail -1 $O/parity_$tag.log)" >&2
done

exit $rc
