#!/bin/bash
# A/B sweep of the wide closest-hit kernel's run-time knobs on the 262 k-triangle scene; run through gpurun.
for cfg in "16 12" "16 8" "16 20" "8 12" "24 12" "8 8" "16 32" "16 64" "$@"; do
  set -- $cfg
  echo -n "refill=$1 tri_batch=$2 : "
  CAP_W8_REFILL=$1 CAP_W8_TRI_BATCH=$2 bash tools/sponza_stages.sh
done
echo -n "TOP: "; CAP_W8_TOP=1 bash tools/sponza_stages.sh
