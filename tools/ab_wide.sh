#!/bin/bash
# Sweep of the wide closest-hit kernel's LDS stack part / register residency on the 262 k-triangle scene (through gpurun).
cd "$(dirname "$0")/.."
for cfg in "32 5 5" "24 5 5" "24 6 5" "24 6 6" "16 8 8" "16 6 6" "12 8 8"; do
    set -- $cfg
    HIPCC_COMPILE_FLAGS_APPEND="-DCAP_WIDE_LDS=$1 -DCAP_WIDE_BLOCKS=$2" make -C capsaicin_amd/csrc -B > /dev/null 2>&1
    echo "lds $1 bounds $2 grid $3: $(CAP_BLOCKS_PER_CU=$3 timeout -k 10 300 bash tools/sponza_stages.sh)"
done
make -C capsaicin_amd/csrc -B > /dev/null 2>&1
