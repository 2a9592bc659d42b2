#!/bin/bash
# A/B of a compile-time switch of kernels.hip on the GPU box: tools/ab_flag.sh -DCAP_SOMETHING
# Runs the small-scene parity tests and tools/quick_stages.py with and without the flag; leaves the default build behind.
set -e
cd "$(dirname "$0")/.."
for flags in "$*" ""; do
    echo "=== flags: '$flags'"
    HIPCC_COMPILE_FLAGS_APPEND="$flags" make -C capsaicin_amd/csrc -B kernels.o > /dev/null 2>&1
    make -C capsaicin_amd/csrc > /dev/null 2>&1
    timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_golden_frames.py tests/test_post_gpu.py -x -q -m gpu 2>&1 | tail -1
    timeout -k 10 200 python tools/quick_stages.py 2>&1 | tail -1
    timeout -k 10 200 python tools/quick_stages.py 2>&1 | tail -1
done
