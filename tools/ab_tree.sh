#!/bin/bash
# A/B of compile-time switches on the tree path (262 k-triangle scene) on the GPU box:
#   tools/ab_tree.sh "<objects to rebuild>" "<flags A>" "<flags B>" ...      e.g.  tools/ab_tree.sh "trace8.o" "" "-DCAP_W8_X=1"
# For every flag set: rebuilds the named objects with EXTRA=<flags>, runs the tree-path parity tests, then the stage split
# (tools/sponza_stages.sh) three times.  Leaves the LAST build behind: put the default last when it matters.
cd "$(dirname "$0")/.."
OBJS=$1
shift
for flags in "$@"; do
    echo "=== flags: '$flags'"
    (cd capsaicin_amd/csrc && make -B $OBJS EXTRA="$flags" > /dev/null 2>&1 && make > /dev/null 2>&1) || { echo "build failed"; continue; }
    timeout -k 10 300 python -m pytest tests/test_bvh_gpu.py tests/test_sponza_class_gpu.py -x -q -m gpu 2>&1 | tail -1
    for i in 1 2 3; do timeout -k 10 120 bash tools/sponza_stages.sh 2>&1 | tail -1; done
done
