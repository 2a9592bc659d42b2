#!/bin/bash
# A/B of compile-time switches of the traversal kernels (trace8.hip / cap_wide_trace.h users) on the GPU box:
#   tools/ab_trace8.sh "-DCAP_W8_PREFETCH" "-DCAP_W8_SETPRIO=1"
# per set: rebuild, one small parity test, then tools/hall_stages.py on the 16.8 M (8 spp) and the 262 k (32 spp) hall.  The product
# build is measured first and restored at exit.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OBJS=${AB_OBJS:-"trace8.o"}
trap '(cd $ROOT/capsaicin_amd/csrc && make -B $OBJS && make) > /dev/null 2>&1' EXIT
run() {
    timeout -k 10 300 python -m pytest "tests/test_sponza_class_gpu.py::test_small_scale_parity" -x -q -m gpu 2>&1 | tail -1
    for sc in ${AB_SCALES:-8 1}; do timeout -k 10 300 python tools/hall_stages.py $sc 2>&1 | grep -v amdgpu.ids | cut -c1-420; done
}
echo "== product"; run
for def in "$@"; do
    (cd capsaicin_amd/csrc && make -B $OBJS EXTRA="$def" > /dev/null 2>&1 && make EXTRA="$def" > /dev/null 2>&1) || { echo "build failed: $def"; continue; }
    echo "== $def"; run
done
