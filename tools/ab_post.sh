#!/bin/bash
# A/B of compile-time switches of post.hip on the GPU box: tools/ab_post.sh "-DCAP_POST_DIAG=1" ...   (per-kernel times, fast mode)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
trap '(cd $ROOT/capsaicin_amd/csrc && make -B post.o && make) > /dev/null 2>&1' EXIT
run() { for m in ${AB_MODES:-fast}; do POST_MODE=$([ $m = fast ] && echo fast) bash tools/post_trace.sh 2>&1 | head -${AB_LINES:-10}; done; }
echo "== product"; run
for def in "$@"; do
    (cd capsaicin_amd/csrc && make -B post.o EXTRA="$def" > /dev/null 2>&1 && make EXTRA="$def" > /dev/null 2>&1) || { echo "build failed: $def"; continue; }
    echo "== $def"; run
done
