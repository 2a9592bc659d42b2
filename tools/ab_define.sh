#!/bin/bash
# A/B of a compile-time switch on the GPU box: rebuilds kernels.o with each -D set given and runs one `bench.py --only X`.
#   tools/ab_define.sh ext "-DCAP_TS_EXT=5" "-DCAP_TS_EXT=7"     (the product build is run first as the baseline)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
what=$1; shift
# whatever happens, the checkout is left with the PRODUCT build (a later bench.py or pytest would measure the last -D set)
trap '(cd $ROOT/capsaicin_amd/csrc && make -B kernels.o trace8.o bvh.o context.o && make) > /dev/null 2>&1' EXIT
run() { python bench.py --only $what --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['result']; print('  value %.0f  ms %.2f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in (d.get('stage_ms') or d.get('batch_64spp', {}).get('stage_ms', {})).items()}))"; }
echo "baseline"; run; run
for def in "$@"; do
    (cd capsaicin_amd/csrc && make -B kernels.o trace8.o bvh.o context.o EXTRA="$def" > /dev/null 2>&1 && make EXTRA="$def" > /dev/null 2>&1) || { echo "build failed: $def"; continue; }
    echo "$def"; run; run
done
