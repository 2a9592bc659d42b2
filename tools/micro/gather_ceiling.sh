#!/bin/bash
# tools/micro/gather_ceiling.hip on the GPU box: timings, then counter passes of the same command in --pmc mode (one launch per
# configuration), joined by tools/micro/gather_ceiling_report.py into gpurun_out/gather_ceiling.json.
#   gpurun --timeout 900 -- 'bash tools/micro/gather_ceiling.sh'
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
BIN=$OUT/gather_ceiling
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/micro/gather_ceiling.hip -o $BIN || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 $BIN ${GC_ARGS:-} > $OUT/gather_ceiling_timing.jsonl 2> $OUT/gather_ceiling_timing.err || { echo "timing run failed"; tail -5 $OUT/gather_ceiling_timing.err; exit 1; }
echo "timing done: $(wc -l < $OUT/gather_ceiling_timing.jsonl) configurations"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/gc_pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/gc_pmc_$tag -- $BIN --pmc > $OUT/gc_pmc_$tag.jsonl 2> $OUT/gc_pmc_$tag.err || echo "pass failed: $pass"
  echo "pass done: $pass"
done
python3 $ROOT/tools/micro/gather_ceiling_report.py $OUT > $OUT/gather_ceiling.json && tail -c 3000 $OUT/gather_ceiling_report.txt
