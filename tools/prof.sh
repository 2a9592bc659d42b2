#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun): kernel trace + stats, then PMC passes (separate runs; never with
# --sys-trace) over the bench command itself (headline + EXT + tree-path variants).  Outputs under gpurun_out/prof_*;
# tools/make_traffic.py and tools/prof_summary.py condense them into profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
# --no-extras: without the shard_cost / post_chain sections, whose launches of the same kernels on other amounts of work would
# pollute the per-launch averages of the three kernels bench.py prices
ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_EXTRA}"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_kt $OUT/prof_pmc_* $OUT/prof_manifest.txt
# the manifest names the passes of THIS run and the sources they were taken on: tools/prof_summary.py and make_traffic.py read
# only what it lists, so pass directories an earlier run left in the local gpurun_out/ (gpurun merges, it never deletes) cannot be
# folded into a new summary
SHA=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_sha())")
echo "source_sha256 $SHA" > $OUT/prof_manifest.txt
echo "pass prof_kt" >> $OUT/prof_manifest.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python3 $ARGS > $OUT/prof_kt.log 2>&1 || exit 1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_pmc_$tag -- python3 $ARGS > $OUT/prof_pmc_$tag.log 2>&1 || echo "pass failed: $pass"
  echo "pass prof_pmc_$tag" >> $OUT/prof_manifest.txt
  echo "pass done: $pass" >> $OUT/prof_progress.log
done
echo done
