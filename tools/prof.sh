#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun): ONE WORKLOAD PER rocprofv3 RUN -- kernel trace + stats, then PMC passes
# (separate runs; never with --sys-trace) -- so that every per-kernel average can be recomputed from one file (VERDICT r3 weak 8:
# one table used to mix Cornell, EXT and the tree path with two overlapping lanes).  Workloads:
#   cornell  bench.py --no-extras                      the contract line (k_trace_shade<false,false,false,true>)
#   ext      bench.py --only ext                       BASELINE's literal "Lambert+GGX" (k_trace_shade<false,true,false,true>)
#   tree     bench.py --only tree, CAP_NO_TWO_LANES=1  262 k-triangle hall, one lane: a launch's duration is its own
#   big      bench.py --only big,  CAP_NO_TWO_LANES=1  16.8 M-triangle hall (the HBM-bound traversal frames)
#   config3  bench.py --only config3                   BASELINE configs[2], "the HBM-roofline run": 3840x2160, 512 spp, EXT model
#   config5  bench.py --only config5                   BASELINE configs[4], one rank's share (shard 0 of 8 of 4096x4096, 1024 spp, depth 16)
# Outputs under gpurun_out/prof_<workload>_*; tools/make_traffic.py and tools/prof_summary.py condense them into profiles/.
#   PROF_WORKLOADS="cornell big" bash tools/prof.sh    (default: all four)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
WORKLOADS=${PROF_WORKLOADS:-"cornell ext tree big config3 config5"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the manifest names the passes of THIS run and the sources they were taken on: tools/prof_summary.py and make_traffic.py read
# only what it lists, so pass directories an earlier run left in the local gpurun_out/ (gpurun merges, it never deletes) cannot be
# folded into a new summary
SHA=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_sha())")
echo "source_sha256 $SHA" > $OUT/prof_manifest.txt
: > $OUT/prof_progress.log
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
for wl in $WORKLOADS; do
  unset CAP_NO_TWO_LANES
  case $wl in
    cornell) ARGS="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" ;;
    ext)     ARGS="$ROOT/bench.py --only ext --steps 2" ;;
    tree)    ARGS="$ROOT/bench.py --only tree"; export CAP_NO_TWO_LANES=1 ;;
    big)     ARGS="$ROOT/bench.py --only big"; export CAP_NO_TWO_LANES=1 ;;
    config3) ARGS="$ROOT/bench.py --only config3" ;;
    config5) ARGS="$ROOT/bench.py --only config5" ;;
    *) echo "unknown workload $wl"; exit 1 ;;
  esac
  rm -rf $OUT/prof_${wl}_kt $OUT/prof_${wl}_pmc_*
  echo "pass prof_${wl}_kt" >> $OUT/prof_manifest.txt
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${wl}_kt -- python3 $ARGS > $OUT/prof_${wl}_kt.log 2>&1 || { echo "kernel trace of $wl failed"; tail -5 $OUT/prof_${wl}_kt.log; exit 1; }
  echo "kt done: $wl" >> $OUT/prof_progress.log
  passes=("$SQ1" "FETCH_SIZE" "WRITE_SIZE")
  [ $wl != ext ] && [ $wl != config3 ] && [ $wl != config5 ] && passes+=("$SQ2" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE")
  for pass in "${passes[@]}"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_${wl}_pmc_$tag -- python3 $ARGS > $OUT/prof_${wl}_pmc_$tag.log 2>&1 || echo "pass failed: $wl $pass"
    echo "pass prof_${wl}_pmc_$tag" >> $OUT/prof_manifest.txt
    echo "pass done: $wl $pass" >> $OUT/prof_progress.log
  done
done
unset CAP_NO_TWO_LANES
echo done
