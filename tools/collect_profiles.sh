#!/bin/bash
# profiles/<tag>_* from what the profiling passes left in gpurun_out/ (run here, after gpurun has merged the outputs back):
#   gpurun: tools/prof.sh, tools/tree_trace.sh, tools/post_trace.sh, tools/tree_pmc.sh x 3 (see README "Profiles"), and
#   tools/w8_counts.py on the diagnostic build (make -B EXTRA=-DCAP_W8_COUNT; rebuild without it afterwards).
# Usage: bash tools/collect_profiles.sh r02
set -e
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
python tools/make_traffic.py "$TAG" gpurun_out/w8_counts.json gpurun_out/w8_counts_big.json > /dev/null
python tools/prof_summary.py > "profiles/${TAG}_rocprofv3_summary.txt" 2>&1
# one table per workload (cornell / ext / tree on one lane / big on one lane): every avg_launch_ms of the bench line can be
# recomputed from one file
for wl in cornell ext tree big config3 config5; do
    f=$(ls -t gpurun_out/prof_${wl}_kt/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats_${wl}.csv"
done
{
    echo "# tools/tree_trace.sh: per-launch durations, one batch (32 spp) of the 262 k-triangle scene"
    cat gpurun_out/tree_trace.txt
    echo
    echo "# tools/tree_pmc.sh passes (same command, counters summed over the dispatches of the run)"
    for f in tree_l2 tree_sq tree_ta; do
        echo "## $f"
        cat gpurun_out/$f.txt
    done
    echo
    echo "# tools/w8_counts.py (diagnostic build): traversal steps of k_trace_closest8 per ray"
    tail -1 gpurun_out/w8_counts.json
    echo "# the same on the 16.8 M-triangle hall (bench.py big_variant)"
    tail -1 gpurun_out/w8_counts_big.json
} > "profiles/${TAG}_tree_path.txt"
{
    echo "# tools/post_trace.sh: reconstruction chain at 1920x1080 (kernel trace of tools/time_post.py), exact weights"
    cat gpurun_out/post_trace.txt
    echo
    echo "# POST_MODE=fast tools/post_trace.sh: CapPostSettings::fast_weights"
    cat gpurun_out/post_trace_fast.txt
} > "profiles/${TAG}_post_chain.txt"
{
    echo "# tools/shard_trace.sh: per-launch durations of ONE step of shard 0 of N (the compute side of the 1 -> N curve, bench.py shard_cost)"
    for f in shard1_cornell shard8_cornell shard1_sponza shard8_sponza; do
        echo "## $f"
        cat gpurun_out/$f.txt
    done
} > "profiles/${TAG}_shard_cost.txt"
echo "profiles/${TAG}_* written; now: gpurun -- 'python bench.py > gpurun_out/bench_${TAG}.json' and copy it to profiles/${TAG}_bench.json"
