#!/bin/bash
# Everything tools/collect_profiles.sh condenses into profiles/<tag>_*, in one gpurun call (~5 GPU-minutes):
#   gpurun --timeout 1190 -- 'bash tools/prof_all.sh'      then here:  bash tools/collect_profiles.sh r04
# The per-launch listings and counter passes of the tree path run with CAP_NO_TWO_LANES=1: with the two batch lanes (context.hip)
# kernels of two streams overlap and a launch's duration is no longer its own.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "prof.sh" > $OUT/prof_all_progress.log
bash tools/prof.sh > $OUT/prof_sh.log 2>&1 || echo "prof.sh failed" >> $OUT/prof_all_progress.log
export CAP_NO_TWO_LANES=1
echo "tree_trace" >> $OUT/prof_all_progress.log
bash tools/tree_trace.sh > $OUT/tree_trace.txt 2>&1
echo "tree_pmc" >> $OUT/prof_all_progress.log
bash tools/tree_pmc.sh "TCC_HIT_sum TCC_MISS_sum" > $OUT/tree_l2.txt 2>&1
bash tools/tree_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $OUT/tree_sq.txt 2>&1
bash tools/tree_pmc.sh "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" > $OUT/tree_ta.txt 2>&1
unset CAP_NO_TWO_LANES
echo "post" >> $OUT/prof_all_progress.log
bash tools/post_trace.sh > $OUT/post_trace.txt 2>&1
POST_MODE=fast bash tools/post_trace.sh > $OUT/post_trace_fast.txt 2>&1
echo "shards" >> $OUT/prof_all_progress.log
for n in 1 8; do
    bash tools/shard_trace.sh $n > $OUT/shard${n}_cornell.txt 2>&1
    CAP_NO_TWO_LANES=1 bash tools/shard_trace.sh $n sponza > $OUT/shard${n}_sponza.txt 2>&1
done
echo "w8_counts (diagnostic build)" >> $OUT/prof_all_progress.log
# the step-counting build is a VARIANT made beside the product build before the call (tools/build_variant.sh w8count -DCAP_W8_COUNT;
# capsaicin_amd/variants/, picked with CAP_LIB_VARIANT): nothing is compiled on the GPU box and the product build is never touched
if [ -f capsaicin_amd/variants/libcapsaicin_hip_w8count.so ]; then
    CAP_LIB_VARIANT=w8count timeout -k 5 200 python tools/w8_counts.py > $OUT/w8_counts.json 2> $OUT/w8_counts.err
    CAP_LIB_VARIANT=w8count timeout -k 5 300 python tools/w8_counts.py 8 > $OUT/w8_counts_big.json 2> $OUT/w8_counts_big.err
else
    echo "no capsaicin_amd/variants/libcapsaicin_hip_w8count.so: run tools/build_variant.sh w8count -DCAP_W8_COUNT first" >> $OUT/prof_all_progress.log
fi
echo "done" >> $OUT/prof_all_progress.log
tail -3 $OUT/prof_sh.log; tail -2 $OUT/tree_trace.txt; tail -1 $OUT/w8_counts.json | cut -c1-200
