cd $GRAFT_REPO_ROOT
O=gpurun_out/node_v2_pmc.txt; : > $O
export CAP_NO_TWO_LANES=1
for v in "" v1; do
  export CAP_LIB_VARIANT=$v; [ -z "$v" ] && unset CAP_LIB_VARIANT
  echo "== variant '${v}' (empty = product, the round-6 node test)" >> $O
  bash tools/tree_pmc.sh "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" 2>&1 | grep -E "k_trace_closest8|k_trace_any<" >> $O
  bash tools/tree_pmc.sh "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" 2>&1 | grep -E "k_trace_closest8|k_trace_any<" >> $O
done
cat $O
