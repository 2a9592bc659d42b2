#!/bin/bash
# Memory-side PMC passes for the fused kernel (run through gpurun).  Few counters per pass (a hardware block offers ~4), each
# pass under its own timeout, progress appended to gpurun_out/prof_mem_progress.log.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
            "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
            "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_ATOMIC_sum" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_pmc_m$i -- python3 $ROOT/tools/quick_stages.py 1 > $OUT/prof_pmc_m$i.log 2>&1
  echo "pass $i rc=$? : $pass" >> $OUT/prof_mem_progress.log
done
echo done
