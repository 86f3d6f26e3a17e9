#!/bin/bash
# GPU call 7: what bounds the 32-channel full-resolution conv (the dominant kernel at 256^2)?  SQ / TA / TCP / GRBM counters
set -u
OUT=gpurun_out/r3c7; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$OUT/counters_avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TA_TA_BUSY_sum TA_BUSY_avr" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/pq_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pq_$i -o r -- python $R/tools/gpu_forward_only.py 256 80 1 > $R/$OUT/run_$i.log 2>&1
  python $R/tools/prof_summary.py /tmp/pq_$i/r_results.db $R/$OUT/pass$i.md > /dev/null 2>&1
  echo "pass $i: $set -> $(grep -c '2, 1, 4, 1, 1, 0, 32' $R/$OUT/pass$i.md 2>/dev/null) rows"
done
cd $R
grep -h "conv_mfma16_kernel<2, 1, 4, 1, 1, 0, 32" $OUT/pass*.md | cut -c1-40,95-200
