# SQ / TA / TCP / GRBM counters of one forward of the 256^2 net at B = 80 (separate --pmc passes, kernel trace only), per kernel:
#   bash tools/pmc_level0.sh   -> gpurun_out/pmc_level0/pass{1,2,3}.md   (what profiles/r04_pmc_level0_counters.md is made from)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_level0; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$i -o r -- python $GRAFT_REPO_ROOT/tools/gpu_forward_only.py 256 80 1 > $O/run_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pl_$i/r_results.db $O/pass$i.md > /dev/null 2>&1
done
ls $O
