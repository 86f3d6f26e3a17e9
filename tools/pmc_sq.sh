cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pq_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pq_$i -o r -- python $GRAFT_REPO_ROOT/tools/gpu_forward_only.py 128 160 1 > $OUT/run_$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pq_$i/r_results.db $OUT/sq_pass$i.md > /dev/null 2>&1
done
ls -la $OUT
