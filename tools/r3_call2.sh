#!/bin/bash
# GPU call 2 of round 3: same-box A/B of conv_dma variants (A-fragment software pipelining, weight ring carried across chunks),
# kernel trace (prep vs conv split) and SQ counters of the base variant.
set -u
OUT=gpurun_out/r3c2; mkdir -p $OUT
L=pnpflow_amd/libpnpflow_hip
{
echo "== correctness of every variant vs the register-staged kernel (bit-exact expected)"
PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_dma_check.py run celeba128 160 1 $OUT/ref.npy
for v in base apipe2 carry both2; do
  PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_dma_check.py run celeba128 160 1 $OUT/$v.npy
  python tools/gpu_dma_check.py cmp $OUT/ref.npy $OUT/$v.npy 1e-6
done
PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_dma_check.py run celeba128 160 2 $OUT/ref2.npy
for v in both2; do
  PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_dma_check.py run celeba128 160 2 $OUT/${v}_p2.npy
  python tools/gpu_dma_check.py cmp $OUT/ref2.npy $OUT/${v}_p2.npy 5e-3
done
rm -f $OUT/*.npy
echo "== forward time, 3 interleaved rounds"
for i in 1 2 3; do
  PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/off     /"
  for v in base apipe2 carry both2; do PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/$v  /"; done
done
echo "== precision 2"
PNPFLOW_PREC=2 PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/p2 off     /"
for v in base both2; do PNPFLOW_PREC=2 PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/p2 $v  /"; done
echo "== per-layer profile"
for v in base apipe2 carry both2; do
  PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/layers_$v.csv > /dev/null
  python tools/layer_summary.py $OUT/layers_$v.csv > $OUT/layers_$v.txt; echo "-- $v"; grep "^total\|H=  16 Cout= 256 K= 2304\|H=  32 Cout= 128 K= 1152\|H=  16 Cout= 768\|H=  32 Cout= 256 K= 2304\|H=  16 Cout= 256 K= 2816" $OUT/layers_$v.txt
done
} > $OUT/ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base both2; do
  rm -rf /tmp/kt_$v
  PNPFLOW_HIP_LIB=$R/${L}_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o r -- python $R/tools/gpu_forward_only.py 128 160 3 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/kt_$v/r_results.db $R/$OUT/kernel_trace_fwd_$v.md > /dev/null 2>&1
  rm -rf /tmp/pq_$v
  PNPFLOW_HIP_LIB=$R/${L}_$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/pq_$v -o r -- python $R/tools/gpu_forward_only.py 128 160 1 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/pq_$v/r_results.db $R/$OUT/pmc_sq_fwd_$v.md > /dev/null 2>&1
done
cd $R
cat $OUT/ab.log | grep -v amdgpu.ids
head -14 $OUT/kernel_trace_fwd_base.md
