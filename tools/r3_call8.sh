#!/bin/bash
# GPU call 8: weight repack ([hi | lo][Cout][16] blocks: one contiguous 1 KiB run per fragment load) vs the interleaved layout, same box
set -u
OUT=gpurun_out/r3c8; mkdir -p $OUT
L=pnpflow_amd/libpnpflow_hip
{
PNPFLOW_HIP_LIB=${L}_wold.so timeout 300 python tools/gpu_dma_check.py run celeba128 160 1 $OUT/old.npy
timeout 300 python tools/gpu_dma_check.py run celeba128 160 1 $OUT/new.npy
python tools/gpu_dma_check.py cmp $OUT/old.npy $OUT/new.npy 1e-7
PNPFLOW_HIP_LIB=${L}_wold.so PNPFLOW_HIP_DMA=2 timeout 300 python tools/gpu_dma_check.py run celeba128 8 2 $OUT/old2.npy
PNPFLOW_HIP_DMA=2 timeout 300 python tools/gpu_dma_check.py run celeba128 8 2 $OUT/new2.npy
python tools/gpu_dma_check.py cmp $OUT/old2.npy $OUT/new2.npy 1e-7
rm -f $OUT/*.npy
for i in 1 2 3; do
  PNPFLOW_HIP_LIB=${L}_wold.so timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/old  /"
  timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/new  /"
done
for i in 1 2; do
  PNPFLOW_HIP_LIB=${L}_wold.so timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/old  /"
  timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/new  /"
done
PNPFLOW_PREC=2 PNPFLOW_HIP_LIB=${L}_wold.so timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/p2 old  /"
PNPFLOW_PREC=2 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/p2 new  /"
timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/layers_new.csv > /dev/null; python tools/layer_summary.py $OUT/layers_new.csv | head -12
} > $OUT/ab.log 2>&1
grep -v amdgpu.ids $OUT/ab.log
