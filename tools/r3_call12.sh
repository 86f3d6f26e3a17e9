#!/bin/bash
set -u
OUT=gpurun_out/r3c12; mkdir -p $OUT
{
for i in 1 2; do
  timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/base            /"
  PNPFLOW_HIP_TILE_L0=32 PNPFLOW_HIP_KC_L0=16 timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/L0=32x16 KC16   /"
  PNPFLOW_HIP_KC_L0=16 timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/L0=16x16 KC16   /"
done
PNPFLOW_HIP_TILE_L0=32 PNPFLOW_HIP_KC_L0=16 timeout 300 python tools/gpu_dma_check.py run celeba128 8 1 $OUT/t32.npy
timeout 300 python tools/gpu_dma_check.py run celeba128 8 1 $OUT/t16.npy
python tools/gpu_dma_check.py cmp $OUT/t16.npy $OUT/t32.npy 1e-5
rm -f $OUT/*.npy
} > $OUT/ab.log 2>&1
grep -v amdgpu $OUT/ab.log
