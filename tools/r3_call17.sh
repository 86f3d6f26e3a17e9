#!/bin/bash
set -u
OUT=gpurun_out/r3c17; mkdir -p $OUT
{
timeout 300 python tools/gpu_dma_check.py run celeba128 40 1 $OUT/base.npy
for R in 2 4 8; do
  PNPFLOW_HIP_PT=$R timeout 300 python tools/gpu_dma_check.py run celeba128 40 1 $OUT/pt$R.npy
  python tools/gpu_dma_check.py cmp $OUT/base.npy $OUT/pt$R.npy 0
done
for rep in 1 2; do
for R in 0 2 4 8 -16; do
  PNPFLOW_HIP_PT=$R timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$R.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$R.csv > $OUT/l_$R.txt; echo "-- PT=$R"; grep "^total\|H= 128 Cout=  32 K=  288\|H= 128 Cout=  32 K=  576\|H=  64 Cout=  64 K=  576\|H=  64 Cout=  64 K= 1152" $OUT/l_$R.txt
done
done
rm -f $OUT/*.npy
} 2>&1 | grep -v amdgpu | tee $OUT/log.txt
