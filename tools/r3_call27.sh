#!/bin/bash
set -u
OUT=gpurun_out/r3c27; mkdir -p $OUT
for rep in 1 2; do
for v in 0 22; do
  PNPFLOW_HIP_DMA=0 PNPFLOW_HIP_TILE_L2=$v timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$v.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$v.csv > $OUT/l_$v.txt; echo "-- DMA=0 TILE_L2=$v"; grep "^total\|H=  16 Cout= 256 K= 2304\|H=  16 Cout= 256 K= 4608\|H=  32 Cout= 128 K= 1152\|H=  32 Cout= 128 K= 2304\|H=  16 Cout= 768\|H=  16 Cout= 256 K=  256" $OUT/l_$v.txt
done
done 2>&1 | tee $OUT/log.txt
PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_dma_check.py run celeba128 8 1 $OUT/a.npy > /dev/null 2>&1
PNPFLOW_HIP_DMA=0 PNPFLOW_HIP_TILE_L2=22 timeout 300 python tools/gpu_dma_check.py run celeba128 8 1 $OUT/b.npy > /dev/null 2>&1
python tools/gpu_dma_check.py cmp $OUT/a.npy $OUT/b.npy 1e-5; rm -f $OUT/*.npy
