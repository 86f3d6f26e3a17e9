#!/bin/bash
set -u
OUT=gpurun_out/r3c23; mkdir -p $OUT
for rep in 1 2; do
for mw in 512 1000; do
  PNPFLOW_HIP_DMA=0 PNPFLOW_HIP_MIN_WGS=$mw timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$mw.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$mw.csv > $OUT/l_$mw.txt; echo "-- DMA=0 MIN_WGS=$mw"; grep "^total\|H=  16 Cout= 256\|H=  16 Cout= 768" $OUT/l_$mw.txt | head -8
done
done 2>&1 | tee $OUT/log.txt
