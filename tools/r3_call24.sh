#!/bin/bash
set -u
OUT=gpurun_out/r3c24; mkdir -p $OUT
L=$PWD/pnpflow_amd/libpnpflow_hip
for rep in 1 2; do
for v in base noinput noepi noinout; do
  if [ $v = base ]; then unset PNPFLOW_HIP_LIB; else export PNPFLOW_HIP_LIB=${L}_$v.so; fi
  PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$v.csv > $OUT/run_$v.log 2>&1 || tail -3 $OUT/run_$v.log
  python tools/layer_summary.py $OUT/l_$v.csv > $OUT/l_$v.txt; echo "-- $v"; grep "^total\|H= 128 Cout=  32 K=  288\|H= 128 Cout=  32 K=  576\|H=  16 Cout= 256 K= 2304\|H=  64 Cout=  64 K=  576 \|H=  32 Cout= 128 K= 1152" $OUT/l_$v.txt
done
done 2>&1 | tee $OUT/log.txt
