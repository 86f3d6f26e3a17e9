#!/bin/bash
set -u
OUT=gpurun_out/r3c16; mkdir -p $OUT
L=$PWD/pnpflow_amd/libpnpflow_hip
for rep in 1 2; do
for v in base nosilu rawstage; do
  if [ $v = base ]; then unset PNPFLOW_HIP_LIB; else export PNPFLOW_HIP_LIB=${L}_$v.so; fi
  timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$v.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$v.csv > $OUT/l_$v.txt; echo "-- $v"; head -8 $OUT/l_$v.txt
done
done
