#!/bin/bash
set -u
OUT=gpurun_out/r3c26; mkdir -p $OUT
L=$PWD/pnpflow_amd/libpnpflow_hip
for rep in 1 2; do
for v in base kc16 kc16lb4; do
  unset PNPFLOW_HIP_LIB PNPFLOW_HIP_KC_L0
  [ $v = kc16 ] && export PNPFLOW_HIP_KC_L0=16
  [ $v = kc16lb4 ] && export PNPFLOW_HIP_KC_L0=16 PNPFLOW_HIP_LIB=${L}_lb4.so
  timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$v.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$v.csv > $OUT/l_$v.txt; echo "-- $v"; grep "^total\|H= 128 Cout=  32 K=  288\|H= 128 Cout=  32 K=  576\|H= 128 Cout=  32 K=  352" $OUT/l_$v.txt
done
done 2>&1 | tee $OUT/log.txt
