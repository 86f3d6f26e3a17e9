#!/bin/bash
set -u
OUT=gpurun_out/r3c20; mkdir -p $OUT
L=$PWD/pnpflow_amd/libpnpflow_hip
{
for i in 1 2 3; do
for v in new t1full; do
  if [ $v = new ]; then unset PNPFLOW_HIP_LIB; else export PNPFLOW_HIP_LIB=${L}_$v.so; fi
  PNPFLOW_PREC=2 timeout 300 python tools/gpu_forward_only.py 128 160 6 | sed "s/^/$v  /"
  PNPFLOW_PREC=2 timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/$v  /"
done
done
} 2>&1 | grep -v amdgpu | tee $OUT/log.txt
