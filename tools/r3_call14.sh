#!/bin/bash
set -u
OUT=gpurun_out/r3c14; mkdir -p $OUT
{
for B in 4 8 16 32 80; do timeout 300 python tools/gpu_forward_only.py 256 $B 6; done
for B in 8 16 32 64 160; do timeout 300 python tools/gpu_forward_only.py 128 $B 6; done
} 2>&1 | grep -v amdgpu > $OUT/batch_sweep.log
cat $OUT/batch_sweep.log
