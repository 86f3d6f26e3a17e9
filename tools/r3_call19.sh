#!/bin/bash
set -u
OUT=gpurun_out/r3c19; mkdir -p $OUT
{
for i in 1 2; do
PNPFLOW_PREC=2 timeout 300 python tools/gpu_forward_only.py 128 160 6
PNPFLOW_PREC=2 timeout 300 python tools/gpu_forward_only.py 256 80 4
done
timeout 900 python -m pytest tests -m gpu -q -k "fp16 or mode or precision or forward or 100x5" 2>&1 | tail -4
} 2>&1 | grep -v amdgpu | tee $OUT/log.txt
