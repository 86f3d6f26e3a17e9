#!/bin/bash
set -u
OUT=gpurun_out/r3c11; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python tools/gpu_pointwise.py > $OUT/pointwise.txt 2>&1; grep -v amdgpu $OUT/pointwise.txt
PNPFLOW_HIP_POISON=1 PNPFLOW_HIP_DMA=2 timeout 900 python -m pytest tests -m gpu -q -k "forward or trajectory or 100x5 or first_outer" > $OUT/pytest_poison_dma2.log 2>&1; tail -3 $OUT/pytest_poison_dma2.log
