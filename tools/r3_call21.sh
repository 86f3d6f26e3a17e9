#!/bin/bash
# robustness passes over the whole -m gpu suite: poisoned workspaces, LDS-DMA conv forced on / off
set -u
OUT=gpurun_out/r3c21; mkdir -p $OUT
PNPFLOW_HIP_POISON=1 timeout 900 python -m pytest tests -m gpu -q -x > $OUT/poison.log 2>&1; echo "poison: $(tail -1 $OUT/poison.log)"
PNPFLOW_HIP_DMA=2 timeout 900 python -m pytest tests -m gpu -q > $OUT/dma2.log 2>&1; echo "dma=2: $(tail -1 $OUT/dma2.log)"
PNPFLOW_HIP_DMA=0 timeout 900 python -m pytest tests -m gpu -q > $OUT/dma0.log 2>&1; echo "dma=0: $(tail -1 $OUT/dma0.log)"
grep -h "FAILED\|ERROR" $OUT/*.log | head -20
