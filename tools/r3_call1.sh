#!/bin/bash
# GPU call 1 of round 3: correctness of the LDS-DMA conv path (forward), layer profile and forward time with / without it, then the suite.
set -u
OUT=gpurun_out/r3c1; mkdir -p $OUT
{
echo "== correctness: DMA on / forced vs off"
for cfg in "celeba128 160 1" "celeba128 160 2" "celeba128 4 1" "celeba128 4 2" "afhq256 6 1"; do
  set -- $cfg
  PNPFLOW_HIP_DMA=0 timeout 300 python tools/gpu_dma_check.py run $1 $2 $3 $OUT/${1}_B$2_p$3_dma0.npy
  PNPFLOW_HIP_DMA=2 timeout 300 python tools/gpu_dma_check.py run $1 $2 $3 $OUT/${1}_B$2_p$3_dma2.npy
  tol=2e-5; [ "$3" = "2" ] && tol=5e-3
  python tools/gpu_dma_check.py cmp $OUT/${1}_B$2_p$3_dma0.npy $OUT/${1}_B$2_p$3_dma2.npy $tol
done
echo "== forward time A/B (same box): DMA 0 / 1, precision 1 and 2"
for i in 1 2; do
  for p in 1 2; do
    for d in 0 1; do PNPFLOW_PREC=$p PNPFLOW_HIP_DMA=$d timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/prec=$p DMA=$d  /"; done
  done
done
for d in 0 1; do PNPFLOW_HIP_DMA=$d timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/prec=1 DMA=$d  /"; done
echo "== per-layer profile, B = 160"
for d in 0 1; do
  PNPFLOW_HIP_DMA=$d timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/layers_dma$d.csv
  python tools/layer_summary.py $OUT/layers_dma$d.csv > $OUT/layers_dma$d.txt; head -40 $OUT/layers_dma$d.txt
done
} > $OUT/check.log 2>&1
rm -f $OUT/*.npy
echo "== pytest (default path)"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.log 2>&1; tail -3 $OUT/pytest_default.log
echo "== pytest forward / trajectory subset with the DMA path forced"
PNPFLOW_HIP_DMA=2 timeout 900 python -m pytest tests -m gpu -q -k "forward or trajectory or unet or fp16 or precision" > $OUT/pytest_dma2.log 2>&1; tail -3 $OUT/pytest_dma2.log
tail -60 $OUT/check.log
