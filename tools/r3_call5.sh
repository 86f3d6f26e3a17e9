#!/bin/bash
set -u
OUT=gpurun_out/r3c5; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -k "main_reads or blur or degradations or grad_step or deblur or fourier or adjoint" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
{
echo "== tile A/B on the 32 / 64 channel levels (same box)"
for i in 1 2; do
  timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/base        /"
  PNPFLOW_HIP_TILE_L0=8 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/L0=8        /"
  PNPFLOW_HIP_TILE_L1=8 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/L1=8        /"
  PNPFLOW_HIP_TILE_L0=8 PNPFLOW_HIP_TILE_L1=8 timeout 300 python tools/gpu_forward_only.py 128 160 8 | sed "s/^/L0=8 L1=8   /"
done
PNPFLOW_HIP_TILE_L0=8 PNPFLOW_HIP_TILE_L1=8 timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/L0=8 L1=8   /"
timeout 300 python tools/gpu_forward_only.py 256 80 4 | sed "s/^/base        /"
for v in base small; do
  if [ $v = small ]; then export PNPFLOW_HIP_TILE_L0=8 PNPFLOW_HIP_TILE_L1=8; fi
  timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/layers_$v.csv > /dev/null
  python tools/layer_summary.py $OUT/layers_$v.csv > $OUT/layers_$v.txt; echo "-- $v"; head -12 $OUT/layers_$v.txt
done
} > $OUT/ab.log 2>&1
grep -v amdgpu.ids $OUT/ab.log
