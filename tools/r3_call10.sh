#!/bin/bash
set -u
OUT=gpurun_out/r3c10; mkdir -p $OUT
L=pnpflow_amd/libpnpflow_hip
timeout 900 python -m pytest tests -m gpu -q -k "blur or degradations or grad_step or deblur or fourier or adjoint or trajectory or lpips" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
{
echo "== pointwise block: fused 2-D blur vs the row + column pair"
timeout 300 python tools/gpu_pointwise.py | grep -i "blur"
PNPFLOW_HIP_BLUR_FUSED=0 timeout 300 python tools/gpu_pointwise.py | grep -i "blur" | sed "s/^/two-pass  /"
echo "== gn_bwd_post variants: retained forward + backward of 32 x 256^2 (same box)"
for i in 1 2; do
  timeout 300 python tools/gpu_vjp_only.py 256 32 3 | sed "s/^/base     /"
  for v in post2 post8 postnt post8nt; do PNPFLOW_HIP_LIB=${L}_$v.so timeout 300 python tools/gpu_vjp_only.py 256 32 3 | sed "s/^/$v   /"; done
done
} > $OUT/ab.log 2>&1
grep -v amdgpu.ids $OUT/ab.log
