# Same-box A/B of conv_sp.hip at the 64-channel level (PNPFLOW_HIP_SP=2: 128-channel level only -> conv_pp64 / conv_mfma16 keep level 1).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/sp64
for shape in "afhq256 40" "celeba128 129"; do
  set -- $shape
  PNPFLOW_HIP_SP=2 timeout 300 python tools/gpu_dma_check.py run $1 $2 1 /tmp/sp64_v_$1_$2_0.npy
  PNPFLOW_HIP_SP=1 timeout 300 python tools/gpu_dma_check.py run $1 $2 1 /tmp/sp64_v_$1_$2_1.npy
  python tools/gpu_dma_check.py cmp /tmp/sp64_v_$1_$2_0.npy /tmp/sp64_v_$1_$2_1.npy 2e-6
done
for shape in "256 160" "128 160"; do
  set -- $shape
  for sp in 2 1 2 1; do
    PNPFLOW_HIP_SP=$sp timeout 300 python tools/gpu_layer_profile.py $1 $2 gpurun_out/sp64/l$1_$2_$sp.csv > /dev/null 2>&1
    echo "== $1^2, B = $2, PNPFLOW_HIP_SP=$sp"
    python tools/layer_summary.py gpurun_out/sp64/l$1_$2_$sp.csv | grep -E "total|Cout=  64 K=.* s=1 up=0"
  done
done
