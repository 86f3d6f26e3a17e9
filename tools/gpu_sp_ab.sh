# Same-box A/B of conv_sp.hip against conv_pp128.hip (PNPFLOW_HIP_SP=0) and conv_mfma16 (both off).  bash tools/gpu_sp_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/sp
for shape in "afhq256 80" "afhq256 81"; do
  set -- $shape
  PNPFLOW_HIP_SP=0 PNPFLOW_HIP_PP128=0 timeout 300 python tools/gpu_dma_check.py run $1 $2 1 /tmp/sp_v_$1_$2_0.npy
  PNPFLOW_HIP_SP=1 timeout 300 python tools/gpu_dma_check.py run $1 $2 1 /tmp/sp_v_$1_$2_1.npy
  python tools/gpu_dma_check.py cmp /tmp/sp_v_$1_$2_0.npy /tmp/sp_v_$1_$2_1.npy 2e-6
done
for shape in "256 160" "256 80"; do
  set -- $shape
  for sp in 0 1 0 1; do
    PNPFLOW_HIP_SP=$sp timeout 300 python tools/gpu_layer_profile.py $1 $2 gpurun_out/sp/l$1_$2_$sp.csv > /dev/null 2>&1
    echo "== $1^2, B = $2, PNPFLOW_HIP_SP=$sp"
    python tools/layer_summary.py gpurun_out/sp/l$1_$2_$sp.csv | grep -E "total|Cout= 128 K=.* s=1 up=0"
  done
done
