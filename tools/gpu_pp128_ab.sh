# Same-box A/B of conv_pp128.hip (PNPFLOW_HIP_PP128=0 keeps its launches on conv_mfma16_kernel): whole-forward equivalence, then per-launch
# HIP-event times of one forward at the headline U-Net batch, the 128-channel classes and the total.  bash tools/gpu_pp128_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/p128
for shape in "afhq256 80" "afhq256 81"; do
  set -- $shape
  for pp in 0 1; do
    PNPFLOW_HIP_PP128=$pp timeout 300 python tools/gpu_dma_check.py run $1 $2 1 /tmp/p128_v_$1_$2_$pp.npy
  done
  python tools/gpu_dma_check.py cmp /tmp/p128_v_$1_$2_0.npy /tmp/p128_v_$1_$2_1.npy 2e-6
done
for shape in "256 160" "256 80"; do
  set -- $shape
  for pp in 0 1 0 1; do
    PNPFLOW_HIP_PP128=$pp timeout 300 python tools/gpu_layer_profile.py $1 $2 gpurun_out/p128/l$1_$2_$pp.csv > /dev/null 2>&1
    echo "== $1^2, B = $2, PNPFLOW_HIP_PP128=$pp"
    python tools/layer_summary.py gpurun_out/p128/l$1_$2_$pp.csv | grep -E "total|Cout= 128 K=.* s=1 up=0"
  done
done
