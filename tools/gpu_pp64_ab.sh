# Same-box A/B of conv_pp64.hip (PNPFLOW_HIP_PP64=0 keeps its launches on conv_mfma16_kernel): per-launch HIP-event times of one forward at
# the two U-Net batch shapes, the 64-channel classes and the total.  bash tools/gpu_pp64_ab.sh   (profiles/r04_level1_probes.md, table 5)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/p64
for shape in "256 80" "128 160"; do
  set -- $shape
  for pp in 0 1 0 1; do
    PNPFLOW_HIP_PP64=$pp timeout 300 python tools/gpu_layer_profile.py $1 $2 gpurun_out/p64/l$1_$pp.csv > /dev/null 2>&1
    echo "== $1^2, B = $2, PNPFLOW_HIP_PP64=$pp"
    python tools/layer_summary.py gpurun_out/p64/l$1_$pp.csv | grep -E "total|Cout=  64 K=.* s=1 up=0"
  done
done
