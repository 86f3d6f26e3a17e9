# Same-box A/B of kernel variants (box-to-box variation is +-2-4 %, so every kernel decision is taken from one gpurun call).
#   build (here, no GPU):   bash tools/ab_variants.sh build
#   run (on the GPU box):   bash tools/ab_variants.sh run [dim] [B] [reps]
set -e
cd "$(dirname "$0")/.."
SRC="engine.hip conv_mfma.hip conv_mfma16.hip conv_ws.hip unet_misc.hip attention.hip unet_bwd.hip pointwise.hip fft2.hip metrics.hip fir_ops.hip ncsnpp_ops.hip"
if [ "$1" = "build" ]; then
  mkdir -p gpurun_out/ab
  build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-result -Wno-unused-value $2 -o pnpflow_amd/libpnpflow_hip_ab_$1.so $(for f in $SRC; do echo pnpflow_amd/csrc/$f; done) & }
  build cur ""
  build nopad "-DPF_ROW_PAD=0"
  wait
  ls -la pnpflow_amd/libpnpflow_hip_ab_*.so
  exit 0
fi
DIM=${2:-128}; NB=${3:-160}; R=${4:-3}
run() { echo -n "$1 "; env PNPFLOW_HIP_LIB=$PWD/pnpflow_amd/libpnpflow_hip_ab_$2.so $3 timeout 180 python tools/gpu_forward_only.py $DIM $NB 8 | tail -1; }
for i in $(seq $R); do
  run "cur (padded patch rows)" cur ""
  run "round-1 patch layout   " nopad ""
done
