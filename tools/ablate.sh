# per-layer-class timing of one forward (160 images, 128^2) with kernel ablation switches / phase tracing
# needs the profiling build:  python __graft_entry__.py --debug
mkdir -p gpurun_out/abl
# ABL="0 1 2 ..." : ablation switches (dbg build);  default: phase tracing only (trace build)
if [ -n "$ABL" ]; then export PNPFLOW_HIP_LIB=$PWD/pnpflow_amd/libpnpflow_hip_dbg.so; else export PNPFLOW_HIP_LIB=$PWD/pnpflow_amd/libpnpflow_hip_trace.so PNPFLOW_HIP_TRACE=1; fi
for d in ${ABL:-0}; do
  PNPFLOW_HIP_DBG=$d python tools/gpu_layer_profile.py 128 160 gpurun_out/abl/l$d.csv > /dev/null 2>&1
  echo "== dbg $d"; python tools/layer_summary.py gpurun_out/abl/l$d.csv | head -${ABL_LINES:-14}
done
