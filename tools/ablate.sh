mkdir -p gpurun_out/abl
for d in ${ABL:-0 15 31 47 79 143 255 16 32 64 128}; do
  PNPFLOW_HIP_DBG=$d python tools/gpu_layer_profile.py 128 160 gpurun_out/abl/l$d.csv > /dev/null 2>&1
  echo "== dbg $d"; python tools/layer_summary.py gpurun_out/abl/l$d.csv | head -${ABL_LINES:-7}
done
