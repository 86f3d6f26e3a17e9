#!/bin/bash
set -u
OUT=gpurun_out/r3c25; mkdir -p $OUT
for rep in 1 2; do
for st in 0 1 2 4; do
  PNPFLOW_HIP_STAGGER=$st timeout 300 python tools/gpu_layer_profile.py 128 160 $OUT/l_$st.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l_$st.csv > $OUT/l_$st.txt; echo "-- stagger $st"; grep "^total\|H= 128 Cout=  32 K=  288\|H= 128 Cout=  32 K=  576\|H=  64 Cout=  64 K=  576 \|H=  32 Cout= 128 K= 1152\|H=  16 Cout= 768" $OUT/l_$st.txt
done
done 2>&1 | tee $OUT/log.txt
for st in 0 2; do PNPFLOW_HIP_STAGGER=$st timeout 300 python tools/gpu_forward_only.py 256 80 4 2>&1 | grep forward | sed "s/^/stagger $st  /"; done | tee -a $OUT/log.txt
