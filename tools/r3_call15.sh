#!/bin/bash
set -u
OUT=gpurun_out/r3c15; mkdir -p $OUT
for B in 8 16 32 160; do
  timeout 300 python tools/gpu_layer_profile.py 256 $B $OUT/l256_$B.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l256_$B.csv > $OUT/l256_$B.txt
  echo "== 256^2 B=$B"; head -14 $OUT/l256_$B.txt
done
for B in 32 64 160; do
  timeout 300 python tools/gpu_layer_profile.py 128 $B $OUT/l128_$B.csv > /dev/null 2>&1
  python tools/layer_summary.py $OUT/l128_$B.csv > $OUT/l128_$B.txt
  echo "== 128^2 B=$B"; head -10 $OUT/l128_$B.txt
done
