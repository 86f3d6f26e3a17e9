#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r3c13; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $REPO/tools/gpu_vjp_only.py 256 32 3 > $OUT/run.log 2>&1
cd $REPO
python tools/trace_by_grid.py $OUT/tr gn_bwd_post $OUT/post_by_grid.md
python tools/trace_by_grid.py $OUT/tr gn_bwd_pre $OUT/pre_by_grid.md | tail -5
python tools/trace_by_grid.py $OUT/tr conv_mfma16 $OUT/conv_by_grid.md | head -40
python tools/trace_by_grid.py $OUT/tr kernel $OUT/all_by_grid.md > /dev/null
rm -rf $OUT/tr
tail -3 $OUT/run.log
