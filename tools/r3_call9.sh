#!/bin/bash
# GPU call 9: the default bench line (c2_256 headline + secondary configs) and rocprofv3 kernel traces of the bench command
set -u
OUT=gpurun_out/r3c9; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; tail -4 $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for wl in c2_256 c2; do
  rm -rf /tmp/kt_$wl
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o r -- python $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $R/$OUT/bench_$wl.json 2> /dev/null
  python $R/tools/prof_summary.py /tmp/kt_$wl/r_results.db $R/$OUT/kernel_trace_bench_$wl.md > /dev/null 2>&1
done
cd $R; head -14 $OUT/kernel_trace_bench_c2_256.md
