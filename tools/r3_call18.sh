#!/bin/bash
# GPU call 18: the whole -m gpu suite, the default bench line, kernel traces of the bench command (headline, C2, C5)
set -u
OUT=gpurun_out/r3c18; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; tail -4 $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for wl in c2_256 c2 c5; do
  rm -rf /tmp/kt_$wl
  EXTRA=""; [ $wl = c5 ] && EXTRA="--no-graph"
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o r -- python $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extra $EXTRA > $R/$OUT/bench_$wl.json 2> /dev/null
  python $R/tools/prof_summary.py /tmp/kt_$wl/r_results.db $R/$OUT/kernel_trace_bench_$wl.md > /dev/null 2>&1
done
cd $R; head -12 $OUT/kernel_trace_bench_c2_256.md; head -12 $OUT/kernel_trace_bench_c5.md
