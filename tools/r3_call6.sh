#!/bin/bash
set -u
OUT=gpurun_out/r3c6; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1; tail -5 $OUT/pmc_traffic.log
cd /tmp && export TMPDIR=/tmp
for cfg in "256 160" "128 160"; do
  set -- $cfg
  rm -rf /tmp/kt_$1
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$1 -o r -- python $R/tools/gpu_forward_only.py $1 $2 3 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/kt_$1/r_results.db $R/$OUT/kernel_trace_fwd_$1_B$2.md > /dev/null 2>&1
done
cd $R
head -16 $OUT/kernel_trace_fwd_256_B160.md
timeout 900 python bench.py --workload c5 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 600 $OUT/bench_c5.json
