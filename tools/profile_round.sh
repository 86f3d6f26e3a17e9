# rocprofv3 evidence of a round (run on the GPU box):  bash tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/*.md
# kernel traces (--kernel-trace --stats only) of one full restoration of C2 / C4 / C5, and HBM-traffic PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, never combined with other trace domains) of the U-Net passes C4 and C5 are made of.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-c2 c4 c5}; do
  rm -rf /tmp/kt_$wl
  # (C5: eager launches - rocprofv3 7.2 segfaults while tracing the ~1000-node hipGraph of one Euler step)
  EXTRA=""; [ $wl = c5 ] && EXTRA="--no-graph"
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o r -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extra $EXTRA > $OUT/bench_$wl.json 2> /dev/null
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt_$wl/r_results.db $OUT/kernel_trace_bench_$wl.md > /dev/null
done
pmc() {  # name, command...
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -o r -- "$@" > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pmc_${name}_$c/r_results.db $OUT/pmc_${name}_$c.md > /dev/null
  done
}
[ -n "$NO_PMC" ] && exit 0
pmc fwd_c2 python $GRAFT_REPO_ROOT/tools/gpu_forward_only.py 128 160 2
pmc fwd_c4 python $GRAFT_REPO_ROOT/tools/gpu_forward_only.py 256 80 2
pmc vjp_c5 python $GRAFT_REPO_ROOT/tools/gpu_vjp_only.py 256 32 1
ls -la $OUT
