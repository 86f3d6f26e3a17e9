# HBM traffic of the conv-GEMM launches of the U-Net forward at the solver's U-Net batch (separate PMC passes, no tracing
# domains besides the kernel trace):  bash tools/pmc_traffic.sh [dim] [B]  -> gpurun_out/pmc/{fetch,write}.md
cd /tmp && export TMPDIR=/tmp
DIM=${1:-128}; B=${2:-160}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o r -- python $GRAFT_REPO_ROOT/tools/gpu_forward_only.py $DIM $B 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc
python tools/prof_summary.py /tmp/pmc_FETCH_SIZE/r_results.db gpurun_out/pmc/fetch.md > /dev/null
python tools/prof_summary.py /tmp/pmc_WRITE_SIZE/r_results.db gpurun_out/pmc/write.md > /dev/null
python - <<'PY'
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\| `(.*?)` \| (FETCH_SIZE|WRITE_SIZE) \| ([0-9.e+]+) \| (\d+) \|", l)
        if m: d[m.group(1)] = (float(m.group(3)), int(m.group(4)))
    return d
f, w = load("gpurun_out/pmc/fetch.md"), load("gpurun_out/pmc/write.md")
tot_b = 0; tot_n = 0
print("| kernel | launches | read MB/launch (2 x FETCH_SIZE) | write MB/launch |")
print("|---|---|---|---|")
for k in sorted(f, key=lambda k: -f[k][0]):
    fs, n = f[k]; ws = w.get(k, (0, n))[0]
    print(f"| `{k[:60]}` | {n} | {2*fs*1024/n/1e6:.1f} | {ws*1024/n/1e6:.1f} |")
    if "conv_mfma" in k: tot_b += (2 * fs + ws) * 1024; tot_n += n
print(f"\nconv-GEMM family: {tot_n} launches, {tot_b/tot_n/1e6:.1f} MB of HBM traffic per launch on average")
PY
