# HBM traffic of the conv-family launches (conv_pp / conv_mfma16 / conv_dma / prep_split) of the U-Net forward at the solver's U-Net batch, per
# BASELINE workload - separate PMC passes (FETCH_SIZE, WRITE_SIZE), no tracing domains besides the kernel trace, FETCH doubled as
# MI355X_MICROARCH.md prescribes for gfx950 (WRITE_SIZE uncalibrated, taken as reported).
#   bash tools/pmc_traffic.sh            -> gpurun_out/pmc/{traffic.json, <workload>_{fetch,write}.md}; copy traffic.json to profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc
for wl in ${WLS:-"c2_256 256 160" "c2 128 160" "c3 128 320" "c4 256 80"}; do
  set -- $wl; name=$1; DIM=$2; B=$3
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -o r -- python $R/tools/gpu_forward_only.py $DIM $B 2 > /dev/null 2>&1
  done
  python $R/tools/prof_summary.py /tmp/pmc_${name}_FETCH_SIZE/r_results.db $R/gpurun_out/pmc/${name}_fetch.md > /dev/null
  python $R/tools/prof_summary.py /tmp/pmc_${name}_WRITE_SIZE/r_results.db $R/gpurun_out/pmc/${name}_write.md > /dev/null
done
cd $R
python - <<'PY'
import json, os, re, subprocess
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\| `(.*?)` \| (FETCH_SIZE|WRITE_SIZE) \| ([0-9.e+]+) \| (\d+) \|", l)
        if m: d[m.group(1)] = (float(m.group(3)), int(m.group(4)))
    return d
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ""
out = {}
for name in ("c2_256", "c2", "c3", "c4"):
    fp, wp = f"gpurun_out/pmc/{name}_fetch.md", f"gpurun_out/pmc/{name}_write.md"
    if not (os.path.isfile(fp) and os.path.isfile(wp)): continue
    f, w = load(fp), load(wp)
    tot_b = 0.0; tot_n = 0; rows = []
    for k in sorted(f, key=lambda k: -f[k][0]):
        fs, n = f[k]; ws = w.get(k, (0, n))[0]
        conv = ("conv_mfma" in k or "conv_dma" in k or "conv_pp" in k or "conv_sp" in k or "prep_split" in k)
        rows.append({"kernel": k[:80], "launches": n, "read_mb_per_launch": round(2 * fs * 1024 / n / 1e6, 2), "write_mb_per_launch": round(ws * 1024 / n / 1e6, 2), "conv_family": conv})
        if conv:
            tot_b += (2 * fs + ws) * 1024
            tot_n += n if "prep_split" not in k else 0          # a prep pass is part of its conv launch's cost, not a launch of its own
    if tot_n:
        out[name] = {"bytes_per_launch": round(tot_b / tot_n, 1), "launches": tot_n, "commit": commit,
                     "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 2 forwards + warm-up of tools/gpu_forward_only.py, tools/pmc_traffic.sh; 2 x FETCH_SIZE + WRITE_SIZE, KiB units; profiles/r06_pmc_{name}_{{fetch,write}}.md",
                     "kernels": rows[:16]}
json.dump(out, open("gpurun_out/pmc/traffic.json", "w"), indent=1)
for k, v in out.items(): print(k, v["bytes_per_launch"] / 1e6, "MB per conv launch over", v["launches"], "launches")
PY
