# round-6 evidence run on the GPU box (one call):  bash tools/r06_final.sh   -> gpurun_out/r06f/*, gpurun_out/prof_r06/*, gpurun_out/pmc_level0/*
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc $?"
WLS="c2_256 c2 c5" NO_PMC=1 bash tools/profile_round.sh r06 > /dev/null 2>&1
bash tools/pmc_level0.sh > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/pmc_level0 > $O/pmc_conv_counters.md 2>&1
for s in "256 160" "128 160"; do
  set -- $s
  timeout 300 python tools/gpu_layer_profile.py $1 $2 $O/l$1_$2.csv > /dev/null 2>&1
  python tools/layer_summary.py $O/l$1_$2.csv > $O/l$1_$2.txt; head -1 $O/l$1_$2.txt
done
./tools/ubench/mfma_f16_chain > $O/mfma_f16_chain.json 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06f/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("power"), {k: (v.get("images_per_s") if isinstance(v, dict) else v) for k, v in d.get("configs", {}).items()})
PY
