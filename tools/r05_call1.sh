# round-5 status run on the GPU box: gpu tests, per-launch layer profile, kernel traces, short bench.   bash tools/r05_call1.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for s in "256 160" "128 160" "256 32"; do
  set -- $s
  timeout 300 python tools/gpu_layer_profile.py $1 $2 $O/l$1_$2.csv > /dev/null 2>&1
  python tools/layer_summary.py $O/l$1_$2.csv > $O/l$1_$2.txt; head -1 $O/l$1_$2.txt
done
WLS="c2_256 c5" NO_PMC=1 bash tools/profile_round.sh r05a > /dev/null 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05a/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("configs", {}).items()})
print(json.dumps(d["roofline"]["classes"], indent=0)[:3000])
PY
