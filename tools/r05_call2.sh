# round-5 call 2: the selection / production-batch tests that changed, conv_sp32 against conv_pp per launch class, SQ / TA / TCP counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "production or pp64_path or conv_sp_and or sp32_is or c5_all or conv_pp_path" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for sp in 0 1 0 1; do
  PNPFLOW_HIP_SP32=$sp timeout 300 python tools/gpu_layer_profile.py 256 160 $O/l256_160_sp32_$sp.csv > /dev/null 2>&1
  echo "== 256^2, B = 160, PNPFLOW_HIP_SP32=$sp"
  python tools/layer_summary.py $O/l256_160_sp32_$sp.csv | grep -E "total|Cout=  32 K=.* s=1 up=0"
done
for sp in 0 1; do
  PNPFLOW_HIP_SP32=$sp timeout 300 python tools/gpu_layer_profile.py 128 160 $O/l128_160_sp32_$sp.csv > /dev/null 2>&1
  echo "== 128^2, B = 160, PNPFLOW_HIP_SP32=$sp"
  python tools/layer_summary.py $O/l128_160_sp32_$sp.csv | grep -E "total|Cout=  32 K=.* s=1 up=0"
done
bash tools/pmc_level0.sh > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/pmc_level0 > $O/pmc_conv_counters.md 2>&1; head -12 $O/pmc_conv_counters.md | cut -c1-230
