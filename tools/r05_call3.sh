# round-5 call 3 (experiments): conv_sp start stagger, conv_sp tiles-per-workgroup threshold
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05c; mkdir -p $O
for st in 0 50 100 0 100 200; do
  PNPFLOW_SP_STAGGER=$st timeout 300 python tools/gpu_layer_profile.py 256 160 $O/l256_160_st$st.csv > /dev/null 2>&1
  echo "== 256^2, B = 160, PNPFLOW_SP_STAGGER=$st"
  python tools/layer_summary.py $O/l256_160_st$st.csv | grep -E "total|Cout= 128 K=.* s=1 up=0|Cout=  64 K=.* s=1 up=0" | head -12
done
for shape in "128 160" "256 32" "128 320"; do
  set -- $shape
  for mt in 4 2 1; do
    PNPFLOW_SP_MINT=$mt timeout 300 python tools/gpu_layer_profile.py $1 $2 $O/l$1_$2_mint$mt.csv > /dev/null 2>&1
    echo "== $1^2, B = $2, PNPFLOW_SP_MINT=$mt"
    python tools/layer_summary.py $O/l$1_$2_mint$mt.csv | grep -E "total|Cout= 128 K=.* s=1 up=0|Cout=  64 K=.* s=1 up=0" | head -8
  done
done
