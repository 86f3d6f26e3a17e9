# same-box A/B of two builds of the library (box-to-box variation is +-2 %):  bash tools/ab.sh <libA.so> <libB.so> [dim] [B] [reps]
A=$1; B=$2; DIM=${3:-128}; NB=${4:-160}; R=${5:-3}
for i in $(seq $R); do
  echo -n "A "; PNPFLOW_HIP_LIB=$A python tools/gpu_forward_only.py $DIM $NB 6 | tail -1
  echo -n "B "; PNPFLOW_HIP_LIB=$B python tools/gpu_forward_only.py $DIM $NB 6 | tail -1
done
