"""GPU-box helper: the `pointwise` block of bench.py on its own (HIP-event timing of the prox / data-fidelity kernels)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pnpflow_amd.degradations as D
torch.cuda.set_device(0)
out = bench.pointwise_block(torch.device("cuda", 0), D)
for k, v in out.items():
    print(f"{k:50s} {v['us']:8.2f} us  {v['achieved_gbs']:8.1f} GB/s")
