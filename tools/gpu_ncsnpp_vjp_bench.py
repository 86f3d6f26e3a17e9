"""GPU box: retained forward + input-gradient backward of the NCSN++ net at the reference's rectified-flow config.
   python tools/gpu_ncsnpp_vjp_bench.py [B] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnpflow_amd.image_generation.configs.rectified_flow.afhq_cat_pytorch_rf_gaussian import get_config
from pnpflow_amd.image_generation.models.ncsnpp import NCSNpp
from tools.synthetic_weights import synthetic_state_dict

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    m = NCSNpp(get_config()); m.load_state_dict(synthetic_state_dict(m))
    x = torch.randn(B, 3, 256, 256, device="cuda"); lab = torch.full((B,), 400.0, device="cuda"); vec = torch.randn_like(x)
    m.vjp(x, lab, vec); torch.cuda.synchronize(); m.check_numerics()
    t0 = time.time()
    for _ in range(reps):
        m.forward_retain(x, lab)
    torch.cuda.synchronize(); tf = (time.time() - t0) / reps
    t0 = time.time()
    for _ in range(reps):
        m.backward(vec)
    torch.cuda.synchronize(); tb = (time.time() - t0) / reps
    print(f"B={B}: retained forward {tf * 1e3:.1f} ms, backward {tb * 1e3:.1f} ms, {m.memory_bytes() / 2**30:.1f} GiB held")
