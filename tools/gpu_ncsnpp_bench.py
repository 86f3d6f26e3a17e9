"""GPU box: forward throughput of the NCSN++ net at the reference's rectified-flow config.   python tools/gpu_ncsnpp_bench.py [B] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import ncsnpp_oracle as NO
from pnpflow_amd.image_generation.configs.rectified_flow.afhq_cat_pytorch_rf_gaussian import get_config
from pnpflow_amd.image_generation.models.ncsnpp import NCSNpp

FLOP_PER_IMAGE = 0.5318e12      # convs + attention matmuls of one forward at 256^2 (counted from the module list)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    m = NCSNpp(get_config()); m.load_state_dict(NO.synthetic_state_dict(NO.ncsnpp_config(), 0))
    x = torch.randn(B, 3, 256, 256, device="cuda"); lab = torch.full((B,), 400.0, device="cuda")
    m(x, lab); torch.cuda.synchronize(); m.check_numerics()
    t0 = time.time()
    for _ in range(reps):
        m(x, lab)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(f"B={B}: {dt * 1e3:.1f} ms per forward, {dt / B * 1e3:.2f} ms per image, {FLOP_PER_IMAGE * B / dt / 1e12:.0f} algorithmic TFLOP/s, "
          f"{m.memory_bytes() / 2**30:.1f} GiB held")
