"""GPU-box helper for profiling: N x (retained forward + input-gradient backward) of the BASELINE U-Net (the OT-ODE inner work).
    python tools/gpu_vjp_only.py [dim] [B] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnpflow_amd.models import UNet
from tools.synthetic_weights import synthetic_state_dict
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(synthetic_state_dict(m, 0))
x = torch.randn(B, 3, dim, dim).cuda(); t = torch.full((B,), 0.37).cuda(); vec = torch.randn(B, 3, dim, dim).cuda()
m.vjp(x, t, vec); torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    m.vjp(x, t, vec)
torch.cuda.synchronize()
print(f"vjp dim={dim} B={B}: {(time.time() - t0) / n * 1e3:.2f} ms")
if len(sys.argv) > 4:      # save (v, J^T vec) of a seeded input for A/B comparisons across environment switches (tests/test_gpu_parity.py)
    import numpy as np
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 3, dim, dim, generator=g).cuda(); t = torch.linspace(0.05, 0.95, B).cuda(); vec = torch.randn(B, 3, dim, dim, generator=g).cuda()
    v, jt = m.vjp(x, t, vec)
    np.savez(sys.argv[4], v=v.float().cpu().numpy(), g=jt.float().cpu().numpy())
