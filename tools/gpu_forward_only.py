"""GPU-box helper for profiling: N forwards of the BASELINE U-Net, nothing else.
    python tools/gpu_forward_only.py [dim] [B] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.synthetic_weights import synthetic_state_dict      # product-side recipe: these drivers produce judged measurements and do not touch oracle/
from pnpflow_amd.models import UNet
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(synthetic_state_dict(m, 0)); m.set_precision(int(os.environ.get('PNPFLOW_PREC', '1')))
x = torch.randn(B, 3, dim, dim).cuda(); t = torch.full((B,), 0.37).cuda()
m(x, t); torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    m(x, t)
torch.cuda.synchronize()
print(f"forward dim={dim} B={B}: {(time.time() - t0) / n * 1e3:.2f} ms")
