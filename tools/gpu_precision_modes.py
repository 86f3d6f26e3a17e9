"""GPU box: U-Net forward error vs the CPU oracle and forward time in the three precision modes.  python tools/gpu_precision_modes.py [dim] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import pnpflow_oracle as O
from pnpflow_amd.models import UNet
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cfg = O.unet_config(3, dim, 32, (1, 2, 4, 8), 6, (16, 8))
sd = O.synthetic_state_dict(cfg, 0)
m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(sd)
g = np.random.Generator(np.random.Philox(key=[5, 0]))
x1 = torch.from_numpy(g.standard_normal(size=(2, 3, dim, dim), dtype=np.float32)); t1 = torch.tensor([0.3, 0.8])
ref = O.unet_forward(sd, cfg, x1, t1)
x = torch.randn(B, 3, dim, dim).cuda(); t = torch.full((B,), 0.37).cuda()
for prec in (1, 2, 0):
    m.set_precision(prec)
    y = m(x1.cuda(), t1.cuda()).cpu()
    err = (y - ref).abs().max().item(); rel = ((y - ref).norm() / ref.norm()).item()
    m(x, t); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        m(x, t)
    torch.cuda.synchronize()
    print(f"precision {prec}: max|hip - oracle| = {err:.3e} (|ref|max {ref.abs().max().item():.2f}), relative L2 {rel:.2e}; forward B={B}: {(time.time() - t0) / 3 * 1e3:.2f} ms")
