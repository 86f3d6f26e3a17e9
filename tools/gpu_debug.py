"""GPU-box debugging aid: U-Net forward vs the oracle with per-tap error localisation.
    PNPFLOW_HIP_KEEP_ACTIVATIONS=1 python tools/gpu_debug.py [mnist|tiny4|celeba128|afhq256] [B]
"""
import os
import sys
import time

os.environ.setdefault("PNPFLOW_HIP_KEEP_ACTIVATIONS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import CFGS, det_normal
from oracle import pnpflow_oracle as O
from pnpflow_amd.models import UNet


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mnist"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    c = CFGS[name]
    cfg = O.unet_config(**c)
    sd = O.synthetic_state_dict(cfg, 0)
    m = UNet(c["input_channels"], c["input_height"], c["ch"], ch_mult=c["ch_mult"], num_res_blocks=c["num_res_blocks"],
             attn_resolutions=c["attn_resolutions"])
    m.load_state_dict(sd)
    m.set_precision(int(os.environ.get('PNPFLOW_PREC', '1')))
    x = det_normal((B, c["input_channels"], c["input_height"], c["input_height"]), 11)
    t = torch.tensor([0.0, 0.37, 0.99, 0.5][:B] if B <= 4 else [0.37] * B, dtype=torch.float32)
    taps_ref = {}
    t0 = time.time()
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, t, taps_ref)
    t_cpu = time.time() - t0
    xd, td = x.cuda(), t.cuda()
    v = m(xd, td)
    torch.cuda.synchronize()
    taps = m.read_taps(B)
    for k, a in taps.items():
        if k in taps_ref:
            r = taps_ref[k].numpy()
            print(f"{k:14s} max|err| {np.abs(a - r).max():.3e}   ref absmax {np.abs(r).max():.3e}  nan={np.isnan(a).any()}")
    out = v.cpu()
    print(f"OUTPUT max|err| {float((out - ref).abs().max()):.3e}  ref absmean {float(ref.abs().mean()):.3e}  cpu {t_cpu:.2f}s")
    # timing
    for _ in range(2):
        m(xd, td)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n):
        m(xd, td)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"forward B={B}: {dt * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
