"""GPU-box helper (round 3): U-Net forward with the LDS-DMA conv path on / off / forced, outputs saved for comparison.
    PNPFLOW_HIP_DMA={0,1,2} python tools/gpu_dma_check.py run <net> <B> <prec> <out.npy>
    python tools/gpu_dma_check.py cmp <ref.npy> <test.npy> <tol>
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

NETS = {
    "celeba128": dict(dim=128, ch=32, ch_mult=(1, 2, 4, 8), nrb=6, attn=(16, 8)),
    "afhq256": dict(dim=256, ch=32, ch_mult=(1, 2, 4, 8), nrb=6, attn=(16, 8)),
}

if sys.argv[1] == "cmp":
    a, b, tol = np.load(sys.argv[2]), np.load(sys.argv[3]), float(sys.argv[4])
    err = float(np.abs(a - b).max()); ref = float(np.abs(a).max())
    ok = np.isfinite(b).all() and err <= tol * max(ref, 1e-30)
    print(f"cmp {os.path.basename(sys.argv[3]):40s} max|diff| {err:.3e}  max|ref| {ref:.3e}  rel {err / max(ref, 1e-30):.3e}  {'OK' if ok else 'FAIL'}")
    sys.exit(0 if ok else 1)

import torch
from tools.synthetic_weights import synthetic_state_dict      # product-side recipe: these drivers produce judged measurements and do not touch oracle/
from pnpflow_amd.models import UNet
net, B, prec, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
c = NETS[net]
m = UNet(3, c["dim"], c["ch"], ch_mult=c["ch_mult"], num_res_blocks=c["nrb"], attn_resolutions=c["attn"])
m.load_state_dict(synthetic_state_dict(m, 0)); m.set_precision(prec)
g = torch.Generator().manual_seed(7)
x = torch.randn(B, 3, c["dim"], c["dim"], generator=g).cuda(); t = torch.linspace(0.05, 0.95, B).cuda()
v = m(x, t); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3): v = m(x, t)
torch.cuda.synchronize()
print(f"{net} B={B} prec={prec} DMA={os.environ.get('PNPFLOW_HIP_DMA', '1')}: forward {(time.time() - t0) / 3 * 1e3:.2f} ms")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.save(out, v.float().cpu().numpy())
