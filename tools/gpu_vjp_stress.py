"""GPU-box stress: the same retained forward + backward repeated N times; every result is compared with the first one.
   Anything but round-off-level differences (the fp64 statistics atomics reorder) is a race or a read of memory that was never written.
   python tools/gpu_vjp_stress.py [dim] [B] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pnpflow_amd.models import UNet
from tools.synthetic_weights import synthetic_state_dict
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(synthetic_state_dict(m, 0))
g = torch.Generator(device="cpu"); g.manual_seed(1)
x = torch.randn(B, 3, dim, dim, generator=g).cuda(); t = torch.full((B,), 0.1).cuda(); vec = torch.randn(B, 3, dim, dim, generator=g).cuda()
v0, g0 = m.vjp(x, t, vec); torch.cuda.synchronize()
sv, sg = v0.abs().max().item(), g0.abs().max().item()
bad = 0
t0 = time.time()
for i in range(n):
    v, gg = m.vjp(x, t, vec)
    dv = (v - v0).abs().max().item(); dg = (gg - g0).abs().max().item()
    if not (dv <= 1e-4 * sv and dg <= 1e-4 * sg):
        bad += 1
        if bad <= 5:
            nb = (gg - g0).abs().amax(dim=(1, 2, 3))
            print(f"iteration {i}: forward diff {dv:.3e} (|v| {sv:.2e}), backward diff {dg:.3e} (|g| {sg:.2e}); per image {[f'{q:.1e}' for q in nb.tolist()][:8]} nan_v={bool(torch.isnan(v).any())} nan_g={bool(torch.isnan(gg).any())}")
print(f"stress dim={dim} B={B}: {n} repeats, {bad} deviating, {(time.time() - t0) / n * 1e3:.1f} ms each")
