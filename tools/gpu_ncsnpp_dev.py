"""Development aid (GPU box): NCSN++ engine vs the CPU oracle, per tap.   python tools/gpu_ncsnpp_dev.py [tiny|wide|afhq256] [B]"""
import os
import sys
import time
import types

os.environ.setdefault("PNPFLOW_HIP_KEEP_ACTIVATIONS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import ncsnpp_oracle as NO
from pnpflow_amd.image_generation.models.ncsnpp import NCSNpp

CFGS = {"tiny": dict(image_size=32, nf=32, ch_mult=(1, 1, 2), num_res_blocks=2, attn_resolutions=(16,)),
        "wide": dict(image_size=32, nf=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,)),
        "afhq256": dict(image_size=256, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))}


def make_config(c):
    NS = types.SimpleNamespace
    return NS(model=NS(name="ncsnpp", nf=c["nf"], ch_mult=c["ch_mult"], num_res_blocks=c["num_res_blocks"], attn_resolutions=c["attn_resolutions"],
                       dropout=0., conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan",
                       progressive="output_skip", progressive_input="input_skip", progressive_combine="sum", embedding_type="fourier",
                       nonlinearity="swish", scale_by_sigma=True),
              data=NS(image_size=c["image_size"], num_channels=3, centered=True), training=NS(continuous=False, sde="rectified_flow"))


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    c = CFGS[name]
    cfg = NO.ncsnpp_config(**c)
    sd = NO.synthetic_state_dict(cfg, 0)
    g = np.random.Generator(np.random.Philox(key=[81, 0]))
    x = torch.from_numpy(g.standard_normal(size=(B, 3, c["image_size"], c["image_size"]), dtype=np.float32))
    t = torch.tensor([0.37, 0.81, 0.05, 0.99][:B])
    taps = {}
    t0 = time.time(); ref = NO.ncsnpp_forward(sd, cfg, x, t * 999, taps); print("oracle %.2fs" % (time.time() - t0))
    m = NCSNpp(make_config(c)); m.load_state_dict(sd)
    for prec in (1, 0):
        m.set_precision(prec)
        y = m(x.cuda(), (t * 999).cuda()); torch.cuda.synchronize()
        m.check_numerics()
        err = (y.cpu() - ref).abs().max().item()
        print(f"precision {prec}: max|hip - oracle| = {err:.3e}  (|ref|max {ref.abs().max().item():.3e}, rel {err / ref.abs().max().item():.2e})")
        if prec == 1:
            got = m.read_taps(B)
            for k, v in got.items():
                if k in taps:
                    r = taps[k].numpy()
                    print(f"   tap {k:14s} {tuple(r.shape)} err {np.abs(v - r).max():.3e} |ref| {np.abs(r).max():.3e}")
    vec = torch.from_numpy(np.random.Generator(np.random.Philox(key=[87, 0])).standard_normal(size=tuple(x.shape), dtype=np.float32))
    t0 = time.time(); gref = NO.ncsnpp_vjp(sd, cfg, x, t * 999, vec); print("oracle vjp %.2fs" % (time.time() - t0))
    for prec in (1, 0):
        m.set_precision(prec)
        v, gg = m.vjp(x.cuda(), (t * 999).cuda(), vec.cuda()); torch.cuda.synchronize(); m.check_numerics()
        e1 = (v.cpu() - ref).abs().max().item(); e2 = (gg.cpu() - gref).abs().max().item()
        print(f"precision {prec}: retained forward err {e1:.3e}; vjp max|hip - oracle| = {e2:.3e} (|ref|max {gref.abs().max().item():.3e}, rel {e2 / gref.abs().max().item():.2e})")
    if name == "afhq256":
        m.set_precision(1)
        for bb in (1, 4, 8):
            xx = torch.randn(bb, 3, 256, 256, device="cuda"); tt = torch.full((bb,), 400.0, device="cuda")
            m(xx, tt); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                m(xx, tt)
            torch.cuda.synchronize()
            print(f"B={bb}: {(time.time() - t0) / 3 * 1e3:.1f} ms per forward, {m.memory_bytes() / 2**30:.2f} GiB")
