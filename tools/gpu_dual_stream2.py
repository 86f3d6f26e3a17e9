"""Experiment: two engines on two streams, each with half the batch, stream 1 started half a forward late (so that one stream walks
the full-resolution levels while the other is in the deep ones).   python tools/gpu_dual_stream2.py [dim] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import pnpflow_oracle as O
from pnpflow_amd.models import UNet
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cfg = O.unet_config(3, dim, 32, (1, 2, 4, 8), 6, (16, 8))
sd = O.synthetic_state_dict(cfg, 0)
def mk():
    m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(sd); return m
m0 = mk(); x = torch.randn(B, 3, dim, dim).cuda(); t = torch.full((B,), 0.37).cuda()
m0(x, t); torch.cuda.synchronize()
n = 6
t0 = time.time()
for _ in range(n): m0(x, t)
torch.cuda.synchronize(); base = (time.time() - t0) / n * 1e3
print(f"dim={dim} B={B} one stream: {base:.2f} ms per full-batch forward", flush=True)
del m0
ms = [mk(), mk()]; streams = [torch.cuda.Stream(), torch.cuda.Stream()]
h = B // 2
xs = [x[:h].contiguous(), x[h:].contiguous()]; ts = [t[:h].contiguous(), t[h:].contiguous()]
delay = mk(); xd = x[:max(h // 2, 1)].contiguous(); td = t[:max(h // 2, 1)].contiguous()
for mm, xx, tt in zip(ms, xs, ts): mm(xx, tt)
delay(xd, td); torch.cuda.synchronize()
for stagger in (False, True):
    torch.cuda.synchronize(); t0 = time.time()
    if stagger:
        with torch.cuda.stream(streams[1]): delay(xd, td)          # ~half of a half-batch forward
    for _ in range(n):
        for mm, st, xx, tt in zip(ms, streams, xs, ts):
            with torch.cuda.stream(st): mm(xx, tt)
    torch.cuda.synchronize(); el = (time.time() - t0) / n * 1e3
    print(f"dim={dim} B={B} two streams{' staggered (incl. the delay forward, 1/' + str(4 * n) + ' extra work)' if stagger else ''}: {el:.2f} ms per full-batch forward ({100 * (el / base - 1):+.1f} %)", flush=True)
