"""Experiment: split the batch over S independent streams (one engine each) and interleave."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import pnpflow_oracle as O
from pnpflow_amd.models import UNet
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = O.unet_config(3, dim, 32, (1, 2, 4, 8), 6, (16, 8))
sd = O.synthetic_state_dict(cfg, 0)
for S in (1, 2, 4):
    ms = [UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)) for _ in range(S)]
    for m in ms: m.load_state_dict(sd)
    streams = [torch.cuda.Stream() for _ in range(S)]
    xs = [torch.randn(B // S, 3, dim, dim).cuda() for _ in range(S)]
    ts = [torch.full((B // S,), 0.37).cuda() for _ in range(S)]
    def run(n):
        for _ in range(n):
            for m, st, x, t in zip(ms, streams, xs, ts):
                with torch.cuda.stream(st):
                    m(x, t)
    run(1); torch.cuda.synchronize()
    t0 = time.time(); n = 4; run(n); torch.cuda.synchronize()
    print(f"dim={dim} B={B} streams={S}: {(time.time() - t0) / n * 1e3:.2f} ms per full-batch forward", flush=True)
    del ms
