"""GPU-box diagnostic: wall time of restore_batch, eager vs hipGraph.
    python tools/gpu_time_restore.py [dim] [B] [steps] [ns]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import pnpflow_oracle as O
import pnpflow_amd.degradations as D
from pnpflow_amd.methods.pnp_flow import PNP_FLOW
from pnpflow_amd.models import UNet
from pnpflow_amd.utils import CfgNode

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 5
cfg = O.unet_config(3, dim, 32, (1, 2, 4, 8), 6, (16, 8))
t0 = time.time(); sd = O.synthetic_state_dict(cfg, 0); print("weights", time.time() - t0, flush=True)
t0 = time.time(); m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(sd)
print("load", time.time() - t0, flush=True)
y = torch.randn(B, 3, dim, dim).cuda()
for use_graph in (False, True, True):
    args = CfgNode(dict(method="pnp_flow", model="ot", problem="inpainting", noise_type="gaussian", num_samples=ns, steps_pnp=steps,
                        lr_pnp=1.0, gamma_style="alpha_1_minus_t", alpha=0.5, max_batch=1, compute_time=False, compute_memory=False,
                        save_results=False, batch=0, sigma_noise=0.05))
    s = PNP_FLOW(m, torch.device("cuda"), args); s.use_graph = use_graph
    torch.cuda.synchronize(); t0 = time.time()
    x = s.restore_batch(y, D.BoxInpainting(dim // 6), 0.05, 0.05 ** 2)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"graph={use_graph} steps={steps} ns={ns}: {dt:.3f}s  -> {dt / (steps * ns) * 1e3:.2f} ms per forward+pointwise", flush=True)
