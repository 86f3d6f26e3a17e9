"""Per-launch timing of the conv-GEMM launches of one forward (HIP events) -> CSV."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.synthetic_weights import synthetic_state_dict      # product-side recipe: these drivers produce judged measurements and do not touch oracle/
from pnpflow_amd.models import UNet
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/layers.csv"
m = UNet(3, dim, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)); m.load_state_dict(synthetic_state_dict(m, 0))
m.set_precision(int(os.environ.get('PNPFLOW_PREC', '1')))
x = torch.randn(B, 3, dim, dim).cuda(); t = torch.full((B,), 0.37).cuda()
m(x, t); m(x, t); torch.cuda.synchronize()
os.environ["PNPFLOW_HIP_PROFILE_CSV"] = out
m.profile(True); m(x, t); print(m.profile_read()); m.profile(False)
