"""GPU box: which sequence of BASELINE workloads (sharing their engines, as bench.py does) trips over PNPFLOW_HIP_POISON
   (reads of never-written engine memory).   PNPFLOW_HIP_POISON=<mask> python tools/gpu_poison_probe.py c2:2 c3:2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pnpflow_amd import _lib
models = {}
for spec in sys.argv[1:]:
    roof = spec.endswith("+roof"); spec = spec.replace("+roof", "")
    name, steps = spec.split(":")
    r = bench.Runner(name, 0, 1, torch.device("cuda", 0), 1, os.environ.get("NOGRAPH") is None, models)
    try:
        x = r.step(0, steps=int(steps))
        print(name, "steps", steps, "ok finite =", bool(torch.isfinite(x).all()), flush=True)
        if roof:
            print("   roofline", bench.conv_roofline(r, 1, name)["achieved"], flush=True)
    except _lib.PnpFlowHipError as e:
        print(name, "steps", steps, "FAILED", str(e)[-160:], flush=True)
