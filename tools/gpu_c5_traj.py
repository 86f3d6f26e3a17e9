import os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from pnpflow_amd import _lib
dev = torch.device("cuda", 0)
r = bench.Runner("c5", 0, 1, dev, 1, True, {})
for rep in range(3):
    mx = {}
    def cb(it, x):
        mx[it] = float(x.abs().max())
    try:
        x = r.solver.restore_batch(r.y, r.degradation, r.sigma, iter_cb=cb, cb_iterations=list(range(0, 100, 8)) + [97, 98, 99])
        print("rep", rep, "ok", {k: f"{v:.3g}" for k, v in sorted(mx.items())})
    except _lib.PnpFlowHipError as e:
        print("rep", rep, "FAILED", {k: f"{v:.3g}" for k, v in sorted(mx.items())}, str(e)[:120])
