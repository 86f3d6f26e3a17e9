"""LPIPS (AlexNet, v0.1) of the reference's logging (pnpflow/utils.py:677-724) on the HIP engine (csrc/lpips.hip).

    m = LPIPS(); m.load_state_dict(sd)            # sd: lpips.LPIPS(net='alex').state_dict(), or torchvision alexnet + lpips lin weights
    d = m(img0, img1, normalize=True)             # (B,) per-pair distances, as lpips.LPIPS.forward(...).flatten()

Weight sources (none is reachable offline; `find_weights` looks where the two packages put them):
    torchvision:  ~/.cache/torch/hub/checkpoints/alexnet-owt-*.pth         keys  features.{0,3,6,8,10}.{weight,bias}
    lpips:        <site-packages>/lpips/weights/v0.1/alex.pth               keys  lin{0..4}.model.1.weight  ([1, C, 1, 1])
A full `lpips.LPIPS` state_dict (keys net.slice{1..5}.{0,3,6,8,10}.*, lin{k}.model.1.weight, lins.{k}.model.1.weight,
scaling_layer.*) is accepted too.  PARITY UNPINNED: neither package is installed in the build image (oracle: O.lpips_forward).
"""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np
import torch

from . import _lib

CONV_IDX = (0, 3, 6, 8, 10)
CHANNELS = (64, 192, 384, 256, 256)


def canonical_state_dict(sd) -> dict:
    """Maps the published key names onto the engine's: features.N.{weight,bias}, lin{k}."""
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        parts = k.split(".")
        if parts[0] == "net" and parts[1].startswith("slice"):        # lpips.LPIPS: net.slice3.6.weight
            k = "features." + ".".join(parts[2:])
        if k.startswith("features.") and int(k.split(".")[1]) in CONV_IDX:
            out[k] = v
        elif parts[0].startswith("lin") and parts[0][3:].isdigit() and k.endswith("model.1.weight"):
            out[parts[0]] = v.reshape(-1)
        elif parts[0] == "lins" and k.endswith("model.1.weight"):
            out["lin" + parts[1]] = v.reshape(-1)
    return out


def find_weights():
    """(alexnet checkpoint path, lpips linear-layer checkpoint path) or (None, None)."""
    roots = [os.environ.get("PNPFLOW_LPIPS_DIR", ""), os.path.expanduser("~/.cache/torch/hub/checkpoints")]
    alex = next((p for r in roots if r for p in sorted(glob.glob(os.path.join(r, "alexnet-owt-*.pth")))), None)
    lin = next((p for r in roots if r for p in sorted(glob.glob(os.path.join(r, "alex.pth")))), None)
    if lin is None:
        try:
            import importlib.util
            spec = importlib.util.find_spec("lpips")
            if spec and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "weights", "v0.1", "alex.pth")
                lin = cand if os.path.isfile(cand) else None
        except Exception:       # noqa: BLE001
            lin = None
    return alex, lin


class LPIPS:
    def __init__(self, net: str = "alex", device_index: int = 0):
        if net != "alex":
            raise NotImplementedError("the reference uses net='alex' (utils.py:685)")
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.pf_lpips_create(device_index, C.byref(h)), None, "pf_lpips_create")
        self.handle = h
        self.loaded = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pf_lpips_destroy(self.handle); self.handle = None
        except Exception:       # noqa: BLE001
            pass

    def _err(self):
        return (self.lib.pf_lpips_last_error(self.handle) or b"").decode()

    def load_state_dict(self, sd):
        sd = canonical_state_dict(sd)
        need = [f"features.{i}.{s}" for i in CONV_IDX for s in ("weight", "bias")] + [f"lin{k}" for k in range(5)]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"LPIPS weights missing: {missing}")
        for k in need:
            a = np.ascontiguousarray(sd[k].detach().cpu().float().numpy())
            shape = (C.c_int64 * a.ndim)(*a.shape)
            rc = self.lib.pf_lpips_load_weight(self.handle, k.encode(), a.ctypes.data, shape, a.ndim)
            if rc != 0:
                raise _lib.PnpFlowHipError(f"pf_lpips_load_weight({k}): {self._err()}")
        self.loaded = True
        return self

    @classmethod
    def from_files(cls, alexnet_path, lin_path, device_index=0):
        sd = dict(torch.load(alexnet_path, map_location="cpu"))
        sd.update(torch.load(lin_path, map_location="cpu"))
        return cls("alex", device_index).load_state_dict(sd)

    def __call__(self, in0: torch.Tensor, in1: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        if not in0.is_cuda:
            raise _lib.PnpFlowHipError("LPIPS needs GPU tensors (there is no CPU path)")
        a = in0.contiguous().float(); b = in1.to(a.device).contiguous().float()
        if b.shape != a.shape:
            raise ValueError("LPIPS expects two tensors of one shape")
        if a.shape[1] == 1:          # lpips' ScalingLayer broadcasts a 1-channel image over its three channel constants: grayscale inputs are legal there
            a = a.expand(-1, 3, -1, -1).contiguous(); b = b.expand(-1, 3, -1, -1).contiguous()
        B, Cc, H, W = a.shape
        if Cc != 3:
            raise ValueError("LPIPS expects (B, 3, H, W) or (B, 1, H, W) tensors")
        out = torch.empty(B, dtype=torch.float32, device=a.device)
        if B == 0:
            return out
        rc = self.lib.pf_lpips_forward(self.handle, a.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, 1 if normalize else 0, _lib.current_stream_ptr())
        if rc != 0:
            raise _lib.PnpFlowHipError(f"pf_lpips_forward: {self._err()}")
        return out
