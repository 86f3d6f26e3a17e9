"""`UNet` with the constructor / call / state_dict surface of pnpflow.models.UNet
(reference pnpflow/models.py:302-495), executed by the HIP engine.

    model = UNet(3, 128, 32, ch_mult=(1,2,4,8), num_res_blocks=6, attn_resolutions=(16,8))
    model.load_state_dict(torch.load(".../model_final.pt"))     # the reference's checkpoint format
    v = model(x, t)            # x: (B,C,H,W) fp32 on the GPU, t: (B,) -> (B,C,H,W)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import _lib


class _EngineNet:
    """What UNet and NCSNpp share: the torch.nn.Module-like surface the reference's callers use, on a pf_engine handle."""

    # ---- torch.nn.Module-like surface used by the reference's callers -------------------
    def to(self, device=None):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def state_dict_keys(self):
        n = self._lib.pf_engine_num_weights(self._h)
        return [self._lib.pf_engine_weight_name(self._h, i).decode() for i in range(n)]

    def state_dict_shapes(self):
        out = {}
        for i, k in enumerate(self.state_dict_keys()):
            shp = (C.c_int64 * 4)()
            nd = self._lib.pf_engine_weight_shape(self._h, i, shp)
            out[k] = tuple(int(shp[j]) for j in range(nd))
        return out

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        keys = self.state_dict_keys()
        missing = [k for k in keys if k not in state_dict]
        unexpected = [k for k in state_dict if k not in set(keys)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for k in keys:
            if k not in state_dict:
                continue
            a = np.ascontiguousarray(state_dict[k].detach().cpu().to(torch.float32).numpy())
            shp = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self._lib.pf_engine_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shp, a.ndim),
                       self._h, f"pf_engine_load_weight({k})")
        _lib.check(self._lib.pf_engine_finalize_weights(self._h), self._h, "pf_engine_finalize_weights")
        self._loaded = True
        return self

    @property
    def handle(self):
        return self._h

    def __call__(self, x: torch.Tensor, temp: torch.Tensor) -> torch.Tensor:
        return self.forward(x, temp)

    def forward(self, x: torch.Tensor, temp: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise _lib.PnpFlowHipError(f"{type(self).__name__}.forward needs tensors on the GPU (there is no CPU path)")
        x = x.contiguous().float()
        t = temp.contiguous().float().to(x.device)
        B, Cc, H, W = x.shape
        assert Cc == self.input_channels and H == self.input_height and W == H and t.shape == (B,)
        v = torch.empty((B, self.output_channels, H, W), dtype=torch.float32, device=x.device)
        _lib.check(self._lib.pf_unet_forward(self._h, x.data_ptr(), t.data_ptr(), v.data_ptr(), B, _lib.current_stream_ptr()),
                   self._h, "pf_unet_forward")
        return v

    def forward_retain(self, x: torch.Tensor, temp: torch.Tensor) -> torch.Tensor:
        """Forward that keeps every activation on the device, for a following `backward`."""
        x = x.contiguous().float(); t = temp.contiguous().float().to(x.device)
        B = x.shape[0]
        v = torch.empty((B, self.output_channels, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        _lib.check(self._lib.pf_unet_forward_retain(self._h, x.data_ptr(), t.data_ptr(), v.data_ptr(), B, _lib.current_stream_ptr()),
                   self._h, "pf_unet_forward_retain")
        return v

    def backward(self, vec: torch.Tensor) -> torch.Tensor:
        """J^T vec (input gradient) at the last `forward_retain`."""
        vec = vec.contiguous().float()
        g = torch.empty((vec.shape[0], self.input_channels, vec.shape[2], vec.shape[3]), dtype=torch.float32, device=vec.device)
        _lib.check(self._lib.pf_unet_backward(self._h, vec.data_ptr(), g.data_ptr(), vec.shape[0], _lib.current_stream_ptr()),
                   self._h, "pf_unet_backward")
        return g

    def vjp(self, x: torch.Tensor, temp: torch.Tensor, vec: torch.Tensor):
        """(v_theta(x,t), J^T vec) - what torch.autograd.functional.vjp(lambda z: model(z,t), x, vec) returns
        (reference pnpflow/methods/ot_ode.py:137-138)."""
        v = self.forward_retain(x, temp)
        return v, self.backward(vec)

    # ---- debugging: named internal activations of the last forward (NCHW numpy) -----------
    def read_taps(self, B):
        """Internal activations of the last forward as {name: (B,C,H,W) numpy}.  Only
        meaningful when PNPFLOW_HIP_KEEP_ACTIVATIONS=1 was set before the first forward
        (otherwise activation buffers are recycled)."""
        out = {}
        big = np.empty(B * self.ch * max(self.ch_mult) * self.input_height * self.input_height, dtype=np.float32)
        for i in range(self._lib.pf_engine_num_taps(self._h)):
            name = self._lib.pf_engine_tap_name(self._h, i).decode()
            dims = (C.c_int32 * 3)()
            _lib.check(self._lib.pf_engine_read_tap(self._h, i, big.ctypes.data_as(C.c_void_p), big.size, dims, _lib.current_stream_ptr()),
                       self._h, "pf_engine_read_tap")
            c, h, w = dims[0], dims[1], dims[2]
            out[name] = big[: B * c * h * w].reshape(B, c, h, w).copy()
        return out

    def set_precision(self, mode: int):
        """1 (default): split-fp16 MFMA (fp32-equivalent, 3 x f16 MFMA per product); 0: exact fp32 MFMA; 2: single fp16 MFMA per
        product (fp16 operands, fp32 accumulate: TF32-class, ~1e-3 relative on the U-Net output)."""
        _lib.check(self._lib.pf_engine_set_precision(self._h, int(mode)), self._h, "pf_engine_set_precision")
        return self

    def check_numerics(self):
        """Synchronises the current stream and raises if any forward since the last check saw a non-finite activation."""
        _lib.check(self._lib.pf_engine_check_numerics(self._h, _lib.current_stream_ptr()), self._h, "pf_engine_check_numerics")

    def memory_bytes(self) -> int:
        """Device bytes the engine currently holds (weights, activation plans, solver buffers)."""
        return int(self._lib.pf_engine_memory_bytes(self._h))

    def profile(self, enable: bool):
        _lib.check(self._lib.pf_engine_profile(self._h, 1 if enable else 0), self._h, "pf_engine_profile")

    def profile_read(self):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self._lib.pf_engine_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(fl)), self._h, "pf_engine_profile_read")
        return n.value, ms.value, fl.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass


class UNet(_EngineNet):
    def __init__(self, input_channels, input_height, ch, output_channels=None, ch_mult=(1, 2, 4, 8),
                 num_res_blocks=2, attn_resolutions=(16,), dropout=0., resamp_with_conv=True, act=None,
                 normalize=None, device_index: int = 0):
        if dropout != 0. or not resamp_with_conv:
            raise NotImplementedError("engine implements the configuration define_model uses: dropout=0, resamp_with_conv=True")
        self.input_channels = input_channels
        self.input_height = input_height
        self.ch = ch
        self.output_channels = input_channels if output_channels is None else output_channels
        self.ch_mult = tuple(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.num_resolutions = len(self.ch_mult)
        self._lib = _lib.load()
        cfg = _lib.PfUnetCfg()
        cfg.input_channels, cfg.output_channels, cfg.input_height, cfg.ch = input_channels, self.output_channels, input_height, ch
        cfg.num_levels, cfg.num_res_blocks = len(self.ch_mult), num_res_blocks
        for i, m in enumerate(self.ch_mult):
            cfg.ch_mult[i] = m
        cfg.num_attn_resolutions = len(self.attn_resolutions)
        for i, r in enumerate(self.attn_resolutions):
            cfg.attn_resolutions[i] = r
        self._device_index = device_index
        h = C.c_void_p()
        _lib.check(self._lib.pf_engine_create(device_index, C.byref(cfg), C.byref(h)), None, "pf_engine_create")
        self._h = h
        self._loaded = False
        self.training = False
