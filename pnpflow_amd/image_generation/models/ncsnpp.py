"""`NCSNpp` with the constructor / call / state_dict surface of pnpflow.image_generation.models.ncsnpp.NCSNpp (reference
ncsnpp.py:34-383), executed by the HIP engine (csrc/engine_ncsnpp.inc, ncsnpp_ops.hip).

    model = NCSNpp(config)                                   # config: configs/rectified_flow/*_rf_gaussian.get_config()
    model.load_state_dict(torch.load(path)["model"], strict=False)      # the reference's checkpoint (`module.` keys accepted)
    v = model(x, t * 999)                                    # methods/pnp_flow.py:23-27

Built: the block list both rectified-flow configs of the reference select (BigGAN blocks, FIR resampling, input_skip / output_skip
with `sum`, Fourier conditioning, skip_rescale, scale_by_sigma).  Any other value of those switches raises NotImplementedError.
`forward_retain` / `backward` / `vjp` give the input-gradient VJP OT_ODE needs (ot_ode.py:137-138).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from ... import _lib
from ...models import _EngineNet


def _get(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


class NCSNpp(_EngineNet):
    def __init__(self, config, device_index: int = 0):
        m, d, tr = _get(config, "model"), _get(config, "data"), _get(config, "training")
        want = dict(resblock_type="biggan", progressive="output_skip", progressive_input="input_skip", progressive_combine="sum",
                    embedding_type="fourier")
        for k, v in want.items():
            if str(_get(m, k, v)).lower() != v:
                raise NotImplementedError(f"NCSNpp engine: config.model.{k} must be '{v}' (the reference's rectified-flow configs)")
        if not _get(m, "conditional", True) or not _get(m, "fir", True) or _get(m, "dropout", 0.) != 0. or \
                str(_get(m, "nonlinearity", "swish")).lower() != "swish":
            raise NotImplementedError("NCSNpp engine: needs conditional=True, fir=True, dropout=0, nonlinearity='swish'")
        if not _get(d, "centered", False):
            raise NotImplementedError("NCSNpp engine: config.data.centered must be True (both rectified-flow configs set it)")
        if tr is not None and not (_get(tr, "continuous", False) or _get(tr, "sde", "") == "rectified_flow"):
            raise AssertionError("Fourier features are only used for continuous training.")      # ncsnpp.py:75
        self.config = config
        self.nf = self.ch = int(_get(m, "nf"))
        self.ch_mult = tuple(_get(m, "ch_mult"))
        self.num_res_blocks = int(_get(m, "num_res_blocks"))
        self.attn_resolutions = tuple(_get(m, "attn_resolutions"))
        self.num_resolutions = len(self.ch_mult)
        self.input_height = int(_get(d, "image_size"))
        self.input_channels = self.output_channels = int(_get(d, "num_channels", 3))
        self.skip_rescale = bool(_get(m, "skip_rescale", True))
        self.scale_by_sigma = bool(_get(m, "scale_by_sigma", False))
        fir_kernel = list(_get(m, "fir_kernel", [1, 3, 3, 1]))
        self._lib = _lib.load()
        cfg = _lib.PfNcsnppCfg()
        cfg.image_size, cfg.num_channels, cfg.nf, cfg.num_levels = self.input_height, self.input_channels, self.nf, len(self.ch_mult)
        for i, v in enumerate(self.ch_mult):
            cfg.ch_mult[i] = v
        cfg.num_res_blocks = self.num_res_blocks
        cfg.num_attn_resolutions = len(self.attn_resolutions)
        for i, r in enumerate(self.attn_resolutions):
            cfg.attn_resolutions[i] = r
        cfg.fir_taps = len(fir_kernel)
        for i, v in enumerate(fir_kernel):
            cfg.fir_kernel[i] = float(v)
        cfg.skip_rescale, cfg.scale_by_sigma, cfg.centered = int(self.skip_rescale), int(self.scale_by_sigma), 1
        self._device_index = device_index
        h = C.c_void_p()
        _lib.check(self._lib.pf_ncsnpp_create(device_index, C.byref(cfg), C.byref(h)), None, "pf_ncsnpp_create")
        self._h = h
        self._loaded = False
        self.training = False

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("inference engine")
        return self

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts NCSNpp's own keys and the reference checkpoint's DataParallel form (`module.all_modules...`, utils.py:7-13); the
        `sigmas` buffer (ncsnpp.py:42) is not an input of the Fourier-conditioned forward and is ignored."""
        sd = {}
        for k, v in state_dict.items():
            k = k[len("module."):] if k.startswith("module.") else k
            if k == "sigmas":
                continue
            sd[k] = v
        return super().load_state_dict(sd, strict=strict)

    def set_solver_time_scale(self, scale: float):
        """label = t * scale inside the engine's solver loops (PNP_FLOW.model_forward passes `t * 999`, methods/pnp_flow.py:23-27)"""
        _lib.check(self._lib.pf_engine_set_solver_time_scale(self._h, float(scale)), self._h, "pf_engine_set_solver_time_scale")
        return self
