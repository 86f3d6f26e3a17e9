"""CPU oracle for the NCSN++ ("rectified" coupling) velocity net  --  TEST INFRASTRUCTURE ONLY.

A from-scratch restatement (torch fp32 functional ops, no nn.Module, no reference import) of
pnpflow/image_generation/models/ncsnpp.py `NCSNpp.forward` for the block list the reference's two rectified-flow configs select
(configs/rectified_flow/{celeba_hq,afhq_cat}_pytorch_rf_gaussian.py:46-64): BigGAN residual blocks with FIR resampling,
input_skip / output_skip pyramids combined by sum, Gaussian Fourier time conditioning, skip_rescale, scale_by_sigma.

Only tests/ (and tools/ that generate fixtures) import it; the product path (pnpflow_amd/) never does.

Parity pinning: tests/golden/ncsnpp_*.npz hold outputs of the REAL reference module (tools/make_golden.py `gen_ncsnpp`, run in the
build container with the reference's JIT-compiled CUDA ops replaced by their own pure-torch definitions, which is the branch the
reference itself takes on CPU tensors, op/upfirdn2d.py:128-139); tests/test_oracle_golden.py checks this file against them.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import pnpflow_oracle as O


def ncsnpp_config(image_size: int = 256, nf: int = 128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks: int = 2, attn_resolutions=(16,),
                  num_channels: int = 3, fir_kernel=(1, 3, 3, 1), skip_rescale: bool = True, scale_by_sigma: bool = True) -> dict:
    """configs/default_lsun_configs.py:52-70 + configs/rectified_flow/celeba_hq_pytorch_rf_gaussian.py:43-64"""
    return dict(image_size=image_size, nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks, attn_resolutions=tuple(attn_resolutions),
                num_channels=num_channels, fir_kernel=tuple(fir_kernel), skip_rescale=skip_rescale, scale_by_sigma=scale_by_sigma)


def _modules(cfg: dict):
    """The module list of NCSNpp.__init__ (ncsnpp.py:72-206) as (kind, in_ch, out_ch, up, down) tuples, in construction order."""
    nf, mult, nres, chn = cfg["nf"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["num_channels"]
    nlev = len(mult)
    mods = [("fourier", 0, 0, False, False), ("linear", 2 * nf, 4 * nf, False, False), ("linear", 4 * nf, 4 * nf, False, False),
            ("conv_in", chn, nf, False, False)]
    hs_c = [nf]
    in_ch, res = nf, cfg["image_size"]
    for lvl in range(nlev):
        for _ in range(nres):
            o = nf * mult[lvl]
            mods.append(("res", in_ch, o, False, False)); in_ch = o
            if res in cfg["attn_resolutions"]:
                mods.append(("attn", in_ch, in_ch, False, False))
            hs_c.append(in_ch)
        if lvl != nlev - 1:
            mods.append(("res", in_ch, in_ch, False, True))
            mods.append(("combine", chn, in_ch, False, False))
            hs_c.append(in_ch); res //= 2
    in_ch = hs_c[-1]
    mods += [("res", in_ch, in_ch, False, False), ("attn", in_ch, in_ch, False, False), ("res", in_ch, in_ch, False, False)]
    for lvl in reversed(range(nlev)):
        for _ in range(nres + 1):
            o = nf * mult[lvl]
            mods.append(("res", in_ch + hs_c.pop(), o, False, False)); in_ch = o
        if res in cfg["attn_resolutions"]:
            mods.append(("attn", in_ch, in_ch, False, False))
        mods.append(("gn", in_ch, in_ch, False, False)); mods.append(("conv_out", in_ch, chn, False, False))
        if lvl != 0:
            mods.append(("res", in_ch, in_ch, True, False)); res *= 2
    assert not hs_c
    return mods


def ncsnpp_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """state_dict keys / shapes of NCSNpp (without DataParallel's `module.` and without the `sigmas` buffer)."""
    shp: "OrderedDict[str, tuple]" = OrderedDict()
    tdim = 4 * cfg["nf"]
    for idx, (kind, i, o, up, down) in enumerate(_modules(cfg)):
        P = f"all_modules.{idx}."
        if kind == "fourier":
            shp[P + "W"] = (cfg["nf"],)
        elif kind == "linear":
            shp[P + "weight"] = (o, i); shp[P + "bias"] = (o,)
        elif kind in ("conv_in", "conv_out"):
            shp[P + "weight"] = (o, i, 3, 3); shp[P + "bias"] = (o,)
        elif kind == "gn":
            shp[P + "weight"] = (i,); shp[P + "bias"] = (i,)
        elif kind == "combine":
            shp[P + "Conv_0.weight"] = (o, i, 1, 1); shp[P + "Conv_0.bias"] = (o,)
        elif kind == "attn":
            shp[P + "GroupNorm_0.weight"] = (i,); shp[P + "GroupNorm_0.bias"] = (i,)
            for k in range(4):
                shp[P + f"NIN_{k}.W"] = (i, i); shp[P + f"NIN_{k}.b"] = (i,)
        elif kind == "res":
            shp[P + "GroupNorm_0.weight"] = (i,); shp[P + "GroupNorm_0.bias"] = (i,)
            shp[P + "Conv_0.weight"] = (o, i, 3, 3); shp[P + "Conv_0.bias"] = (o,)
            shp[P + "Dense_0.weight"] = (o, tdim); shp[P + "Dense_0.bias"] = (o,)
            shp[P + "GroupNorm_1.weight"] = (o,); shp[P + "GroupNorm_1.bias"] = (o,)
            shp[P + "Conv_1.weight"] = (o, o, 3, 3); shp[P + "Conv_1.bias"] = (o,)
            if i != o or up or down:
                shp[P + "Conv_2.weight"] = (o, i, 1, 1); shp[P + "Conv_2.bias"] = (o,)
    return shp


def synthetic_state_dict(cfg: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seed-fixed synthetic weights, one numpy Philox stream per tensor keyed by (seed, crc32(name)) - NOT the reference
    initialiser (init_scale = 0 leaves every Conv_1 / NIN_3 / output conv at 1e-10, ncsnpp.py:61,95-97, so the net would output
    ~0): conv / Linear weights ~ U(-a, a) with a = sqrt(3 / fan_in); NIN W [in][out] likewise with fan_in = in; GroupNorm
    gamma ~ 1 + 0.1 U; biases ~ 0.05 U; the Fourier frequencies W ~ 16 N(0,1) as GaussianFourierProjection draws them."""
    sd = {}
    for name, shape in ncsnpp_param_shapes(cfg).items():
        rng = np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))
        leaf = name.rsplit(".", 1)[1]
        if leaf == "W" and len(shape) == 1:
            u = (rng.standard_normal(size=shape) * 16.0).astype(np.float32)
        else:
            u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
            if leaf == "W":
                u *= np.float32(math.sqrt(3.0 / shape[0]))
            elif leaf == "weight" and len(shape) >= 2:
                u *= np.float32(math.sqrt(3.0 / int(np.prod(shape[1:]))))
            elif leaf == "weight":
                u = np.float32(1.0) + np.float32(0.1) * u
            else:
                u *= np.float32(0.05)
        sd[name] = torch.from_numpy(u)
    return sd


def _gn(x, w, b):
    """nn.GroupNorm(num_groups=min(C // 4, 32), eps=1e-6) (layerspp.py:66, 213, 224; ncsnpp.py:192)"""
    return F.group_norm(x, min(x.shape[1] // 4, 32), w, b, eps=1e-6)


def _nin(x, W, b):
    """layers.py:546-555: channel contraction with W [in][out]"""
    return torch.einsum("bchw,co->bohw", x, W) + b[None, :, None, None]


def _res_block(sd, P, x, temb, up, down, fir, skip_rescale):
    """ResnetBlockBigGANpp.forward (layerspp.py:235-274)"""
    out_ch = sd[P + "Conv_0.weight"].shape[0]
    h = F.silu(_gn(x, sd[P + "GroupNorm_0.weight"], sd[P + "GroupNorm_0.bias"]))
    if up:
        h = O.upsample_2d(h, fir, factor=2); x = O.upsample_2d(x, fir, factor=2)
    elif down:
        h = O.downsample_2d(h, fir, factor=2); x = O.downsample_2d(x, fir, factor=2)
    h = F.conv2d(h, sd[P + "Conv_0.weight"], sd[P + "Conv_0.bias"], padding=1)
    h = h + F.linear(F.silu(temb), sd[P + "Dense_0.weight"], sd[P + "Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(h, sd[P + "GroupNorm_1.weight"], sd[P + "GroupNorm_1.bias"]))
    h = F.conv2d(h, sd[P + "Conv_1.weight"], sd[P + "Conv_1.bias"], padding=1)       # Dropout_0: p = 0 / eval
    if x.shape[1] != out_ch or up or down:
        x = F.conv2d(x, sd[P + "Conv_2.weight"], sd[P + "Conv_2.bias"])
    return (x + h) / np.sqrt(2.0) if skip_rescale else x + h


def _attn_block(sd, P, x, skip_rescale):
    """AttnBlockpp.forward (layerspp.py:76-94)"""
    B, C, H, W = x.shape
    h = _gn(x, sd[P + "GroupNorm_0.weight"], sd[P + "GroupNorm_0.bias"])
    q = _nin(h, sd[P + "NIN_0.W"], sd[P + "NIN_0.b"]); k = _nin(h, sd[P + "NIN_1.W"], sd[P + "NIN_1.b"]); v = _nin(h, sd[P + "NIN_2.W"], sd[P + "NIN_2.b"])
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, W, H * W), dim=-1).reshape(B, H, W, H, W)
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = _nin(h, sd[P + "NIN_3.W"], sd[P + "NIN_3.b"])
    return (x + h) / np.sqrt(2.0) if skip_rescale else x + h


def ncsnpp_forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, time_cond: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """NCSNpp.forward (ncsnpp.py:219-383) for data.centered = True; `time_cond` is what the solvers pass as `t * 999`
    (methods/pnp_flow.py:23-27).  `taps`, when given, receives named intermediate activations."""
    with torch.no_grad():
        return _forward(sd, cfg, x, time_cond, taps)


def ncsnpp_vjp(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, time_cond: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """J^T vec w.r.t. the image, as OT_ODE takes it: torch.autograd.functional.vjp(lambda x: model_forward(x, t), x, vec)[1]
    (ot_ode.py:137-138) - autograd through the restated forward."""
    return torch.autograd.functional.vjp(lambda z: _forward(sd, cfg, z, time_cond, None), x, vec)[1]


def _forward(sd, cfg, x, time_cond, taps):
    mods = _modules(cfg)
    fir, sr = cfg["fir_kernel"], cfg["skip_rescale"]
    nlev, nres = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    name = lambda i: f"all_modules.{i}."
    if True:
        # Gaussian Fourier features of log(sigma) (ncsnpp.py:226-230; layerspp.py:39-41), then the conditioning MLP (:240-246)
        used_sigmas = time_cond
        x_proj = torch.log(used_sigmas)[:, None] * sd[name(0) + "W"][None, :] * 2 * np.pi
        temb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
        temb = F.linear(temb, sd[name(1) + "weight"], sd[name(1) + "bias"])
        temb = F.linear(F.silu(temb), sd[name(2) + "weight"], sd[name(2) + "bias"])
        m = 3
        input_pyramid = x
        hs = [F.conv2d(x, sd[name(m) + "weight"], sd[name(m) + "bias"], padding=1)]; m += 1
        if taps is not None: taps["conv_in"] = hs[0]
        for lvl in range(nlev):
            for blk in range(nres):
                h = _res_block(sd, name(m), hs[-1], temb, False, False, fir, sr); m += 1
                if h.shape[-1] in cfg["attn_resolutions"]:
                    h = _attn_block(sd, name(m), h, sr); m += 1
                hs.append(h)
                if taps is not None: taps[f"down{lvl}_{blk}"] = h
            if lvl != nlev - 1:
                h = _res_block(sd, name(m), hs[-1], temb, False, True, fir, sr); m += 1
                input_pyramid = O.downsample_2d(input_pyramid, fir, factor=2)                      # pyramid_downsample (:284)
                h = F.conv2d(input_pyramid, sd[name(m) + "Conv_0.weight"], sd[name(m) + "Conv_0.bias"]) + h; m += 1   # Combine 'sum'
                hs.append(h)
                if taps is not None: taps[f"downsample{lvl}"] = h
        h = hs[-1]
        h = _res_block(sd, name(m), h, temb, False, False, fir, sr); m += 1
        h = _attn_block(sd, name(m), h, sr); m += 1
        h = _res_block(sd, name(m), h, temb, False, False, fir, sr); m += 1
        if taps is not None: taps["mid"] = h
        pyramid = None
        for lvl in reversed(range(nlev)):
            for blk in range(nres + 1):
                h = _res_block(sd, name(m), torch.cat([h, hs.pop()], dim=1), temb, False, False, fir, sr); m += 1
                if taps is not None: taps[f"up{lvl}_{blk}"] = h
            if h.shape[-1] in cfg["attn_resolutions"]:
                h = _attn_block(sd, name(m), h, sr); m += 1
            ph = F.silu(_gn(h, sd[name(m) + "weight"], sd[name(m) + "bias"])); m += 1
            ph = F.conv2d(ph, sd[name(m) + "weight"], sd[name(m) + "bias"], padding=1); m += 1
            pyramid = ph if pyramid is None else O.upsample_2d(pyramid, fir, factor=2) + ph       # output_skip (:332-343)
            if lvl != 0:
                h = _res_block(sd, name(m), h, temb, True, False, fir, sr); m += 1
                if taps is not None: taps[f"upsample{lvl}"] = h
        assert not hs and m == len(mods)
        h = pyramid
        if cfg["scale_by_sigma"]:
            h = h / used_sigmas.reshape(x.shape[0], 1, 1, 1)                                   # (:378-381)
        return h
