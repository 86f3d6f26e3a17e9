"""Config surface, model factory/loader and the PSNR bookkeeping of pnpflow/utils.py
(reference :37-240, :560-674, :1112-1120), for the restoration path only.
"""
from __future__ import annotations

import copy
import os
from ast import literal_eval
from collections import defaultdict
from typing import List

import numpy as np
import torch
import yaml

from . import _lib
from .models import UNet


# ---- configuration (reference pnpflow/utils.py:37-167) -------------------------------------
class CfgNode(dict):
    """Attribute-style dict, as the reference's CfgNode: `args.key` everywhere, mutable."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        for k, v in list(init_dict.items()):
            if type(v) is dict:
                init_dict[k] = CfgNode(v)
        super().__init__(init_dict)

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    if type(new) == type(old):
        return new
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) with values ({old} vs. {new}) for config key: {key}")


def load_cfg_from_cfg_file(file: str) -> CfgNode:
    """One top-level section per YAML file, flattened away (reference utils.py:135-148)."""
    assert os.path.isfile(file) and file.endswith('.yaml'), f'{file} is not a yaml file'
    with open(file, 'r') as f:
        raw = yaml.safe_load(f)
    flat = {}
    for section in raw:
        flat.update(raw[section])
    return CfgNode(flat)


def merge_cfg_from_list(cfg: CfgNode, cfg_list: List[str]) -> CfgNode:
    """`--opts k v k v ...`; known keys are type-checked, unknown keys accepted (utils.py:151-167)."""
    new = copy.deepcopy(cfg)
    assert len(cfg_list) % 2 == 0, cfg_list
    for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key = full_key.split('.')[-1]
        val = _decode(v)
        if key in cfg:
            val = _coerce(val, cfg[key], key)
        setattr(new, key, val)
    return new


# ---- model factory / loader (reference pnpflow/utils.py:170-240) ----------------------------
def define_model(args):
    if args.model in ("ot", "indep"):
        model = UNet(input_channels=args.num_channels, input_height=args.dim_image, ch=32, ch_mult=(1, 2, 4, 8),
                     num_res_blocks=6, attn_resolutions=(16, 8), resamp_with_conv=True)
        return (model, None)
    raise Exception("Unknown model! (this engine implements the 'ot'/'indep' U-Net velocity field)")


def load_model(name_model, model, state, download=False, checkpoint_path=None, dataset=None, device='cuda'):
    if name_model not in ("ot", "indep"):
        raise NotImplementedError(name_model)
    if download:
        raise RuntimeError("no network: place the reference's model_final.pt at checkpoint_path")
    model.load_state_dict(torch.load(checkpoint_path, map_location='cpu'))
    model.to(device)


# ---- metric (reference pnpflow/utils.py:560-577, 594-674) ------------------------------------
def postprocess(img, args=None):
    return (img + 1) / 2


def psnr_per_image(rec: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """PSNR (data_range 1) of postprocess(rec) vs postprocess(clean) per image, on the GPU
    (pf_psnr).  Restates torchmetrics peak_signal_noise_ratio(dim=(1,2,3))."""
    import ctypes as C
    lib = _lib.load()
    rec = rec.contiguous().float(); clean = clean.to(rec.device).contiguous().float()
    out = torch.empty(rec.shape[0], dtype=torch.float32, device=rec.device)
    _lib.check(lib.pf_psnr(rec.data_ptr(), clean.data_ptr(), out.data_ptr(), rec.shape[0], rec[0].numel(), _lib.current_stream_ptr()),
               None, "pf_psnr")
    return out


def compute_psnr(clean_img, noisy_img, rec_img, args, H_adj, iter='final'):
    """Appends '{iter} {psnr}' lines to psnr_{rec,noisy}_batch{b}.txt (reference utils.py:594-625)."""
    dev = rec_img.device
    clean = clean_img.to(dev)
    noisy = noisy_img.to(dev)
    if args.problem in ('superresolution', 'superresolution_bicubic'):
        noisy = H_adj(noisy)
    psnr_rec = float(psnr_per_image(rec_img, clean).mean())
    psnr_noisy = float(psnr_per_image(noisy, clean).mean())
    for word, val in (('rec', psnr_rec), ('noisy', psnr_noisy)):
        with open(os.path.join(args.save_path_ip, f'psnr_{word}_batch{args.batch}.txt'), 'a') as f:
            f.write(f'{iter} {val}\n')
    return psnr_rec, psnr_noisy


def compute_average_psnr(args):
    """reference utils.py:628-674"""
    final = {}
    for word in ['rec', 'noisy']:
        by_it = defaultdict(list)
        for batch in range(args.max_batch):
            with open(os.path.join(args.save_path_ip, f'psnr_{word}_batch{batch}.txt'), 'r') as f:
                for line in f:
                    it, val = map(float, line.strip().split())
                    by_it[int(it)].append(val)
        avg_file = os.path.join(args.save_path_ip, f'psnr_{word}_average.txt')
        with open(avg_file, 'a') as f:
            for it, vals in sorted(by_it.items()):
                f.write(f'{it} {np.mean(vals):.4f}\n')
        with open(avg_file, 'r') as f:
            final[word] = [float(l.split()[1]) for l in f.readlines()][-1]
    path = os.path.join(args.save_path, 'final_psnr.txt')
    with open(path, 'a') as f:
        if os.stat(path).st_size == 0:
            f.write('psnr_rec psnr_noisy ' + ''.join(f'{k} ' for k in args.dict_cfg_method.keys()) + '\n')
        f.write(f"{final['rec']} {final['noisy']} " + ''.join(f'{v} ' for v in args.dict_cfg_method.values()) + '\n')
    return final


def save_time_use(d, args):
    with open(os.path.join(args.save_path_ip, 'time_stats.txt'), "a") as f:
        f.write(str(d) + '\n')


def get_save_path_ip(dict_cfg_method):
    """key1=value1/key2=value2/... (reference utils.py:1112-1120)"""
    path = ""
    for key, value in dict_cfg_method.items():
        path = os.path.join(path, f"{key}={value}")
    return path
