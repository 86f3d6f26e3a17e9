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
                     num_res_blocks=6, attn_resolutions=(16, 8), resamp_with_conv=True,
                     device_index=int(getattr(args, "device_index", 0)))
        return (model, None)
    if args.model == "rectified":
        # utils.py:186-204 of the reference: the NCSN++ net of the dataset's rectified-flow config.  `state` keeps the key the
        # reference's load_model fills ('model'); the optimizer / EMA entries of the training state are not part of inference.
        from .image_generation.models import utils as mutils
        from .image_generation.configs.rectified_flow.celeba_hq_pytorch_rf_gaussian import get_config as get_config_celebahq
        from .image_generation.configs.rectified_flow.afhq_cat_pytorch_rf_gaussian import get_config as get_config_afhq_cat
        if args.dataset == "celebahq":
            config = get_config_celebahq()
        elif args.dataset == "afhq_cat":
            config = get_config_afhq_cat()
        else:
            raise Exception("the rectified model exists for celebahq and afhq_cat")
        score_model = mutils.create_model(config, device_index=int(getattr(args, "device_index", 0)))
        return score_model, dict(model=score_model, step=0)
    raise Exception("Unknown model! (this engine implements the 'ot'/'indep' U-Net and the 'rectified' NCSN++ velocity fields)")


def load_model(name_model, model, state, download=False, checkpoint_path=None, dataset=None, device='cuda'):
    if name_model == "rectified":
        # image_generation/utils.py:7-13 `restore_checkpoint`: state['model'].load_state_dict(loaded_state['model'], strict=False)
        loaded_state = torch.load(checkpoint_path, map_location='cpu')
        state['model'].load_state_dict(loaded_state['model'], strict=False)
        state['step'] = loaded_state.get('step', 0)
        return state
    if name_model not in ("ot", "indep"):
        raise NotImplementedError(name_model)
    if download:
        raise RuntimeError("no network: place the reference's model_final.pt at checkpoint_path")
    model.load_state_dict(torch.load(checkpoint_path, map_location='cpu'))
    model.to(device)


# ---- metric (reference pnpflow/utils.py:560-577, 594-674) ------------------------------------
def draw_measurement_noise(batch, gshape, lo, hi, device, source="cpu"):
    """The `torch.manual_seed(batch); torch.randn_like(noisy_img)` draw of pnp_flow.py:79-80 / ot_ode.py:44-45 for images [lo, hi) of a
    global batch of shape `gshape`.  source "cpu" (default): the CPU generator - the same values on any device and on every rank.
    source "device": the reference's own behaviour - the draw is made on `device`'s generator (for the whole global batch, then
    sliced: a shard reproduces the single-device run), so a seed-fixed run equals a run of the reference on the same device type."""
    torch.manual_seed(batch)
    if source == "device":
        return torch.randn(gshape, dtype=torch.float32, device=device)[lo:hi].contiguous()
    if source != "cpu":
        raise ValueError(f"measurement_noise must be 'cpu' or 'device', not {source!r}")
    return torch.randn(gshape, dtype=torch.float32)[lo:hi].to(device)


def postprocess(img, args=None):
    return (img + 1) / 2


def psnr_per_image(rec: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """PSNR (data_range 1) of postprocess(rec) vs postprocess(clean) per image, on the GPU
    (pf_psnr).  Restates torchmetrics peak_signal_noise_ratio(dim=(1,2,3))."""
    lib = _lib.load()
    rec = rec.contiguous().float(); clean = clean.to(rec.device).contiguous().float()
    out = torch.empty(rec.shape[0], dtype=torch.float32, device=rec.device)
    if rec.shape[0] == 0:
        return out              # an empty shard contributes a 0-length vector to the gather (parallel.gather_in_image_order)
    _lib.check(lib.pf_psnr(rec.data_ptr(), clean.data_ptr(), out.data_ptr(), rec.shape[0], rec[0].numel(), _lib.current_stream_ptr()),
               None, "pf_psnr")
    return out


def ssim_per_image(rec: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """SSIM (data_range 1, 11x11 Gaussian window sigma 1.5, k1 0.01, k2 0.03, reflect padding) of postprocess(rec) vs
    postprocess(clean) per image, on the GPU (pf_ssim).  Restates ignite.metrics.SSIM as the reference uses it
    (utils.py:780-802); ignite is not installed here, so this metric is PARITY UNPINNED (oracle: O.ssim_per_image)."""
    lib = _lib.load()
    rec = rec.contiguous().float(); clean = clean.to(rec.device).contiguous().float()
    B, Cc, H, W = rec.shape
    out = torch.empty(B, dtype=torch.float64, device=rec.device)
    if B == 0:
        return out
    _lib.check(lib.pf_ssim(rec.data_ptr(), clean.data_ptr(), out.data_ptr(), B, Cc, H, W, _lib.current_stream_ptr()), None, "pf_ssim")
    return out


def _is_writer():
    from . import parallel
    return parallel.rank_world()[0] == 0


def _global_mean(per_image: torch.Tensor) -> float:
    """Mean over the (global) batch; with several ranks the per-image values are all_gathered in image order first -
    the ONE data-path collective of a sharded run (parallel.py)."""
    from . import parallel
    return float(parallel.gather_in_image_order(per_image).double().mean())


def _append_metric(args, name, word, iter, val):
    if _is_writer():
        with open(os.path.join(args.save_path_ip, f'{name}_{word}_batch{args.batch}.txt'), 'a') as f:
            f.write(f'{iter} {val}\n')


def compute_psnr(clean_img, noisy_img, rec_img, args, H_adj, iter='final'):
    """Appends '{iter} {psnr}' lines to psnr_{rec,noisy}_batch{b}.txt (reference utils.py:594-625): the mean over the
    batch of the per-image PSNR."""
    dev = rec_img.device
    clean = clean_img.to(dev)
    noisy = noisy_img.to(dev)
    if args.problem in ('superresolution', 'superresolution_bicubic'):
        # the reference post-processes the measurement BEFORE H_adj and the result again (utils.py:598-607);
        # psnr_per_image applies the outer postprocess itself
        noisy = H_adj(postprocess(noisy))
    psnr_rec = _global_mean(psnr_per_image(rec_img, clean))
    psnr_noisy = _global_mean(psnr_per_image(noisy, clean))
    _append_metric(args, 'psnr', 'rec', iter, psnr_rec)
    _append_metric(args, 'psnr', 'noisy', iter, psnr_noisy)
    return psnr_rec, psnr_noisy


def compute_ssim(clean_img, noisy_img, rec_img, args, H_adj, iter='final'):
    """ssim_{rec,noisy}_batch{b}.txt (reference utils.py:780-816).  PARITY UNPINNED: see ssim_per_image."""
    dev = rec_img.device
    clean = clean_img.to(dev)
    noisy = noisy_img.to(dev)
    if args.problem in ('superresolution', 'superresolution_bicubic'):
        noisy = H_adj(noisy)                      # single postprocess here (utils.py:782-783)
    ssim_rec = _global_mean(ssim_per_image(rec_img, clean))
    ssim_noisy = _global_mean(ssim_per_image(noisy, clean))
    _append_metric(args, 'ssim', 'rec', iter, ssim_rec)
    _append_metric(args, 'ssim', 'noisy', iter, ssim_noisy)
    return ssim_rec, ssim_noisy


# ---- image grids (reference utils.py:433-538) ---------------------------------------------------------------------------------
_WARNED = set()


def _imshow_grid(path, imgs: np.ndarray, gray: bool):
    """One figure with every image of the batch, laid out as the reference lays it out: 1 image -> a plain figure, 2 -> one row,
    otherwise int(sqrt(B)) columns x int(B / cols) rows on a 20 x 20 inch canvas, image k = i + j * rows at axes (i, j), ticks off.
    Drawn with matplotlib's object API on an Agg canvas (the process-wide pyplot state and backend are left alone)."""
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    B = imgs.shape[0]
    show = lambda ax, im: ax.imshow(im[..., 0], cmap='gray', vmin=0, vmax=1) if gray else ax.imshow(im)
    if B == 1:
        fig = Figure(); FigureCanvasAgg(fig)
        show(fig.add_subplot(111), imgs[0])
    elif B == 2:
        fig = Figure(); FigureCanvasAgg(fig)
        axes = fig.subplots(1, 2)
        for k in range(2):
            show(axes[k], imgs[k]); axes[k].set_xticks([]); axes[k].set_yticks([])
    else:
        cols = int(np.sqrt(B)); rows = int(B / cols)
        fig = Figure(figsize=(20, 20)); FigureCanvasAgg(fig)
        axes = fig.subplots(rows, cols, squeeze=False)     # (the reference indexes a squeezed array: it raises for B = 3)
        for i in range(rows):
            for j in range(cols):
                show(axes[i, j], imgs[i + j * rows]); axes[i, j].set_xticks([]); axes[i, j].set_yticks([])
    fig.savefig(path)


def save_images(clean_img, noisy_img, rec_img, args, H_adj, iter='final'):
    """Image files of one batch (reference utils.py:433-538, called with iter='final' from pnp_flow.py:156 / ot_ode.py:182):
      iter != 'final':  {problem}_{method}_batch{b}_iter{iter}.png          grid of the restored images
      iter == 'final':  {problem}_{clean|noisy|<method>}_batch{b}_final.png  one grid each
      eval_split == 'test' and batch < 4: one .eps per image (clean / measurement / restored), the PSNR of the image in the file name.
    Everything goes through postprocess (no clamp; imshow clips to [0, 1]).  With several ranks the shards are gathered in image
    order (every rank must call this) and rank 0 draws.  Needs matplotlib, as the reference does; without it a warning is printed once
    and nothing is written."""
    from . import parallel
    dev = rec_img.device
    noisy_dev = noisy_img.to(dev)
    tensors = (clean_img.to(dev), noisy_dev, rec_img, H_adj(torch.ones_like(noisy_dev)))
    full = [parallel.gather_in_image_order(postprocess(t.detach().float()).contiguous().reshape(-1)).reshape((-1,) + tuple(t.shape[1:])) for t in tensors]
    if not _is_writer() or full[0].shape[0] == 0:
        return
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        if 'matplotlib' not in _WARNED:
            _WARNED.add('matplotlib'); print('save_images: matplotlib is not installed - no image files are written')
        return
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure
    clean, noisy, rec, hadj_ones = (t.permute(0, 2, 3, 1).cpu().numpy() for t in full)
    gray = int(getattr(args, 'num_channels', clean.shape[-1])) == 1
    words = ['clean', 'noisy', args.method]
    if iter != 'final':
        _imshow_grid(os.path.join(args.save_path_ip, f"{args.problem}_{args.method}_batch{args.batch}_iter{iter}.png"), rec, gray)
    else:
        for word, imgs in zip(words, (clean, noisy, rec)):
            _imshow_grid(os.path.join(args.save_path_ip, f"{args.problem}_{word}_batch{args.batch}_final.png"), imgs, gray)
    if getattr(args, 'eval_split', None) == 'test' and ((args.batch < 8 and args.method == 'd_flow') or args.batch < 4):
        psnr = lambda a, b: float(10.0 * np.log10(1.0 / np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))   # skimage's, data_range 1
        for i in range(clean.shape[0]):
            measured = hadj_ones[i] if args.problem in ('superresolution', 'superresolution_bicubic') else noisy[i]   # (sic: H_adj of ones, utils.py:436,509-511)
            p_noisy, p_rec = psnr(clean[i], measured), psnr(clean[i], rec[i])
            stem = os.path.join(args.save_path_ip, f"{args.problem}_")
            targets = []
            if args.method == 'pnp_flow':
                targets += [(clean[i], f"{stem}clean_batch{args.batch}_im{i}.eps"), (noisy[i], f"{stem}noisy_batch{args.batch}_im{i}_pnsr{p_noisy:4.2f}.eps")]
            targets.append((rec[i], f"{stem}{args.method}_batch{args.batch}_im{i}_iter{iter}_pnsr{p_rec:4.2f}.eps"))
            for im, path in targets:
                fig = Figure(); FigureCanvasAgg(fig)
                ax = fig.add_subplot(111); ax.imshow(im[..., 0] if im.shape[-1] == 1 else im); ax.axis('off')
                fig.savefig(path, bbox_inches='tight', pad_inches=0)


# ---- LPIPS (reference utils.py:677-776) ------------------------------------------------------------------------
_LPIPS = {"model": None, "resolved": False}


def set_lpips_model(model):
    """Installs the LPIPS network the logging uses (tests: synthetic weights under the published key names)."""
    _LPIPS["model"], _LPIPS["resolved"] = model, True


def lpips_model(device_index=0):
    """The AlexNet-LPIPS network on the engine, or None when its weight files are not on this machine (the reference downloads
    them on first use; there is no network here): looked up once, where torchvision / lpips keep them (lpips.find_weights)."""
    if not _LPIPS["resolved"]:
        from . import lpips as L
        alex, lin = L.find_weights()
        _LPIPS["resolved"] = True
        if alex and lin:
            _LPIPS["model"] = L.LPIPS.from_files(alex, lin, device_index)
        else:
            print("[pnpflow_amd] LPIPS weights not found (alexnet-owt-*.pth / lpips alex.pth; set PNPFLOW_LPIPS_DIR): "
                  "lpips_*.txt files are not written (`--opts lpips require` makes this an error)")
    return _LPIPS["model"]


def compute_lpips(clean_img, noisy_img, rec_img, args, H_adj, iter='final'):
    """lpips_{rec,noisy}_batch{b}.txt (reference utils.py:677-724), including its input convention: images post-processed to
    [0, 1], mapped back to [-1, 1] and then passed with normalize=True (a second 2x - 1: the network sees [-3, 1]); for
    superresolution the 'noisy' image is postprocess(H_adj(postprocess(y))).  PARITY UNPINNED (pnpflow_amd/lpips.py).
    args.lpips: 'auto' (default: skipped with one notice when the weight files are absent), 'require', 'off'."""
    mode = getattr(args, 'lpips', 'auto')
    if mode == 'off':
        return None
    dev = rec_img.device
    model = lpips_model(dev.index or 0)
    if model is None:
        if mode == 'require':
            raise FileNotFoundError("LPIPS weights (torchvision alexnet-owt-*.pth + lpips weights/v0.1/alex.pth) not found; set PNPFLOW_LPIPS_DIR")
        return None
    clean = postprocess(clean_img.to(dev).clone()); noisy = postprocess(noisy_img.to(dev).clone()); rec = postprocess(rec_img.clone())
    if args.problem in ('superresolution', 'superresolution_bicubic'):
        noisy = postprocess(H_adj(noisy))
    clean, rec, noisy = 2 * clean - 1, 2 * rec - 1, 2 * noisy - 1
    lp_rec = _global_mean(model(clean, rec, normalize=True))
    lp_noisy = _global_mean(model(clean, noisy, normalize=True))
    _append_metric(args, 'lpips', 'rec', iter, lp_rec)
    _append_metric(args, 'lpips', 'noisy', iter, lp_noisy)
    return lp_rec, lp_noisy


def compute_average_lpips(args):
    """reference utils.py:727-776; nothing to average when LPIPS was skipped (no weight files)."""
    if getattr(args, 'lpips', 'auto') == 'off' or _LPIPS["model"] is None:
        return None
    return _average_metric(args, 'lpips')


PARITY_UNPINNED = {
    'ssim': "ignite.metrics.SSIM restated from its published algorithm (ignite is not installed in the build image); checked against this repo's own oracle only",
    'lpips': "lpips.LPIPS(net='alex') restated from the lpips / torchvision definitions (neither package installed in the build image); checked against this repo's own oracle only",
}


def _average_metric(args, name):
    """reference utils.py:628-674 (psnr) / 819-863 (ssim): per-iteration mean over the batches, then the last value into
    final_{name}.txt next to the method's hyper-parameters."""
    if not _is_writer():
        return None
    final = {}
    for word in ['rec', 'noisy']:
        by_it = defaultdict(list)
        for batch in range(args.max_batch):
            with open(os.path.join(args.save_path_ip, f'{name}_{word}_batch{batch}.txt'), 'r') as f:
                for line in f:
                    it, val = map(float, line.strip().split())
                    by_it[int(it)].append(val)
        avg_file = os.path.join(args.save_path_ip, f'{name}_{word}_average.txt')
        with open(avg_file, 'a') as f:
            for it, vals in sorted(by_it.items()):
                f.write(f'{it} {np.mean(vals):.4f}\n')
        with open(avg_file, 'r') as f:
            final[word] = [float(l.split()[1]) for l in f.readlines()][-1]
    path = os.path.join(args.save_path, f'final_{name}.txt')
    with open(path, 'a') as f:
        if os.stat(path).st_size == 0:
            f.write(f'{name}_rec {name}_noisy ' + ''.join(f'{k} ' for k in args.dict_cfg_method.keys()) + '\n')
        f.write(f"{final['rec']} {final['noisy']} " + ''.join(f'{v} ' for v in args.dict_cfg_method.values()) + '\n')
    if name in PARITY_UNPINNED:
        # the result files keep the reference's exact format (scripts parse them); what is a restatement of an absent third-party
        # implementation is said in a sidecar next to them
        note = os.path.join(args.save_path, 'PARITY_UNPINNED.txt')
        line = f'final_{name}.txt, {name}_*: {PARITY_UNPINNED[name]}\n'
        if not os.path.isfile(note) or line not in open(note).read():
            with open(note, 'a') as f:
                f.write(line)
    return final


def compute_average_psnr(args):
    return _average_metric(args, 'psnr')


def compute_average_ssim(args):
    return _average_metric(args, 'ssim')


def save_time_use(d, args):
    if _is_writer():
        with open(os.path.join(args.save_path_ip, 'time_stats.txt'), "a") as f:
            f.write(str(d) + '\n')


def save_memory_use(d, args):
    """reference utils.py:580-584"""
    if _is_writer():
        with open(os.path.join(args.save_path_ip, 'memory_stats.txt'), "a") as f:
            f.write(str(d) + '\n')


def _average_stat(args, stats_file, key, out_file, label):
    """reference utils.py:866-901: the first record of every batch index, averaged."""
    if not _is_writer():
        return None
    vals = torch.zeros(args.max_batch)
    for batch in range(args.max_batch):
        with open(os.path.join(args.save_path_ip, stats_file), 'r') as f:
            for line in f:
                rec = literal_eval(line.strip())
                if rec['batch'] == batch:
                    vals[batch] = rec[key]
                    break
    with open(os.path.join(args.save_path_ip, out_file), 'a') as f:
        f.write(f'{label}: {vals.mean().item():.4f}\n')
    return vals.mean().item()


def compute_average_time(args):
    return _average_stat(args, 'time_stats.txt', 'time_per_batch', 'time_average.txt', 'average time')


def compute_average_memory(args):
    return _average_stat(args, 'memory_stats.txt', 'max_allocated', 'max_memory_average.txt', 'average mem')


def get_save_path_ip(dict_cfg_method):
    """key1=value1/key2=value2/... (reference utils.py:1112-1120)"""
    path = ""
    for key, value in dict_cfg_method.items():
        path = os.path.join(path, f"{key}={value}")
    return path
