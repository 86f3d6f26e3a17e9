"""Degradation operators with the API of pnpflow/degradations.py (reference :6-127):
`Degradation.H(x)`, `.H_adj(x)` on (B,C,H,W) fp32 GPU tensors, executed by the HIP
kernels in csrc/pointwise.hip.  Each operator also exposes `descriptor(B, H, W, device)`,
the pf_degradation record the fused solver kernels consume.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _scratch(x):
    return torch.empty_like(x)


class Degradation:
    kind = None

    def H(self, x):
        raise NotImplementedError()

    def H_adj(self, x):
        raise NotImplementedError()

    # -- engine side -----------------------------------------------------------------------
    def descriptor(self, B, H, W, device):
        raise NotImplementedError()

    def _apply(self, x, adjoint: bool, out_hw=None):
        if not x.is_cuda:
            raise _lib.PnpFlowHipError("degradation operators need GPU tensors (there is no CPU path)")
        lib = _lib.load()
        x = x.contiguous().float()
        B, Cc = x.shape[0], x.shape[1]
        Hf, Wf = out_hw if out_hw is not None else (x.shape[2], x.shape[3])   # full-resolution size
        d = self.descriptor(B, Hf, Wf, x.device)
        if adjoint:
            out = torch.empty((B, Cc, Hf, Wf), dtype=torch.float32, device=x.device)
        else:
            sf = getattr(self, "sf", 1) if self.kind in (_lib.PF_DEG_SUPERRESOLUTION, _lib.PF_DEG_SR_FILTERED) else 1
            out = torch.empty((B, Cc, Hf // sf, Wf // sf), dtype=torch.float32, device=x.device)
        if B == 0:
            return out          # an empty shard (global batch smaller than the number of ranks): nothing to launch
        n_scr = {_lib.PF_DEG_GAUSSIAN_BLUR: 1, _lib.PF_DEG_SR_FILTERED: 2}.get(self.kind, 0)
        scratch = torch.empty((n_scr, B, Cc, Hf, Wf), dtype=torch.float32, device=x.device) if n_scr else None
        fn = lib.pf_degradation_H_adj if adjoint else lib.pf_degradation_H
        _lib.check(fn(C.byref(d), x.data_ptr(), out.data_ptr(), B, Cc, Hf, Wf,
                      scratch.data_ptr() if scratch is not None else None, _lib.current_stream_ptr()), None, fn.__name__)
        return out


class Denoising(Degradation):
    """reference pnpflow/degradations.py:15-20"""
    kind = _lib.PF_DEG_DENOISING

    def descriptor(self, B, H, W, device):
        d = _lib.PfDegradation(); d.kind = self.kind
        return d

    def H(self, x):
        return x

    def H_adj(self, x):
        return x


class BoxInpainting(Degradation):
    """reference pnpflow/degradations.py:23-32 (mask: utils.py:327-336)"""
    kind = _lib.PF_DEG_BOX_INPAINTING

    def __init__(self, half_size_mask):
        super().__init__()
        self.half_size_mask = half_size_mask

    def descriptor(self, B, H, W, device):
        d = _lib.PfDegradation(); d.kind = self.kind; d.half_size_mask = int(self.half_size_mask)
        return d

    def H(self, x):
        return self._apply(x, False)

    def H_adj(self, x):
        return self._apply(x, True)


class RandomInpainting(Degradation):
    """reference pnpflow/degradations.py:35-44.  The mask is the reference's
    np.random.seed(42); binomial(1, 1-p, (B,H,W)) bit pattern (utils.py:357-359),
    generated ONCE per (global batch, H, W) instead of on every H/H_adj call and, for
    multi-GPU shards, sliced [offset, offset+B) out of the global-batch draw (the draw is
    prefix-consistent in B)."""
    kind = _lib.PF_DEG_MASK_INPAINTING

    def __init__(self, p, global_batch=None, batch_offset=0):
        super().__init__()
        self.p = p
        self.global_batch, self.batch_offset = global_batch, batch_offset
        self._cache = {}

    def set_shard(self, global_batch, batch_offset):
        """This process restores images [batch_offset, batch_offset + B) of a `global_batch`-image batch."""
        if (global_batch, batch_offset) != (self.global_batch, self.batch_offset):
            self.global_batch, self.batch_offset = global_batch, batch_offset
            self._cache = {}

    def mask(self, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in self._cache:
            G = self.global_batch if self.global_batch is not None else B + self.batch_offset
            m = np.random.RandomState(42).binomial(n=1, p=1 - self.p, size=(G, H, W))
            m = m[self.batch_offset:self.batch_offset + B].astype(np.uint8)
            self._cache[key] = torch.from_numpy(np.ascontiguousarray(m)).to(device)
        return self._cache[key]

    def descriptor(self, B, H, W, device):
        d = _lib.PfDegradation(); d.kind = self.kind
        d.mask = self.mask(B, H, W, device).data_ptr()
        return d

    def H(self, x):
        return self._apply(x, False)

    def H_adj(self, x):
        return self._apply(x, True)


def paintbrush_masks(B, H, W):
    """Keep-masks (1 = observed) of pnpflow/utils.py:339-350 + MaskGenerator._generate_mask (:904-924) for a batch of B images:
    `random.seed(42)`, then per image 10 strokes with endpoints randint(W//2-30, W//2+30) x randint(H//2-30, H//2+30) and
    thickness randint(8, int((W+H)*0.08)) - Python's own Mersenne-Twister sequence, so endpoints and thicknesses ARE the
    reference's.  Each stroke is rasterised as cv2.line does it (cv_draw.py restates OpenCV's ThickLine: 16.16 fixed-point
    quadrilateral fill + outline + midpoint-circle caps).  cv2 itself is not installed here: PARITY UNPINNED."""
    import random
    from .cv_draw import thick_line
    if W < 64 or H < 64:
        raise Exception("Width and Height of mask must be at least 64!")
    rng = random.Random(42)
    size = int((W + H) * 0.08)
    out = np.ones((B, H, W), dtype=np.uint8)
    for b in range(B):
        img = np.zeros((H, W), dtype=np.uint8)
        for _ in range(10):
            x1, x2 = rng.randint(W // 2 - 30, W // 2 + 30), rng.randint(W // 2 - 30, W // 2 + 30)
            y1, y2 = rng.randint(H // 2 - 30, H // 2 + 30), rng.randint(H // 2 - 30, H // 2 + 30)
            thick_line(img, (x1, y1), (x2, y2), rng.randint(8, size))
        out[b][img != 0] = 0
    return out


class PaintbrushInpainting(Degradation):
    """reference pnpflow/degradations.py:47-52 (mask: utils.py:339-350, 904-924).  H = H_adj = mask * x with the seeded
    brush-stroke masks, generated once per (global batch, H, W) and sliced for multi-GPU shards (the stroke sequence is
    prefix-consistent in the batch index).  Stroke rasterisation: see `paintbrush_masks` (parity unpinned without cv2)."""
    kind = _lib.PF_DEG_MASK_INPAINTING

    def __init__(self, global_batch=None, batch_offset=0):
        super().__init__()
        self.global_batch, self.batch_offset = global_batch, batch_offset
        self._cache = {}

    def set_shard(self, global_batch, batch_offset):
        if (global_batch, batch_offset) != (self.global_batch, self.batch_offset):
            self.global_batch, self.batch_offset = global_batch, batch_offset
            self._cache = {}

    def mask(self, B, H, W, device):
        key = (B, H, W, str(device))
        if key not in self._cache:
            m = paintbrush_masks(self.batch_offset + B, H, W)[self.batch_offset:self.batch_offset + B]
            self._cache[key] = torch.from_numpy(np.ascontiguousarray(m)).to(device)
        return self._cache[key]

    def descriptor(self, B, H, W, device):
        d = _lib.PfDegradation(); d.kind = self.kind
        d.mask = self.mask(B, H, W, device).data_ptr()
        return d

    def H(self, x):
        return self._apply(x, False)

    def H_adj(self, x):
        return self._apply(x, True)


class GaussianDeblurring(Degradation):
    """reference pnpflow/degradations.py:55-89, mode 'fft': circular convolution with the
    61x61 normalised Gaussian (utils.py:273-280).  The kernel is exactly separable, so the
    engine runs two 1-D circular passes instead of three FFTs per call."""
    kind = _lib.PF_DEG_GAUSSIAN_BLUR

    def __init__(self, sigma_blur, kernel_size, mode="fft", num_channels=3, dim_image=128, device="cuda"):
        super().__init__()
        if mode != "fft":
            raise NotImplementedError("only the circular ('fft') mode used by main.py is implemented")
        self.mode, self.sigma, self.kernel_size = mode, sigma_blur, kernel_size
        ax = np.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
        g = np.exp(-(ax ** 2) / (2 * sigma_blur ** 2))
        self.taps_host = (g / g.sum()).astype(np.float32)
        # Taps the fp32 sums cannot see are not walked: a 61-tap Gaussian of sigma 1 has 15 taps above 2^-36 of its peak (the
        # contribution of the rest is < 1e-10 of a sum that is rounded at 6e-8; the reference applies the filter through three
        # FFTs whose own rounding is three orders above that).  The engine's passes loop over, and stage halos for, these taps only.
        keep = np.nonzero(self.taps_host >= self.taps_host.max() * 2.0 ** -36)[0]
        re = int(max(kernel_size // 2 - keep[0], keep[-1] - kernel_size // 2))
        self.taps_eff = np.ascontiguousarray(self.taps_host[kernel_size // 2 - re: kernel_size // 2 + re + 1])
        self._taps = {}
        self.num_channels, self.dim_image, self._device = num_channels, dim_image, device
        self._filter = None

    @property
    def filter(self):
        """The reference's attribute (degradations.py:59-69; read by a Python-side ot_ode loop, ot_ode.py:110): the normalised
        2-D Gaussian, zero-padded to (1, C, dim, dim) and rolled so that its centre sits at (0, 0)."""
        if self._filter is None:
            K, D = self.kernel_size, self.dim_image
            ax = torch.arange(-K // 2 + 1.0, K // 2 + 1.0)                      # utils.py:273-280, same fp32 arithmetic
            xx, yy = torch.meshgrid(ax, ax, indexing="ij")
            k2 = torch.exp(-(xx ** 2 + yy ** 2) / (2 * self.sigma ** 2))
            k2 = k2 / k2.sum()
            f = torch.zeros((1, self.num_channels, D, D), dtype=torch.float32)
            f[:, :, :K, :K] = k2
            f = torch.roll(f, shifts=(-(K - 1) // 2, -(K - 1) // 2), dims=(2, 3))
            self._filter = f.to(self._device if torch.cuda.is_available() else "cpu")
        return self._filter

    def descriptor(self, B, H, W, device):
        key = str(device)
        if key not in self._taps:
            self._taps[key] = torch.from_numpy(self.taps_eff).to(device)
        d = _lib.PfDegradation(); d.kind = self.kind; d.ntaps = int(self.taps_eff.shape[0])
        d.taps = self._taps[key].data_ptr()
        return d

    def H(self, x):
        return self._apply(x, False)

    def H_adj(self, x):
        return self._apply(x, True)


def bicubic_taps(factor):
    """1-D factor of the reference's bicubic filter (utils.py:365-396): Keys cubic (a = -0.5) sampled at
    (k + 0.5)/factor - 2; the reference's normalised outer(w, w) equals outer(w/sum w, w/sum w)."""
    x = np.abs(np.arange(start=-2 * factor + 0.5, stop=2 * factor, step=1) / factor)
    a = -0.5
    w = ((a + 2) * x ** 3 - (a + 3) * x ** 2 + 1) * (x <= 1) + (a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a) * (x > 1) * (x < 2)
    return (w / w.sum()).astype(np.float32)


class Superresolution(Degradation):
    """reference pnpflow/degradations.py:92-127.  mode=None (main.py:165): H = x[..., ::sf, ::sf],
    H_adj = zero-fill.  mode="bicubic": circular convolution with the 4sf x 4sf bicubic filter (separable:
    two 1-D passes instead of the reference's FFTs), then decimation; H_adj = zero-fill, then the conjugate
    filter.  The dense (HW/sf^2, HW) matrix the reference builds in the constructor is only read by ot_ode
    (where diag(D D^T) = 1) and is not materialised here."""
    kind = _lib.PF_DEG_SUPERRESOLUTION

    def __init__(self, sf, dim_image, mode=None, device="cuda"):
        super().__init__()
        if mode not in (None, "bicubic"):
            raise NotImplementedError(f"Superresolution mode {mode!r}")
        self.sf, self.dim_image, self.mode = sf, dim_image, mode
        self._dm = None
        if mode == "bicubic":
            self.kind = _lib.PF_DEG_SR_FILTERED
            self.taps_host = bicubic_taps(sf)
            self._taps = {}

    def descriptor(self, B, H, W, device):
        d = _lib.PfDegradation(); d.kind = self.kind; d.sf = int(self.sf)
        if self.mode == "bicubic":
            key = str(device)
            if key not in self._taps:
                self._taps[key] = torch.from_numpy(self.taps_host).to(device)
            d.ntaps = int(self.taps_host.shape[0]); d.taps = self._taps[key].data_ptr()
        return d

    @property
    def downsampling_matrix(self):
        """The reference's attribute (degradations.py:110-111, utils.py:1124-1146; read by ot_ode.py:98): the
        (HW/sf^2, HW) 0/1 decimation matrix.  Built on first access only (268 MB at 256^2 / sf 4; the engine's own
        solvers never need it: diag(D D^T) = 1)."""
        if self._dm is None:
            D, sf = self.dim_image, self.sf
            Dl = D // sf
            rows = torch.arange(Dl * Dl)
            cols = (rows // Dl) * sf * D + (rows % Dl) * sf
            m = torch.zeros((Dl * Dl, D * D), dtype=torch.float32)
            m[rows, cols] = 1.0
            self._dm = m
        return self._dm

    def H(self, x):
        return self._apply(x, False)

    def H_adj(self, x):
        return self._apply(x, True, out_hw=(x.shape[2] * self.sf, x.shape[3] * self.sf))
