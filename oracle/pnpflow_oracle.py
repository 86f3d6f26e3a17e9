"""CPU oracle for the PnP-Flow restoration hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch fp32 functional ops, no nn.Module,
no reference import) of the algorithm on the path named by BASELINE.json's north_star:

    PNP_FLOW.solve_ip  ->  Degradation.H / H_adj  ->  UNet.forward

It is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product path (pnpflow_amd/) never imports anything from oracle/ and
fails loudly when the HIP extension is missing.

Parity pinning: the oracle is pinned against golden vectors generated in the build
container by importing the *real* reference from /root/reference (tools/make_golden.py;
fixtures in tests/golden/*.npz) and against the reference's single known-answer test
(pnpflow/tests/test_unit.py:14-20).  See tests/test_oracle_golden.py.

Every function cites the reference file:line (relative to /root/reference) it restates.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# U-Net velocity field  v_theta(x, t)            (pnpflow/models.py:24-495)
# --------------------------------------------------------------------------------------

def unet_config(input_channels: int, input_height: int, ch: int = 32,
                ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int = 6,
                attn_resolutions: Sequence[int] = (16, 8), output_channels: Optional[int] = None) -> dict:
    """Hyper-parameter record of `UNet.__init__` (pnpflow/models.py:302-334).

    `define_model` (pnpflow/utils.py:170-180) uses ch=32, ch_mult=(1,2,4,8), 6 blocks,
    attention at resolutions (16, 8)."""
    assert input_height % 2 ** (len(ch_mult) - 1) == 0  # models.py:334-335
    return dict(input_channels=input_channels, input_height=input_height, ch=ch,
                ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks,
                attn_resolutions=tuple(attn_resolutions),
                output_channels=input_channels if output_channels is None else output_channels)


def swish(x: torch.Tensor) -> torch.Tensor:
    """pnpflow/models.py:24-30 : sigmoid(x) * x."""
    return torch.sigmoid(x) * x


def group_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """pnpflow/models.py:33-38 : GroupNorm(32 groups, eps 1e-6, affine)."""
    return F.group_norm(x, 32, w, b, eps=1e-6)


def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """pnpflow/models.py:253-279.  t is used raw (in [0,1)), not scaled by 999
    (pnpflow/methods/pnp_flow.py:21)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    e = t.to(torch.float32)[:, None] * freq[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1))
    return e


def timestep_embedding(sd: Dict[str, torch.Tensor], t: torch.Tensor, ch: int) -> torch.Tensor:
    """pnpflow/models.py:282-299 : Linear(ch->4ch) . Swish . Linear(4ch->4ch)."""
    e = sinusoidal_embedding(t, ch)
    e = F.linear(e, sd["temb_net.main.0.weight"], sd["temb_net.main.0.bias"])
    e = swish(e)
    return F.linear(e, sd["temb_net.main.2.weight"], sd["temb_net.main.2.bias"])


def residual_block(sd, p: str, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
    """pnpflow/models.py:94-113."""
    h = swish(group_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = h + F.linear(swish(temb), sd[p + "temb_proj.weight"], sd[p + "temb_proj.bias"])[:, :, None, None]
    h = swish(group_norm(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"]))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "shortcut.weight") in sd:  # 1x1 conv when in_ch != out_ch (models.py:85-92)
        x = F.conv2d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
    return x + h


def self_attention(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """pnpflow/models.py:145-162 : single head, softmax over keys, scale C^-0.5."""
    B, C, H, W = x.shape
    h = group_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    q = F.conv2d(h, sd[p + "attn_q.weight"], sd[p + "attn_q.bias"]).view(B, C, H * W)
    k = F.conv2d(h, sd[p + "attn_k.weight"], sd[p + "attn_k.bias"]).view(B, C, H * W)
    v = F.conv2d(h, sd[p + "attn_v.weight"], sd[p + "attn_v.bias"]).view(B, C, H * W)
    attn = torch.bmm(q.permute(0, 2, 1), k) * (int(C) ** (-0.5))
    attn = torch.softmax(attn, dim=-1)
    h = torch.bmm(v, attn.permute(0, 2, 1)).view(B, C, H, W)
    h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + h


def unet_forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, t: torch.Tensor,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """pnpflow/models.py:442-495.  `taps`, if given, receives named intermediates
    (used to localise a parity failure to one block)."""
    nres, nlev = cfg["num_res_blocks"], len(cfg["ch_mult"])
    attn_res = cfg["attn_resolutions"]
    temb = timestep_embedding(sd, t, cfg["ch"])
    if taps is not None:
        taps["temb"] = temb
    hs: List[torch.Tensor] = [F.conv2d(x, sd["begin_conv.weight"], sd["begin_conv.bias"], padding=1)]
    for lvl in range(nlev):
        for blk in range(nres):
            p = f"down_modules.{lvl}.{lvl}a_{blk}a_block."
            h = residual_block(sd, p, hs[-1], temb)
            if h.shape[2] in attn_res:
                h = self_attention(sd, f"down_modules.{lvl}.{lvl}a_{blk}b_attn.", h)
            hs.append(h)
            if taps is not None:
                taps[f"down{lvl}_{blk}"] = h
        if lvl != nlev - 1:
            p = f"down_modules.{lvl}.{lvl}b_downsample."
            hs.append(F.conv2d(hs[-1], sd[p + "weight"], sd[p + "bias"], stride=2, padding=1))
            if taps is not None:
                taps[f"downsample{lvl}"] = hs[-1]
    h = hs[-1]
    h = residual_block(sd, "mid_modules.0.", h, temb)
    h = self_attention(sd, "mid_modules.1.", h)
    h = residual_block(sd, "mid_modules.2.", h, temb)
    if taps is not None:
        taps["mid"] = h
    for idx, lvl in enumerate(reversed(range(nlev))):
        for blk in range(nres + 1):
            p = f"up_modules.{idx}.{lvl}a_{blk}a_block."
            h = residual_block(sd, p, torch.cat([h, hs.pop()], dim=1), temb)
            if h.shape[2] in attn_res:
                h = self_attention(sd, f"up_modules.{idx}.{lvl}a_{blk}b_attn.", h)
            if taps is not None:
                taps[f"up{lvl}_{blk}"] = h
        if lvl != 0:
            p = f"up_modules.{idx}.{lvl}b_upsample.up_conv."
            h = F.interpolate(h, scale_factor=2, mode="nearest")  # models.py:41-47
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
            if taps is not None:
                taps[f"upsample{lvl}"] = h
    assert not hs
    h = swish(group_norm(h, sd["end_conv.0.weight"], sd["end_conv.0.bias"]))
    return F.conv2d(h, sd["end_conv.2.weight"], sd["end_conv.2.bias"], padding=1)


def unet_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """Names and shapes of the state_dict of `UNet` (pnpflow/models.py:337-436),
    derived from the hyper-parameters only.  Order = module registration order."""
    ch, mult, nres = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    nlev, attn_res = len(mult), cfg["attn_resolutions"]
    cin, cout_img, ht = cfg["input_channels"], cfg["output_channels"], cfg["input_height"]
    tch = 4 * ch
    shp: Dict[str, tuple] = {}

    def lin(p, i, o):
        shp[p + "weight"] = (o, i); shp[p + "bias"] = (o,)

    def conv(p, i, o, k):
        shp[p + "weight"] = (o, i, k, k); shp[p + "bias"] = (o,)

    def gn(p, c):
        shp[p + "weight"] = (c,); shp[p + "bias"] = (c,)

    def res(p, i, o):
        lin(p + "temb_proj.", tch, o); gn(p + "norm1.", i); conv(p + "conv1.", i, o, 3)
        gn(p + "norm2.", o); conv(p + "conv2.", o, o, 3)
        if i != o:
            conv(p + "shortcut.", i, o, 1)

    def attn(p, c):
        for n in ("attn_q.", "attn_k.", "attn_v.", "proj_out."):
            conv(p + n, c, c, 1)
        gn(p + "norm.", c)

    lin("temb_net.main.0.", ch, tch); lin("temb_net.main.2.", tch, tch)
    conv("begin_conv.", cin, ch, 3)
    chans, c, h = [ch], ch, ht
    for lvl in range(nlev):
        o = ch * mult[lvl]
        for blk in range(nres):
            res(f"down_modules.{lvl}.{lvl}a_{blk}a_block.", c, o)
            if h in attn_res:
                attn(f"down_modules.{lvl}.{lvl}a_{blk}b_attn.", o)
            chans.append(o); c = o
        if lvl != nlev - 1:
            conv(f"down_modules.{lvl}.{lvl}b_downsample.", o, o, 3)
            h //= 2; chans.append(o)
    res("mid_modules.0.", c, c); attn("mid_modules.1.", c); res("mid_modules.2.", c, c)
    for idx, lvl in enumerate(reversed(range(nlev))):
        o = ch * mult[lvl]
        for blk in range(nres + 1):
            res(f"up_modules.{idx}.{lvl}a_{blk}a_block.", c + chans.pop(), o)
            if h in attn_res:
                attn(f"up_modules.{idx}.{lvl}a_{blk}b_attn.", o)
            c = o
        if lvl != 0:
            conv(f"up_modules.{idx}.{lvl}b_upsample.up_conv.", o, o, 3)
            h *= 2
    assert not chans
    gn("end_conv.0.", c); conv("end_conv.2.", c, cout_img, 3)
    return shp


def synthetic_state_dict(cfg: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seed-fixed synthetic weights (no checkpoint is reachable offline).

    NOT the reference initialiser (which leaves conv2/proj_out/end_conv at gain 1e-10,
    models.py:84,131-137,432, so half the net would be untested): every tensor is drawn
    from its own numpy Philox stream keyed by (seed, name-hash) so the recipe is
    reproducible anywhere without torch RNG:  weights ~ U(-a, a), a = sqrt(3/fan_in);
    GN gamma ~ 1 + 0.1 U(-1,1); biases/GN beta ~ 0.05 U(-1,1)."""
    import zlib
    sd = {}
    for name, shape in unet_param_shapes(cfg).items():
        rng = np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))
        u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
        if name.endswith("weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            u *= np.float32(math.sqrt(3.0 / fan_in))
        elif name.endswith("weight"):
            u = np.float32(1.0) + np.float32(0.1) * u
        else:
            u *= np.float32(0.05)
        sd[name] = torch.from_numpy(u)
    return sd


# --------------------------------------------------------------------------------------
# Degradation operators H / H_adj                (pnpflow/degradations.py:6-127)
# --------------------------------------------------------------------------------------

def square_mask(x: torch.Tensor, half: int) -> torch.Tensor:
    """pnpflow/utils.py:327-336 (d = x.shape[2]//2 is used for BOTH axes)."""
    d = x.shape[2] // 2
    m = torch.ones_like(x)
    m[:, :, d - half:d + half, d - half:d + half] = 0
    return m * x


def random_mask_array(B: int, H: int, W: int, p: float) -> np.ndarray:
    """pnpflow/utils.py:357-359 : np.random.seed(42); binomial(1, 1-p, (B,H,W)) int64.
    (Uses a private RandomState -- identical stream, without clobbering the global one.)"""
    return np.random.RandomState(42).binomial(n=1, p=1 - p, size=(B, H, W))


def random_mask(x: torch.Tensor, p: float) -> torch.Tensor:
    """pnpflow/utils.py:353-361."""
    m = torch.from_numpy(random_mask_array(x.shape[0], x.shape[2], x.shape[3], p))
    return m.unsqueeze(1) * x


def gaussian_2d_kernel(sigma: float, size: int) -> torch.Tensor:
    """pnpflow/utils.py:273-280."""
    ax = torch.arange(-size // 2 + 1.0, size // 2 + 1.0)
    xx, yy = torch.meshgrid(ax, ax, indexing="ij")
    k = torch.exp(-(xx ** 2 + yy ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def gaussian_1d_taps(sigma: float, size: int) -> np.ndarray:
    """Separable factor of `gaussian_2d_kernel`: k2d = outer(g, g) with
    g = exp(-x^2/2s^2)/sum (exact up to fp32 rounding). float64."""
    ax = np.arange(-size // 2 + 1.0, size // 2 + 1.0)
    g = np.exp(-(ax ** 2) / (2 * sigma ** 2))
    return g / g.sum()


class Degradation:
    """pnpflow/degradations.py:6-12."""
    def H(self, x):
        raise NotImplementedError()

    def H_adj(self, x):
        raise NotImplementedError()


class Denoising(Degradation):
    """pnpflow/degradations.py:15-20."""
    def H(self, x):
        return x

    def H_adj(self, x):
        return x


class BoxInpainting(Degradation):
    """pnpflow/degradations.py:23-32."""
    def __init__(self, half_size_mask):
        self.half_size_mask = half_size_mask

    def H(self, x):
        return square_mask(x, self.half_size_mask)

    H_adj = H


class RandomInpainting(Degradation):
    """pnpflow/degradations.py:35-44."""
    def __init__(self, p):
        self.p = p

    def H(self, x):
        return random_mask(x, self.p)

    H_adj = H


# ---- cv2.line restated (opencv/modules/imgproc/src/drawing.cpp: ThickLine / FillConvexPoly / Line2 / Circle) -------------------
# OpenCV (`opencv-python`, unpinned in the reference's requirements.txt) is not installed here: PARITY UNPINNED.  This restatement is
# written separately from the product's (pnpflow_amd/cv_draw.py walks the edges incrementally; here every edge segment is a closed
# form x(row) = xs + dx * (row - row0) and the outline / circle points are generated as arrays).
_XS, _X1 = 16, 1 << 16


def _cdiv(a: int, b: int) -> int:
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _cv_line2_points(p1, p2, W, H):
    """Pixels of Line2 between two 16.16 points (clipped to the image as cv::clipLine does it)."""
    (x1, y1), (x2, y2) = p1, p2
    # cv::clipLine on the 16.16 rectangle [0, W << 16) x [0, H << 16): outcodes, then the y borders, then the x borders
    R, Bm = (W << _XS) - 1, (H << _XS) - 1
    code = lambda x, y: (x < 0) + (x > R) * 2 + (y < 0) * 4 + (y > Bm) * 8
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else Bm
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1)); y1 = a; c1 = (x1 < 0) + (x1 > R) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else Bm
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1)); y2 = a; c2 = (x2 < 0) + (x2 > R) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else R
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1)); x1 = a; c1 = 0
            if c2:
                a = 0 if c2 == 1 else R
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1)); x2 = a; c2 = 0
    if (c1 | c2) != 0:
        return []
    dx, dy = x2 - x1, y2 - y1
    ax, ay = abs(dx), abs(dy)
    pts = [((x2 + (_X1 >> 1)) >> _XS, (y2 + (_X1 >> 1)) >> _XS)]
    if ax > ay:
        if dx < 0:
            x1, x2, y1, y2, dy = x2, x1, y2, y1, -dy
        step = _cdiv(dy << _XS, ax | 1)
        n = ((x2 - x1) >> _XS) + 1
        xs = ((x1 + (_X1 >> 1)) >> _XS) + np.arange(n)
        ys = ((y1 + (_X1 >> 1)) + step * np.arange(n, dtype=np.int64)) >> _XS
    else:
        if dy < 0:
            x1, x2, y1, y2, dx = x2, x1, y2, y1, -dx
        step = _cdiv(dx << _XS, ay | 1)
        n = ((y2 - y1) >> _XS) + 1
        ys = ((y1 + (_X1 >> 1)) >> _XS) + np.arange(n)
        xs = ((x1 + (_X1 >> 1)) + step * np.arange(n, dtype=np.int64)) >> _XS
    pts += list(zip(xs.tolist(), ys.tolist()))
    return [(x, y) for x, y in pts if 0 <= x < W and 0 <= y < H]


def _cv_poly_rows(v):
    """{row: (x_left, x_right)} of FillConvexPoly's scan conversion for 16.16 vertices (LINE_8: both ends rounded with + 2^15)."""
    n = len(v)
    half = _X1 >> 1
    rows_of = [(p[1] + half) >> _XS for p in v]
    imin = min(range(n), key=lambda i: (v[i][1], i))
    y0, ymax = rows_of[imin], max(rows_of)

    def chain(di):
        """[(row0, row1_exclusive, xs, dx)] of one walker (di = +1 / -1 around the vertex list)"""
        segs, y, i0 = [], y0, imin
        for _ in range(n):
            i1 = (i0 + di) % n
            ty = rows_of[i1]
            if ty > y:
                xs, xe = v[i0][0], v[i1][0]
                segs.append((y, ty, xs, _cdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))))
                y = ty
            i0 = i1
        return segs
    a, b = chain(+1), chain(-1)

    def at(segs, row):
        for r0, r1, xs, dx in segs:
            if r0 <= row < r1:
                return xs + dx * (row - r0)
        r0, r1, xs, dx = segs[-1]
        return xs + dx * (row - r0)                       # the last row continues the last segment (the loop ends at ymax inclusive)
    out = {}
    last = min(a[-1][1], b[-1][1]) if a and b else y0
    for row in range(y0, min(ymax, last) + 1):
        if not a or not b:
            break
        if row >= a[-1][1] or row >= b[-1][1]:
            # a walker that has run out of vertices ends the fill (edges < 0) - except on the final row, where it keeps its slope
            if row > ymax:
                break
        xa, xb = at(a, row), at(b, row)
        lo, hi = (xb, xa) if xa > xb else (xa, xb)
        out[row] = ((lo + half) >> _XS, (hi + half) >> _XS)
    return out


def _cv_circle_runs(r):
    """[(dy, half-width)] runs of the filled midpoint circle (both signs of dy are drawn)"""
    runs, err, dx, dy, plus, minus = [], 0, r, 0, 1, (r << 1) - 1
    while dx >= dy:
        runs.append((dy, dx)); runs.append((dx, dy))
        dy += 1; err += plus; plus += 2
        if err > 0:
            err -= minus; dx -= 1; minus -= 2
    return runs


def cv2_thick_line(img: np.ndarray, p0, p1, thickness: int) -> np.ndarray:
    """cv2.line(img, p0, p1, 255, thickness), thickness > 1, LINE_8, on a (H, W) uint8 array (in place)."""
    H, W = img.shape
    x0, y0, x1, y1 = int(p0[0]) << _XS, int(p0[1]) << _XS, int(p1[0]) << _XS, int(p1[1]) << _XS
    ddx, ddy = (x0 - x1) / _X1, (y1 - y0) / _X1
    rr = ddx * ddx + ddy * ddy
    th = thickness << (_XS - 1)
    if rr > 2.220446049250313e-16:
        k = (th + (thickness & 1) * _X1 * 0.5) / math.sqrt(rr)
        dpx, dpy = int(round(ddy * k)), int(round(ddx * k))
        v = [(x0 + dpx, y0 + dpy), (x0 - dpx, y0 - dpy), (x1 - dpx, y1 - dpy), (x1 + dpx, y1 + dpy)]
        for i in range(4):
            for x, y in _cv_line2_points(v[i - 1], v[i], W, H):
                img[y, x] = 255
        for row, (xa, xb) in _cv_poly_rows(v).items():
            if 0 <= row < H and xb >= 0 and xa < W and xb >= xa:
                img[row, max(xa, 0):min(xb, W - 1) + 1] = 255
    rad = (th + (_X1 >> 1)) >> _XS
    for cx, cy in ((int(p0[0]), int(p0[1])), (int(p1[0]), int(p1[1]))):
        for dy, hw in _cv_circle_runs(rad):
            for yy in (cy - dy, cy + dy):
                if 0 <= yy < H and cx + hw >= 0 and cx - hw < W:
                    img[yy, max(cx - hw, 0):min(cx + hw, W - 1) + 1] = 255
    return img


def paintbrush_mask_array(B: int, H: int, W: int) -> np.ndarray:
    """pnpflow/utils.py:339-350 with MaskGenerator._generate_mask (:904-924): random.seed(42); per image 10 strokes
    (endpoints randint(W//2 +- 30), randint(H//2 +- 30), thickness randint(8, int((W+H)*0.08))) drawn with cv2.line (restated above,
    PARITY UNPINNED: cv2 is not installed here).  Keep-mask: 1 = observed, 0 = under a stroke."""
    import random
    rng = random.Random(42)
    size = int((W + H) * 0.08)
    out = np.ones((B, H, W), dtype=np.uint8)
    for b in range(B):
        img = np.zeros((H, W), dtype=np.uint8)
        for _ in range(10):
            x1 = rng.randint(W // 2 - 30, W // 2 + 30); x2 = rng.randint(W // 2 - 30, W // 2 + 30)
            y1 = rng.randint(H // 2 - 30, H // 2 + 30); y2 = rng.randint(H // 2 - 30, H // 2 + 30)
            cv2_thick_line(img, (x1, y1), (x2, y2), rng.randint(8, size))
        out[b][img != 0] = 0
    return out


class PaintbrushInpainting(Degradation):
    """pnpflow/degradations.py:47-52"""

    def H(self, x):
        m = torch.from_numpy(paintbrush_mask_array(x.shape[0], x.shape[2], x.shape[3]).astype(np.float32))
        return m[:, None] * x

    H_adj = H


class GaussianDeblurring(Degradation):
    """pnpflow/degradations.py:55-89, mode 'fft' : circular convolution via FFT."""
    def __init__(self, sigma_blur, kernel_size, mode="fft", num_channels=3, dim_image=128, device="cpu"):
        assert mode == "fft"
        self.sigma, self.kernel_size = sigma_blur, kernel_size
        self.kernel = gaussian_2d_kernel(sigma_blur, kernel_size)
        f = torch.zeros((1, num_channels, dim_image, dim_image))
        f[..., :kernel_size, :kernel_size] = self.kernel
        s = -(kernel_size - 1) // 2
        self.filter = torch.roll(f, shifts=(s, s), dims=(2, 3))

    def H(self, x):
        return torch.real(torch.fft.ifft2(torch.fft.fft2(x) * torch.fft.fft2(self.filter)))

    def H_adj(self, x):
        return torch.real(torch.fft.ifft2(torch.fft.fft2(x) * torch.conj(torch.fft.fft2(self.filter))))


def bicubic_filter(factor: int = 2) -> torch.Tensor:
    """pnpflow/utils.py:365-396: Keys bicubic (a=-0.5), size (4 factor)^2, normalised to sum 1."""
    x = np.abs(np.arange(start=-2 * factor + 0.5, stop=2 * factor, step=1) / factor)
    a = -0.5
    w = ((a + 2) * np.power(x, 3) - (a + 3) * np.power(x, 2) + 1) * (x <= 1)
    w += (a * np.power(x, 3) - 5 * a * np.power(x, 2) + 8 * a * x - 4 * a) * (x > 1) * (x < 2)
    w = np.outer(w, w)
    return torch.Tensor(w / np.sum(w)).unsqueeze(0).unsqueeze(0)


class Superresolution(Degradation):
    """pnpflow/degradations.py:92-127.  mode=None (the only one main.py:165 uses):
    H = x[..., ::sf, ::sf] (utils.py:302-310); H_adj = zero-fill (utils.py:283-299).
    mode="bicubic": circular FFT convolution with the rolled bicubic filter, then decimation (:117-120);
    H_adj = zero-fill then the conjugate filter (:125-127).
    The dense downsampling matrix (degradations.py:110-111) is only read by ot_ode."""
    def __init__(self, sf, dim_image, mode=None, device="cpu"):
        assert mode in (None, "bicubic")
        self.sf, self.mode = sf, mode
        if mode == "bicubic":
            k = bicubic_filter(sf)
            f = torch.zeros((1, 3, dim_image, dim_image))
            f[..., :k.shape[-1], :k.shape[-1]] = k
            s = -(k.shape[-1] - 1) // 2                  # (-(K-1))//2 : floor division of the negative number, as :108
            self.filter = torch.roll(f, shifts=(s, s), dims=(2, 3))

    def _down(self, x):
        return x[..., ::self.sf, ::self.sf]

    def _up(self, x):
        z = torch.zeros((x.shape[0], x.shape[1], x.shape[2] * self.sf, x.shape[3] * self.sf), dtype=x.dtype)
        z[..., ::self.sf, ::self.sf] = x
        return z

    def H(self, x):
        if self.mode is None:
            return self._down(x)
        return self._down(torch.real(torch.fft.ifft2(torch.fft.fft2(x) * torch.fft.fft2(self.filter))))

    def H_adj(self, x):
        if self.mode is None:
            return self._up(x)
        return torch.real(torch.fft.ifft2(torch.fft.fft2(self._up(x)) * torch.conj(torch.fft.fft2(self.filter))))


def make_degradation(problem: str, dim_image: int, num_channels: int = 3, noise_type: str = "gaussian"):
    """Problem table of main.py:120-179  ->  (degradation, sigma_noise)."""
    lap = noise_type == "laplace"
    if problem == "denoising":
        return Denoising(), (0.3 if lap else 0.2)
    if problem == "inpainting":
        half = {128: 20, 256: 40}[dim_image]
        return BoxInpainting(half), (0.3 if lap else 0.05)
    if problem == "random_inpainting":
        return RandomInpainting(0.7), (0.3 if lap else 0.01)
    if problem == "superresolution":
        sf = {128: 2, 256: 4}[dim_image]
        return Superresolution(sf, dim_image), (0.3 if lap else 0.05)
    if problem == "gaussian_deblurring_FFT":
        sb = {128: 1.0, 256: 3.0}[dim_image]
        return GaussianDeblurring(sb, 61, "fft", num_channels, dim_image), (0.3 if lap else 0.05)
    raise ValueError(problem)


# --------------------------------------------------------------------------------------
# PnP-Flow solver                                 (pnpflow/methods/pnp_flow.py:10-188)
# --------------------------------------------------------------------------------------

def learning_rate_strat(lr: float, t: torch.Tensor, gamma_style: str, alpha: float) -> torch.Tensor:
    """pnpflow/methods/pnp_flow.py:29-37."""
    t = t.view(-1, 1, 1, 1)
    if gamma_style == "1_minus_t":
        return lr * (1 - t)
    if gamma_style == "sqrt_1_minus_t":
        return lr * torch.sqrt(1 - t)
    if gamma_style == "alpha_1_minus_t":
        return lr * (1 - t) ** alpha
    return lr * torch.ones_like(t)  # 'constant' and unknown styles


def pnp_flow_restore(model: Callable, degradation: Degradation, noisy_img: torch.Tensor, sigma_noise: float, *,
                     steps: int, num_samples: int, lr_pnp: float = 1.0, gamma_style: str = "alpha_1_minus_t",
                     alpha: float = 1.0, noise_fn: Optional[Callable] = None,
                     record: Optional[Callable] = None, noise_type: str = "gaussian") -> torch.Tensor:
    """Inner loop of PNP_FLOW.solve_ip for one batch, gaussian noise
    (pnpflow/methods/pnp_flow.py:60-62, 93, 102-121).

    model(x, t)->v;  noise_fn(iteration, sample, like)->eps replaces torch.randn_like
    (pnp_flow.py:48) so that trajectories are comparable across devices;  record(it, x)
    is called after each outer iteration."""
    H, H_adj = degradation.H, degradation.H_adj
    lr = (sigma_noise ** 2 if noise_type == "gaussian" else sigma_noise) * lr_pnp      # :60-66
    delta = 1.0 / steps
    x = H_adj(torch.ones_like(noisy_img))                            # :93
    if noise_fn is None:
        noise_fn = lambda it, s, like: torch.randn_like(like)
    with torch.no_grad():
        for it in range(int(steps)):
            t1 = torch.ones(len(x)) * delta * it                     # :107-108
            lr_t = learning_rate_strat(lr, t1, gamma_style, alpha)   # :109
            if noise_type == "gaussian":
                grad = H_adj(H(x) - noisy_img) / sigma_noise ** 2        # :39-41
            else:                                                        # laplace, :42-43
                r = H(x) - noisy_img
                grad = H_adj(2 * torch.heaviside(r, torch.zeros_like(r)) - 1) / sigma_noise
            z = x - lr_t * grad                                          # :111-112
            x_new = torch.zeros_like(x)
            tv = t1.view(-1, 1, 1, 1)
            for s in range(num_samples):                             # :114-118
                z_tilde = tv * z + noise_fn(it, s, z) * (1 - tv)     # :47-48
                x_new += z_tilde + (1 - tv) * model(z_tilde, t1)     # :50-52
            x_new /= num_samples                                     # :120
            x = x_new
            if record is not None:
                record(it, x)
    return x


def make_measurement(clean: torch.Tensor, degradation: Degradation, sigma_noise: float, batch: int,
                     noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pnpflow/methods/pnp_flow.py:77-80 : y = H(clean) + sigma * randn (seed = batch index).
    `noise` overrides the draw (multi-GPU shards slice one global draw)."""
    y = degradation.H(clean.clone())
    if noise is None:
        torch.manual_seed(batch)
        noise = torch.randn_like(y)
    return y + noise * sigma_noise


# --------------------------------------------------------------------------------------
# Metric                                           (pnpflow/utils.py:560-577, 594-625)
# --------------------------------------------------------------------------------------

def postprocess(img: torch.Tensor) -> torch.Tensor:
    """pnpflow/utils.py:560-577 for model in {ot,...}: (img+1)/2, no clamp.
    (The non-afhq branch is Normalize(mean=-1, std=2) == (x+1)/2 as well.)"""
    return (img + 1) / 2


def psnr_per_image(rec: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """torchmetrics peak_signal_noise_ratio(data_range=1.0, dim=(1,2,3), reduction=None)
    restated (torchmetrics is absent offline; its published formula):
    10*log10(data_range^2 / mean_{c,h,w}((a-b)^2)) per image (pnpflow/utils.py:610)."""
    a, b = postprocess(rec), postprocess(clean)
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return 10.0 * torch.log10(1.0 / mse)


def ssim_per_image(rec: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """SSIM of postprocess(rec) vs postprocess(clean) per image as ignite.metrics.SSIM(data_range=1.0) computes it for the
    reference (pnpflow/utils.py:780-802): 11x11 Gaussian window (sigma 1.5), k1 0.01, k2 0.03, reflect padding, per-channel
    filtering of [x, y, x*x, y*y, x*y], map averaged over (C,H,W) in float64.  PARITY UNPINNED: ignite is not installed in
    the build container; this follows ignite's published algorithm (ignite/metrics/ssim.py, update())."""
    x, y = postprocess(rec).float(), postprocess(clean).float()
    C = x.shape[1]
    k = torch.linspace(-5.0, 5.0, steps=11)
    g = torch.exp(-0.5 * (k / 1.5).pow(2)); g = g / g.sum()
    kernel = (g.unsqueeze(1) @ g.unsqueeze(0)).expand(C, 1, 11, 11)
    c1, c2 = (0.01 * 1.0) ** 2, (0.03 * 1.0) ** 2
    xp = torch.nn.functional.pad(x, [5, 5, 5, 5], mode="reflect"); yp = torch.nn.functional.pad(y, [5, 5, 5, 5], mode="reflect")
    outs = torch.nn.functional.conv2d(torch.cat([xp, yp, xp * xp, yp * yp, xp * yp]), kernel, groups=C)
    B = x.shape[0]
    o = [outs[i * B:(i + 1) * B] for i in range(5)]
    mxx, myy, mxy = o[0].pow(2), o[1].pow(2), o[0] * o[1]
    sxx, syy, sxy = o[2] - mxx, o[3] - myy, o[4] - mxy
    a1, a2, b1, b2 = 2 * mxy + c1, 2 * sxy + c2, mxx + myy + c1, sxx + syy + c2
    return torch.mean((a1 * a2) / (b1 * b2), (1, 2, 3), dtype=torch.float64)


def psnr_batch(rec: torch.Tensor, clean: torch.Tensor) -> float:
    """torchmetrics default reduction 'elementwise_mean' over the per-image values."""
    return float(psnr_per_image(rec, clean).mean())


# --------------------------------------------------------------------------------------
# Engine RNG restatement (build-defined; no reference counterpart: the reference draws
# torch.randn_like on the device generator, pnp_flow.py:48, which is not reproducible
# across devices).  Philox4x32-10 + Box-Muller exactly as csrc/pointwise.hip does it.
# --------------------------------------------------------------------------------------

_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr: (n,4) uint32, key: (2,) uint32 -> (n,4) uint32 (Salmon et al. 2011)."""
    c = ctr.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(10):
        p0 = _PH_M0 * c[:, 0].astype(np.uint64)
        p1 = _PH_M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        with np.errstate(over="ignore"):
            k0 = np.uint32(k0 + _PH_W0); k1 = np.uint32(k1 + _PH_W1)
    return c


def engine_normal(n: int, seed: int, stream: int, offset: int = 0) -> np.ndarray:
    """normal numbers [offset, offset+n) of the stream: number 4q+j comes from Philox counter (q_lo, q_hi, stream_lo,
    stream_hi), key (seed_lo, seed_hi); u = (r + 0.5) * 2^-32; Box-Muller pairs (0,1) and (2,3):
    z0 = sqrt(-2 ln u0) cos(2 pi u1), z1 = sqrt(-2 ln u0) sin(2 pi u1).  `offset` = a shard's first element of the
    global batch's flat noise tensor (pf_pnp_params.elem_offset)."""
    q_lo = offset // 4
    nq = (offset + n + 3) // 4 - q_lo
    ctr = np.zeros((nq, 4), dtype=np.uint32)
    qs = np.arange(q_lo, q_lo + nq, dtype=np.uint64)
    ctr[:, 0] = (qs & np.uint64(0xFFFFFFFF)).astype(np.uint32); ctr[:, 1] = (qs >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(stream & 0xFFFFFFFF); ctr[:, 3] = np.uint32((stream >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32))
    u = (r.astype(np.float64) + 0.5) * (2.0 ** -32)
    rad0, rad1 = np.sqrt(-2.0 * np.log(u[:, 0])), np.sqrt(-2.0 * np.log(u[:, 2]))
    a0, a1 = 2.0 * np.pi * u[:, 1], 2.0 * np.pi * u[:, 3]
    z = np.stack([rad0 * np.cos(a0), rad0 * np.sin(a0), rad1 * np.cos(a1), rad1 * np.sin(a1)], axis=1)
    return z.reshape(-1)[offset - 4 * q_lo:offset - 4 * q_lo + n].astype(np.float32)


# --------------------------------------------------------------------------------------
# OT-ODE solver                                   (pnpflow/methods/ot_ode.py:9-213)
# --------------------------------------------------------------------------------------

def unet_vjp(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, t: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """J^T vec of z -> v_theta(z, t) at z = x, as torch.autograd.functional.vjp computes it
    (pnpflow/methods/ot_ode.py:137-138)."""
    xx = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        out = unet_forward(sd, cfg, xx, t)
        (g,) = torch.autograd.grad(out, xx, grad_outputs=vec)
    return g.detach()


def ot_ode_solution(problem: str, d: torch.Tensor, degradation: Degradation, x_like: torch.Tensor, t1: torch.Tensor,
                    sigma_noise: float, delta: float, iteration: int) -> torch.Tensor:
    """Closed-form solve of (r_t^2 H H^T + sigma^2 I) sol = d  (pnpflow/methods/ot_ode.py:72, 81-106).
    The superresolution branch reproduces the reference's operator-precedence quirk
    `delta * iteration**2` (ot_ode.py:96)."""
    rt_squared = ((1 - t1) ** 2 / ((1 - t1) ** 2 + t1 ** 2)).view(-1, 1, 1, 1)
    if problem in ("inpainting", "random_inpainting", "paintbrush_inpainting"):
        mask = degradation.H(torch.ones_like(x_like))
        return 1 / (mask * rt_squared + sigma_noise ** 2) * d          # reciprocal-then-multiply, as ot_ode.py:84-86
    if problem == "denoising":
        return d / (rt_squared + sigma_noise ** 2)
    if problem == "superresolution":
        rt2 = torch.tensor((1 - delta * iteration) ** 2 / ((1 - delta * iteration) ** 2 + delta * iteration ** 2))
        # diag(D D^T) = 1 for the decimation matrix (utils.py:1124-1146)
        return (1 / (rt2 + torch.tensor(sigma_noise) ** 2)) * d
    if problem == "gaussian_deblurring_FFT":                            # ot_ode.py:108-117 (sol stays complex; H_adj takes the real part)
        fft_kernel = torch.fft.fft2(degradation.filter)
        inv = rt_squared * fft_kernel * torch.conj(fft_kernel) + sigma_noise ** 2
        return torch.fft.ifft2(torch.fft.fft2(d) / inv)
    # generic operator (ot_ode.py:118-128): per image, GMRES(max_iter=100) on C z = r_t^2 H(H_adj(z)) + sigma^2 z
    sol = torch.zeros_like(d)
    for i in range(d.shape[0]):
        def c_ope(z, i=i):
            zz = z.reshape(d.shape[1:]).unsqueeze(0)
            return (rt_squared[i].unsqueeze(0) * degradation.H(degradation.H_adj(zz)) + sigma_noise ** 2 * zz).reshape(-1)
        sol[i] = gmres(c_ope, d[i].reshape(-1), max_iter=100).reshape(d[i].shape)
    return sol


def gmres(avp: Callable, b: torch.Tensor, max_iter: int, tol: float = 1e-6, atol: float = 1e-6) -> torch.Tensor:
    """pnpflow/utils.py:972-1109 `GMRES` with x0 = 0: Arnoldi by modified Gram-Schmidt, the Hessenberg matrix reduced by Givens
    rotations as it grows, stop when |residual| < tol*|b| or < atol, then the triangular solve and x = V y.  (The reference returns
    `b` itself when |b| < 1e-8 or max_iter == 0, :996-997.)"""
    bnorm = torch.norm(b)
    if max_iter == 0 or bnorm < 1e-8:
        return b
    eps = torch.finfo(b.dtype).eps
    unit = lambda v: ((v / torch.norm(v)) if torch.norm(v) > eps else torch.zeros_like(v), torch.norm(v))     # _safe_normalize (:1055)
    v0, rnorm = unit(b)
    beta = torch.zeros(max_iter + 1); beta[0] = rnorm
    V = [v0]
    Hm = torch.zeros((max_iter + 1, max_iter + 1)); cs = torch.zeros(max_iter); ss = torch.zeros(max_iter)
    j = 0
    for j in range(max_iter):
        w = avp(V[j])
        for i in range(j + 1):                                           # arnoldi (:1067-1082)
            Hm[i, j] = torch.dot(w, V[i]); w = w - Hm[i, j] * V[i]
        vn, wn = unit(w); Hm[j + 1, j] = wn; V.append(vn)
        for i in range(j):                                               # apply_given_rotation (:1098-1109)
            tmp = cs[i] * Hm[i, j] - ss[i] * Hm[i + 1, j]
            Hm[i + 1, j] = cs[i] * Hm[i + 1, j] + ss[i] * Hm[i, j]
            Hm[i, j] = tmp
        r = torch.sqrt(Hm[j, j] * Hm[j, j] + Hm[j + 1, j] * Hm[j + 1, j])
        cs[j], ss[j] = Hm[j, j] / r, -Hm[j + 1, j] / r                    # cal_rotation (:1085-1095)
        Hm[j, j] = cs[j] * Hm[j, j] - ss[j] * Hm[j + 1, j]; Hm[j + 1, j] = 0
        beta[j + 1] = ss[j] * beta[j]; beta[j] = cs[j] * beta[j]
        res = torch.abs(beta[j + 1])
        if res < tol * bnorm or res < atol:
            break
    y = torch.linalg.solve_triangular(Hm[0:j + 1, 0:j + 1], beta[0:j + 1].unsqueeze(-1), upper=True)
    return torch.stack(V[:-1], dim=0).T @ y.squeeze(-1)


def ot_ode_restore(model: Callable, vjp: Callable, degradation: Degradation, problem: str, noisy_img: torch.Tensor,
                   sigma_noise: float, *, steps: int, start_time: float, gamma: str = "constant",
                   init_noise: Optional[torch.Tensor] = None, record: Optional[Callable] = None) -> torch.Tensor:
    """Loop of OT_ODE.solve_ip for one batch (pnpflow/methods/ot_ode.py:49-52, 63-147).
    model(x,t)->v; vjp(x,t,vec)->J^T vec; init_noise replaces the randn_like of `initialization` (:27-28)."""
    H, H_adj = degradation.H, degradation.H_adj
    hy = H_adj(noisy_img.clone())
    if init_noise is None:
        init_noise = torch.randn_like(hy)
    x = start_time * hy + (1 - start_time) * init_noise                       # :27-28, :50-52
    delta = 1 / steps
    for iteration in range(int(steps * start_time), int(steps)):             # :63
        with torch.no_grad():
            t1 = torch.ones(len(x)) * delta * iteration                      # :69-70
            vt = model(x, t1)                                                # :71
            x1_hat = x + (1 - t1.view(-1, 1, 1, 1)) * vt                     # :74
            d = noisy_img - H(x1_hat)                                        # :77
            sol = ot_ode_solution(problem, d, degradation, x, t1, sigma_noise, delta, iteration)
            vec = H_adj(sol)                                                 # :130
        t = t1.view(-1, 1, 1, 1)
        gam = 1 if gamma == "constant" else torch.sqrt(t / (t ** 2 + (1 - t) ** 2))   # :133-136
        g = vjp(x, t1, vec)                                                  # :137-138
        with torch.no_grad():
            g = vec + (1 - t) * g                                            # :141
            ratio = (1 - t) / t                                              # :143
            x = x + delta * (vt + ratio * gam * g)                           # :144-147
        if record is not None:
            record(iteration, x)
    return x


# --------------------------------------------------------------------------------------
# the reference's native ops (NCSN++ "rectified" net)         pnpflow/image_generation/op/
# --------------------------------------------------------------------------------------

def upfirdn2d(inp: torch.Tensor, kernel: torch.Tensor, up_x: int = 1, up_y: int = 1, down_x: int = 1, down_y: int = 1,
              pad_x0: int = 0, pad_x1: int = 0, pad_y0: int = 0, pad_y1: int = 0) -> torch.Tensor:
    """op/upfirdn2d.py:142-187 (`upfirdn2d_native`, the pure-torch definition of the CUDA kernel): zero-insert upsampling,
    padding (negative = crop), correlation with the FLIPPED kernel (= true convolution), decimation; (N,C,H,W) -> (N,C,Ho,Wo)."""
    N, C, in_h, in_w = inp.shape
    kh, kw = kernel.shape
    x = inp.reshape(N * C, in_h, 1, in_w, 1)
    x = torch.nn.functional.pad(x, [0, up_x - 1, 0, 0, 0, up_y - 1])                   # zero insertion            (:152-154)
    x = x.reshape(N * C, in_h * up_y, in_w * up_x)
    x = torch.nn.functional.pad(x, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])     # (:156-158)
    x = x[:, max(-pad_y0, 0): x.shape[1] - max(-pad_y1, 0), max(-pad_x0, 0): x.shape[2] - max(-pad_x1, 0)]   # (:159-164)
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)                                   # (:170)
    out = torch.nn.functional.conv2d(x.unsqueeze(1), w)[:, 0]                           # (:171)
    out = out[:, ::down_y, ::down_x]                                                    # (:179)
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    return out.reshape(N, C, out_h, out_w)


def fir_kernel_2d(k, gain: float = 1.0) -> torch.Tensor:
    """models/up_or_down_sampling.py:191-198 `_setup_kernel`: outer product of a separable filter, normalised to sum 1."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    return torch.from_numpy(k * gain)


def upsample_2d(x: torch.Tensor, k=None, factor: int = 2, gain: float = 1.0) -> torch.Tensor:
    """models/up_or_down_sampling.py:205-230"""
    k = [1] * factor if k is None else k
    kk = fir_kernel_2d(k, gain * factor ** 2)
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, up_x=factor, up_y=factor, pad_x0=(p + 1) // 2 + factor - 1, pad_x1=p // 2, pad_y0=(p + 1) // 2 + factor - 1, pad_y1=p // 2)


def downsample_2d(x: torch.Tensor, k=None, factor: int = 2, gain: float = 1.0) -> torch.Tensor:
    """models/up_or_down_sampling.py:233-259"""
    k = [1] * factor if k is None else k
    kk = fir_kernel_2d(k, gain)
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, down_x=factor, down_y=factor, pad_x0=(p + 1) // 2, pad_x1=p // 2, pad_y0=(p + 1) // 2, pad_y1=p // 2)


def fused_bias_act(x: torch.Tensor, bias: Optional[torch.Tensor], ref: Optional[torch.Tensor], act: int, grad: int, alpha: float,
                   scale: float) -> torch.Tensor:
    """op/fused_bias_act_kernel.cu:19-49: y = act(x + b[channel]) * scale (channel = dim 1); act 1 linear, 3 leaky ReLU;
    grad 1 takes the sign from `ref`, grad 2 is identically zero."""
    if bias is not None:
        x = x + bias.view(1, -1, *([1] * (x.ndim - 2)))
    if grad == 2:
        y = torch.zeros_like(x)
    elif act == 3:
        sel = x if grad == 0 else ref
        y = torch.where(sel > 0, x, x * alpha)
    else:
        y = x
    return y * scale


def fused_leaky_relu(x: torch.Tensor, bias: torch.Tensor, negative_slope: float = 0.2, scale: float = 2 ** 0.5) -> torch.Tensor:
    """op/fused_act.py:84-96 (the GPU branch: the reference's CPU branch hard-codes the slope 0.2)."""
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


# ---- LPIPS (AlexNet, v0.1) - third-party: `lpips` (unpinned in the reference's requirements.txt) on torchvision's AlexNet ----------
# Neither package nor the weight files are installed here: PARITY UNPINNED.  Restated from the published definition
# (lpips/lpips.py: LPIPS.forward, ScalingLayer, NetLinLayer, normalize_tensor, spatial_average; torchvision.models.alexnet.features).
LPIPS_CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))


def synthetic_lpips_state_dict(seed: int = 0) -> dict:
    """Seed-fixed stand-in for the two published checkpoints, under THEIR key names: torchvision alexnet `features.N.{weight,bias}`
    and lpips `lin{k}.model.1.weight` ([1, C, 1, 1], non-negative like the trained heads)."""
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for idx, ci, co, k, s_, p in LPIPS_CONVS:
        sd[f"features.{idx}.weight"] = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        sd[f"features.{idx}.bias"] = torch.randn(co, generator=g) * 0.05
    for kk, (_, _, co, _, _, _) in enumerate(LPIPS_CONVS):
        sd[f"lin{kk}.model.1.weight"] = torch.rand(1, co, 1, 1, generator=g) * (2.0 / co)
    return sd


def lpips_forward(sd: dict, in0: torch.Tensor, in1: torch.Tensor, normalize: bool = False) -> torch.Tensor:
    """lpips.LPIPS(net='alex', version='0.1', lpips=True, spatial=False).forward(in0, in1, normalize=normalize).flatten()"""
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift = torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1); scale = torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1)

    def feats(x):
        x = (x - shift) / scale
        out = []
        for idx, ci, co, k, s_, p in LPIPS_CONVS:
            if idx in (3, 6):
                x = F.max_pool2d(x, kernel_size=3, stride=2)
            x = F.relu(F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], stride=s_, padding=p))
            out.append(x)
        return out
    f0, f1 = feats(in0.float()), feats(in1.float())
    val = 0
    for kk in range(5):
        n0 = f0[kk] / (torch.sqrt(torch.sum(f0[kk] ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1[kk] / (torch.sqrt(torch.sum(f1[kk] ** 2, dim=1, keepdim=True)) + 1e-10)
        d = (n0 - n1) ** 2
        val = val + F.conv2d(d, sd[f"lin{kk}.model.1.weight"]).mean(dim=(2, 3), keepdim=True)
    return val.flatten()
