"""CPU checks (fp64, torch) of the two exact algebraic rewrites the engine applies to the reference's module graph in round 6 - the identities themselves,
independent of any kernel; the GPU A/B tests (tests/test_gpu_parity.py: test_upsampling_conv_phase_form_is_fp32_equivalent,
test_attention_output_projection_folded_into_the_value_projection) hold the engine's implementation of them against the unrewritten path and the goldens.

  * Upsample (pnpflow/models.py:41-47: nn.Upsample(scale_factor=2, mode='nearest') then a 3x3 conv, padding 1) = four 2x2 convs of the SOURCE image, one per
    output phase (dy, dx), whose tap (ty, tx) is the sum of the 3x3 weights over rows R(dy, ty) x columns R(dx, tx) with R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1},
    R(1,1) = {2}; the window of phase d starts at source offset d - 1 (engine.hip phase_weight, conv_dma.hip UP = 2).
  * SelfAttention (pnpflow/models.py:145-162): x + proj_out(bmm(v, attn^T)) = x + bmm(v', attn^T) + bp with v' = (Wp Wv) h + Wp bv (engine.hip attn_block).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

R = {(0, 0): (0, 1), (0, 1): (1, 3), (1, 0): (0, 2), (1, 1): (2, 3)}      # (d, t) -> [first, one past last) index of the 3x3 kernel


def phase_weights(w):
    """w [O][I][3][3] -> [dy][dx] -> [O][I][2][2]"""
    out = {}
    for dy in range(2):
        for dx in range(2):
            p = torch.zeros(w.shape[0], w.shape[1], 2, 2, dtype=w.dtype)
            for ty in range(2):
                for tx in range(2):
                    y0, y1 = R[(dy, ty)]; x0, x1 = R[(dx, tx)]
                    p[:, :, ty, tx] = w[:, :, y0:y1, x0:x1].sum(dim=(2, 3))
            out[(dy, dx)] = p
    return out


@pytest.mark.parametrize("shape", [(2, 5, 7, 8, 8), (1, 3, 4, 5, 9), (3, 8, 8, 1, 1), (1, 2, 3, 2, 6)])
def test_nearest_upsample_then_conv3x3_equals_four_phase_convs_of_the_source(shape):
    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64); b = torch.randn(Co, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    out = torch.empty_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))                       # the zero border of the source IS the zero padding of the upsampled image
    for (dy, dx), pw in phase_weights(w).items():
        # output (2i + dy, 2j + dx) = sum_t pw[t] * x[i + ty + dy - 1, j + tx + dx - 1]: a valid 2x2 correlation of the padded source from offset (dy, dx)
        out[:, :, dy::2, dx::2] = F.conv2d(xp[:, :, dy:dy + H + 1, dx:dx + W + 1], pw, b)
    assert float((out - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    # 16 instead of 36 multiply-adds per source pixel and channel pair
    assert sum(int(pw[0, 0].numel()) for pw in phase_weights(w).values()) == 16


@pytest.mark.parametrize("shape", [(2, 8, 4, 4), (1, 16, 3, 5)])
def test_proj_out_folds_into_the_value_projection(shape):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(12)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    h, x = rnd(B, C, H * W), rnd(B, C, H * W)            # h = GroupNorm(x) in the module; any tensor serves the identity
    wq, wk, wv, wp = (rnd(C, C) for _ in range(4)); bq, bk, bv, bp = (rnd(C) for _ in range(4))
    q = wq @ h + bq[:, None]; k = wk @ h + bk[:, None]; v = wv @ h + bv[:, None]
    attn = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (C ** -0.5), dim=-1)
    ref = x + (wp @ torch.bmm(v, attn.permute(0, 2, 1)) + bp[:, None])
    v2 = (wp @ wv) @ h + (wp @ bv)[:, None]
    out = x + torch.bmm(v2, attn.permute(0, 2, 1)) + bp[:, None]      # the rows of attn sum to 1, so Wp bv passes through the average unchanged
    assert float((out - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(2, 4, 4, 6, 6), (1, 3, 3, 5, 8)])
def test_adjoint_of_upsample_conv_is_four_phase_convs_of_the_strided_gradient(shape):
    """engine.hip build_backward, TP_UP: dx[i, j] = sum over phases (dy, dx) of a 2x2 conv of g[2i + dy][2j + dx] with the transposed phase weights, window rows
    u + (1 - dy) of the 3x3 neighbourhood, tap u carrying the forward tap 1 - u.  Reference: autograd through interpolate + conv2d."""
    B, Ci, Co, H, W = shape
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(B, Ci, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, generator=gen, dtype=torch.float64)
    g = torch.randn(B, Co, 2 * H, 2 * W, generator=gen, dtype=torch.float64)
    F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=1).backward(g)
    ref = x.grad
    out = torch.zeros_like(ref)
    for (dy, dx), pw in phase_weights(w).items():
        gp = F.pad(g[:, :, dy::2, dx::2], (1, 1, 1, 1))                       # the strided view, zero outside
        v = pw.flip(2, 3).transpose(0, 1)                                     # [Ci][Co][u][v] = pw[co][ci][1 - u][1 - v]
        oy, ox = 1 - dy, 1 - dx
        out += F.conv2d(gp[:, :, oy:oy + H + 1, ox:ox + W + 1], v)
    assert float((out - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(2, 4, 4, 8, 8), (1, 3, 5, 6, 10)])
def test_adjoint_of_stride2_conv_per_fine_phase(shape):
    """engine.hip build_backward, TP_DOWN: fine pixel (2a + py, 2b + px) receives tap ky of output row i only when 2i + ky - 1 = 2a + py, i.e. py = 0: (ky, i) = (1, a);
    py = 1: (0, a + 1), (2, a) - so each fine phase of dx is a conv of the coarse gradient with 1 / 2 / 2 / 4 taps inside rows {a, a + 1} x columns {b, b + 1}."""
    B, Ci, Co, H, W = shape                     # H, W: the fine (input) size, even
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(B, Ci, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, generator=gen, dtype=torch.float64)
    g = torch.randn(B, Co, H // 2, W // 2, generator=gen, dtype=torch.float64)
    F.conv2d(x, w, None, stride=2, padding=1).backward(g)
    ref = x.grad
    KOF = {(0, 0): 1, (0, 1): None, (1, 0): 2, (1, 1): 0}
    out = torch.zeros_like(ref)
    gp = F.pad(g, (0, 1, 0, 1))                                               # rows a, a + 1 / columns b, b + 1: one zero row / column past the end
    taps = 0
    for py in range(2):
        for px in range(2):
            v = torch.zeros(Ci, Co, 2, 2, dtype=torch.float64)
            for u in range(2):
                for t in range(2):
                    ky, kx = KOF[(py, u)], KOF[(px, t)]
                    if ky is not None and kx is not None:
                        v[:, :, u, t] = w[:, :, ky, kx].transpose(0, 1); taps += 1
            out[:, :, py::2, px::2] = F.conv2d(gp, v)
    assert taps == 9
    assert float((out - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
