"""`fused_leaky_relu` / `FusedLeakyReLU` with the API of pnpflow/image_generation/op/fused_act.py (reference :70-96) on the
gfx950 kernel of csrc/fir_ops.hip (pf_fused_bias_act).  Inference only."""
from __future__ import annotations

import torch

from ... import _lib


def fused_bias_act(input, bias=None, refer=None, act=3, grad=0, alpha=0.2, scale=1.0):
    """The op the reference binds (fused_bias_act_kernel.cu:52-99): act 1 linear / 3 leaky ReLU, grad 0 / 1 / 2."""
    if not input.is_cuda:
        raise _lib.PnpFlowHipError("fused_bias_act needs a GPU tensor (there is no CPU path)")
    lib = _lib.load()
    x = input.contiguous().float()
    out = torch.empty_like(x)
    b = bias.contiguous().float() if bias is not None and bias.numel() else None
    r = refer.contiguous().float() if refer is not None and refer.numel() else None
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.shape[i]
    _lib.check(lib.pf_fused_bias_act(x.data_ptr(), b.data_ptr() if b is not None else None, r.data_ptr() if r is not None else None,
                                     out.data_ptr(), x.numel(), step_b, b.numel() if b is not None else 1, act, grad, float(alpha), float(scale),
                                     _lib.current_stream_ptr()), None, "pf_fused_bias_act")
    return out


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)


class FusedLeakyReLU:
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        self.bias = torch.zeros(channel)
        self.negative_slope = negative_slope
        self.scale = scale

    def __call__(self, input):
        return fused_leaky_relu(input, self.bias.to(input.device), self.negative_slope, self.scale)

    forward = __call__
