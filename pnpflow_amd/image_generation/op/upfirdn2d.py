"""`upfirdn2d` with the API of pnpflow/image_generation/op/upfirdn2d.py (reference :128-139): the reference JIT-compiles a CUDA
kernel (op/upfirdn2d_kernel.cu) and falls back to a pure-torch definition on the CPU; here the gfx950 kernel of
csrc/fir_ops.hip runs through the C ABI (pf_upfirdn2d).  Inference only (the restoration path takes no gradient through it)."""
from __future__ import annotations

import torch

from ... import _lib


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """out = decimate_down(FIR_kernel(pad(zero_insert_up(input))))  on (N, C, H, W) fp32 GPU tensors; kernel: (kh, kw)."""
    return upfirdn2d_xy(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_xy(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    if not input.is_cuda:
        raise _lib.PnpFlowHipError("upfirdn2d needs a GPU tensor (there is no CPU path)")
    lib = _lib.load()
    x = input.contiguous().float()
    k = torch.as_tensor(kernel, dtype=torch.float32).to(x.device).contiguous()
    N, Cc, in_h, in_w = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = torch.empty((N, Cc, out_h, out_w), dtype=torch.float32, device=x.device)
    _lib.check(lib.pf_upfirdn2d(x.data_ptr(), k.data_ptr(), out.data_ptr(), N * Cc, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, _lib.current_stream_ptr()), None, "pf_upfirdn2d")
    return out
