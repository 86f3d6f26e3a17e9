"""ctypes binding of libpnpflow_hip.so (C ABI: include/pnpflow_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an
exception is raised.  The product path never routes through oracle/ or a CPU/eager
implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNPFLOW_HIP_LIB") or os.path.join(_HERE, "libpnpflow_hip.so")   # override: A/B builds of the kernels

PF_ABI_VERSION = 4

PF_DEG_DENOISING, PF_DEG_BOX_INPAINTING, PF_DEG_MASK_INPAINTING, PF_DEG_SUPERRESOLUTION, PF_DEG_GAUSSIAN_BLUR, PF_DEG_SR_FILTERED = range(6)


class PfUnetCfg(C.Structure):
    _fields_ = [("input_channels", C.c_int32), ("output_channels", C.c_int32), ("input_height", C.c_int32),
                ("ch", C.c_int32), ("num_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("num_res_blocks", C.c_int32), ("num_attn_resolutions", C.c_int32),
                ("attn_resolutions", C.c_int32 * 8)]


class PfNcsnppCfg(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("num_channels", C.c_int32), ("nf", C.c_int32), ("num_levels", C.c_int32),
                ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32), ("num_attn_resolutions", C.c_int32),
                ("attn_resolutions", C.c_int32 * 8), ("fir_taps", C.c_int32), ("fir_kernel", C.c_float * 8),
                ("skip_rescale", C.c_int32), ("scale_by_sigma", C.c_int32), ("centered", C.c_int32)]


class PfDegradation(C.Structure):
    _fields_ = [("kind", C.c_int32), ("half_size_mask", C.c_int32), ("sf", C.c_int32), ("ntaps", C.c_int32),
                ("mask", C.c_void_p), ("taps", C.c_void_p)]


class PfPnpParams(C.Structure):
    _fields_ = [("steps", C.c_int32), ("num_samples", C.c_int32), ("host_t", C.POINTER(C.c_float)),
                ("host_coef", C.POINTER(C.c_float)), ("seed", C.c_uint64), ("stream_base", C.c_uint64),
                ("noise", C.c_void_p), ("use_graph", C.c_int32), ("noise_model", C.c_int32), ("batch_samples", C.c_int32),
                ("reserved0", C.c_int32), ("elem_offset", C.c_uint64), ("host_cb_mask", C.c_void_p)]


class PfOtOdeParams(C.Structure):
    _fields_ = [("steps", C.c_int32), ("first", C.c_int32), ("host_t", C.POINTER(C.c_float)), ("host_one_minus_t", C.POINTER(C.c_float)),
                ("host_rt2", C.POINTER(C.c_float)), ("host_coef", C.POINTER(C.c_float)), ("sigma2", C.c_float), ("delta", C.c_float),
                ("use_graph", C.c_int32), ("reserved0", C.c_int32), ("host_cb_mask", C.c_void_p)]


ITER_CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

# name -> (restype, argtypes); this table is also what tests use to check that every symbol
# declared in include/pnpflow_hip.h is exported.
SIGNATURES = {
    "pf_abi_version": (C.c_int, []),
    "pf_engine_create": (C.c_int, [C.c_int, C.POINTER(PfUnetCfg), C.POINTER(C.c_void_p)]),
    "pf_ncsnpp_create": (C.c_int, [C.c_int, C.POINTER(PfNcsnppCfg), C.POINTER(C.c_void_p)]),
    "pf_engine_set_solver_time_scale": (C.c_int, [C.c_void_p, C.c_float]),
    "pf_engine_destroy": (None, [C.c_void_p]),
    "pf_last_error": (C.c_char_p, [C.c_void_p]),
    "pf_engine_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "pf_engine_finalize_weights": (C.c_int, [C.c_void_p]),
    "pf_engine_num_weights": (C.c_int, [C.c_void_p]),
    "pf_engine_weight_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "pf_engine_weight_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "pf_engine_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_unet_forward_retain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_unet_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_ot_ode_vec": (C.c_int, [C.POINTER(PfDegradation), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_ot_ode_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "pf_engine_num_taps": (C.c_int, [C.c_void_p]),
    "pf_engine_tap_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "pf_engine_read_tap": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_void_p]),
    "pf_degradation_H": (C.c_int, [C.POINTER(PfDegradation), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_degradation_H_adj": (C.c_int, [C.POINTER(PfDegradation), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_grad_step": (C.c_int, [C.POINTER(PfDegradation), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_grad_step_laplace": (C.c_int, [C.POINTER(PfDegradation), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_interpolate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pf_denoise_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "pf_fill_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pf_fill_normal_at": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pf_attention_core": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_psnr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pf_upfirdn2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 13 + [C.c_void_p]),
    "pf_fused_bias_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "pf_ssim": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_lpips_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pf_lpips_destroy": (None, [C.c_void_p]),
    "pf_lpips_last_error": (C.c_char_p, [C.c_void_p]),
    "pf_lpips_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "pf_lpips_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_ot_ode_restore": (C.c_int, [C.c_void_p, C.POINTER(PfDegradation), C.POINTER(PfOtOdeParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, ITER_CB, C.c_void_p]),
    "pf_pnp_flow_restore": (C.c_int, [C.c_void_p, C.POINTER(PfDegradation), C.POINTER(PfPnpParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, ITER_CB, C.c_void_p]),
    "pf_engine_memory_bytes": (C.c_int64, [C.c_void_p]),
    "pf_engine_check_numerics": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_engine_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_engine_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class PnpFlowHipError(RuntimeError):
    pass


def load():
    """Load libpnpflow_hip.so (built by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PnpFlowHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback implementation.")
    # torch first: the engine shares device pointers and streams with torch, so both must bind to the SAME HIP runtime
    # instance (torch ships its own libamdhip64; loading ours first pulls in /opt/rocm's copy and the process ends up
    # with two runtimes - the second one reports "no ROCm-capable device")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pf_abi_version() != PF_ABI_VERSION:
        raise PnpFlowHipError("libpnpflow_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, engine=None, what: str = ""):
    if rc != 0:
        msg = load().pf_last_error(engine)
        raise PnpFlowHipError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class solver_stream:
    """Context for the solver loops: the engine captures one hipGraph per iteration, which the legacy NULL stream cannot do, and
    the host callbacks of a loop (metrics: torch ops + pf_psnr / pf_ssim) must be ORDERED with the engine's own launches.  When
    torch's current stream is the NULL stream, this switches torch to a side stream for the duration of the call - the engine
    and every callback then enqueue on that one stream - and joins it with the caller's stream on both sides.
    (Round 2: with the engine on a private non-blocking stream and the metric callbacks on the NULL stream, a second batch replaying
    the cached graph produced NaNs after a few logged iterations; single-stream ordering removes the hazard class.)"""
    _side = {}

    def __enter__(self):
        import torch
        self.cur = torch.cuda.current_stream()
        self.ctx = None
        if self.cur.cuda_stream == 0:
            dev = torch.cuda.current_device()
            if dev not in solver_stream._side:
                solver_stream._side[dev] = torch.cuda.Stream(device=dev)
            self.side = solver_stream._side[dev]
            self.side.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.cur.wait_stream(self.side)
        return False
