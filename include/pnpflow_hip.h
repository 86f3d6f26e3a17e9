/* pnpflow_hip.h -- C ABI of libpnpflow_hip.so, the MI355X (gfx950) engine behind the
 * PnP-Flow restoration hot path.
 *
 * The reference (annegnx/PnP-Flow) has no FFI layer: its hot path is Python calling
 * PyTorch ops.  This ABI is the boundary the build creates *below* the reference's
 * `pnpflow.methods` / `pnpflow.degradations` / `pnpflow.models.UNet` Python API; each
 * entry point cites the reference code it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pf_status otherwise; no C++
 *     exception crosses the ABI; pf_last_error() gives the message of the last failure
 *     on that engine (or of pf_engine_create when called with NULL).
 *   - all tensor arguments are DEVICE pointers to contiguous fp32 NCHW buffers owned by
 *     the caller (what torch.Tensor.data_ptr() yields), unless named `host_*`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work
 *     is enqueued on it, nothing synchronises the device except where stated.
 *   - one engine per device; an engine is not thread-safe.
 */
#ifndef PNPFLOW_HIP_H
#define PNPFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 4

typedef enum pf_status {
    PF_OK = 0,
    PF_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    PF_ERR_HIP = -2,         /* a HIP runtime call failed */
    PF_ERR_WEIGHTS = -3,     /* unknown / missing / mis-shaped weight tensor */
    PF_ERR_STATE = -4,       /* call made in the wrong state (e.g. forward before finalize) */
    PF_ERR_NUMERIC = -5      /* a non-finite activation was detected inside the U-Net (see pf_engine_check_numerics) */
} pf_status;

typedef struct pf_engine pf_engine;

/* Hyper-parameters of pnpflow.models.UNet.__init__ (pnpflow/models.py:302-334);
 * define_model (pnpflow/utils.py:170-180) uses ch=32, ch_mult=(1,2,4,8),
 * num_res_blocks=6, attn_resolutions=(16,8). */
typedef struct pf_unet_cfg {
    int32_t input_channels;
    int32_t output_channels;
    int32_t input_height;      /* square images (the reference assumes it, utils.py:331) */
    int32_t ch;
    int32_t num_levels;
    int32_t ch_mult[8];
    int32_t num_res_blocks;
    int32_t num_attn_resolutions;
    int32_t attn_resolutions[8];
} pf_unet_cfg;


/* ---- NCSN++ ("rectified") velocity net ----------------------------------------------------
 * Hyper-parameters of pnpflow.image_generation.models.ncsnpp.NCSNpp.__init__ (ncsnpp.py:38-206) as the reference's two
 * rectified-flow configs set them (configs/rectified_flow/{celeba_hq,afhq_cat}_pytorch_rf_gaussian.py:46-64): BigGAN residual
 * blocks with FIR [1,3,3,1] resampling, input_skip / output_skip pyramids combined by `sum`, Gaussian Fourier time
 * conditioning, skip_rescale.  Only that block list is built (resblock_type 'biggan', progressive 'output_skip',
 * progressive_input 'input_skip', embedding 'fourier', conditional); nf must be a multiple of 32. */
typedef struct pf_ncsnpp_cfg {
    int32_t image_size;        /* config.data.image_size (256) */
    int32_t num_channels;      /* config.data.num_channels (3) */
    int32_t nf;                /* 128 */
    int32_t num_levels;        /* len(ch_mult) (7) */
    int32_t ch_mult[8];        /* (1,1,2,2,2,2,2) */
    int32_t num_res_blocks;    /* 2 */
    int32_t num_attn_resolutions;
    int32_t attn_resolutions[8];   /* (16,) */
    int32_t fir_taps;          /* len(fir_kernel) (4) */
    float fir_kernel[8];       /* [1,3,3,1] */
    int32_t skip_rescale;      /* 1 */
    int32_t scale_by_sigma;    /* 1: the output is divided by the time label (ncsnpp.py:378-381) */
    int32_t centered;          /* config.data.centered; 0: x -> 2x - 1 first (ncsnpp.py:248-250; not built: must be 1) */
} pf_ncsnpp_cfg;

/* replaces mutils.create_model(config) (models/utils.py:91-103).  The handle is a pf_engine: weights are loaded with
 * pf_engine_load_weight under NCSNpp's own state_dict keys ("all_modules.<i>.<...>", without DataParallel's "module."),
 * pf_unet_forward(e, x, labels, out, B, stream) is model(x, labels) with labels = the reference's time_cond (t * 999,
 * methods/pnp_flow.py:23-27); pf_unet_forward_retain / pf_unet_backward give its input-gradient VJP (ot_ode.py:137-138). */
int pf_ncsnpp_create(int device_id, const pf_ncsnpp_cfg* cfg, pf_engine** out);
/* label = t * scale inside pf_pnp_flow_restore / pf_ot_ode_restore (PNP_FLOW.model_forward: `t * 999`); 1 for the OT net */
int pf_engine_set_solver_time_scale(pf_engine* e, float scale);

int pf_abi_version(void);

/* ---- engine life cycle ------------------------------------------------------------ */
/* replaces UNet.__init__ + .to(device)  (pnpflow/models.py:302-436, methods/pnp_flow.py:15) */
int pf_engine_create(int device_id, const pf_unet_cfg* cfg, pf_engine** out);
void pf_engine_destroy(pf_engine* e);
const char* pf_last_error(const pf_engine* e);

/* replaces model.load_state_dict (pnpflow/utils.py:225): one call per state_dict entry,
 * `name` = the reference's key (e.g. "down_modules.3.3a_0b_attn.attn_q.weight"),
 * `host_data` = fp32 values in the reference's layout (OIHW / (out,in) / (C,)).
 * The engine repacks into its own device layout and owns the copy. */
int pf_engine_load_weight(pf_engine* e, const char* name, const float* host_data,
                          const int64_t* shape, int ndim);
/* checks that every tensor of the architecture was supplied; uploads. */
int pf_engine_finalize_weights(pf_engine* e);
/* number of state_dict entries the architecture expects, and the i-th name. */
int pf_engine_num_weights(const pf_engine* e);
const char* pf_engine_weight_name(const pf_engine* e, int i);
/* shape of the i-th entry (reference layout); returns the rank (<= 4) or a negative status. */
int pf_engine_weight_shape(const pf_engine* e, int i, int64_t shape[4]);

/* 1 (default) = split-fp16: every operand carried as an fp16 pair hi+lo, 3 x v_mfma_f32_32x32x16_f16 per product,
 *     fp32 accumulate - fp32-equivalent results (the parity tests hold both modes to the same tolerance) at 3/16 of
 *     the matrix-pipe time.  Applies to the packed-weight convs of the forward and of the backward (the VJP input is
 *     normalised by a power of two first) and selects the fused attention core (q k^T, softmax, P v in one launch)
 *     where its shapes apply; the unfused attention matmuls and their adjoints run on the fp32 MFMA.
 * 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere.
 * 2 = single fp16 MFMA per product: the packed-weight convs round both operands to fp16 (11-bit significands; the per-image
 *     power-of-two operand scales keep them in range) and accumulate in fp32 - the precision class of the TF32 convolutions the
 *     reference's CUDA runs use by PyTorch default (torch.backends.cudnn.allow_tf32), NOT fp32-equivalent: U-Net outputs
 *     agree with the fp32 reference to ~1e-3 relative, restored images to the +-0.05 dB PSNR the BASELINE tolerance asks. */
int pf_engine_set_precision(pf_engine* e, int mode);

/* ---- velocity field ---------------------------------------------------------------- */
/* replaces UNet.forward (pnpflow/models.py:442-495) as called by PNP_FLOW.model_forward
 * (pnpflow/methods/pnp_flow.py:19-21):  v[B,Cout,H,W] = v_theta(x[B,Cin,H,W], t[B]). */
int pf_unet_forward(pf_engine* e, const float* x, const float* t, float* v, int B, void* stream);

/* debugging / parity localisation: copy an internal NHWC activation (by plan tensor
 * index) of the last forward to a HOST fp32 buffer in NCHW order; returns C*H*W via
 * dims[3] = {C,H,W}.  Synchronises the stream. */
int pf_engine_num_taps(const pf_engine* e);
const char* pf_engine_tap_name(const pf_engine* e, int i);
int pf_engine_read_tap(pf_engine* e, int i, float* host_out, int64_t capacity, int32_t dims[3], void* stream);

/* ---- degradation operators (pnpflow/degradations.py) -------------------------------- */
typedef enum pf_degradation_kind {
    PF_DEG_DENOISING = 0,       /* Denoising           degradations.py:15-20  */
    PF_DEG_BOX_INPAINTING = 1,  /* BoxInpainting       degradations.py:23-32, utils.py:327-336 */
    PF_DEG_MASK_INPAINTING = 2, /* RandomInpainting    degradations.py:35-44 (mask supplied: utils.py:353-361) */
    PF_DEG_SUPERRESOLUTION = 3, /* Superresolution     degradations.py:92-127, mode=None, utils.py:283-310 */
    PF_DEG_GAUSSIAN_BLUR = 4,   /* GaussianDeblurring  degradations.py:55-89 (circular, separable) */
    PF_DEG_SR_FILTERED = 5      /* Superresolution mode="bicubic": circular separable filter (utils.py:365-396), then decimation
                                   degradations.py:97-127; uses sf, ntaps (4*sf, even), taps */
} pf_degradation_kind;

typedef struct pf_degradation {
    int32_t kind;
    int32_t half_size_mask;     /* BOX */
    int32_t sf;                 /* SUPERRESOLUTION */
    int32_t ntaps;              /* GAUSSIAN_BLUR / SR_FILTERED: <= 127; tap ntaps/2 sits at offset 0 (the reference's roll by -(K-1)//2) */
    const uint8_t* mask;        /* MASK: device [B][H][W] bytes, 1 = keep */
    const float* taps;          /* GAUSSIAN_BLUR: device [ntaps] separable 1-D taps */
} pf_degradation;

/* y = H(x)      x:[B,C,H,W] -> y:[B,C,Hy,Wy]  (Hy = H/sf for SR, else H) */
int pf_degradation_H(const pf_degradation* d, const float* x, float* y, int B, int C, int H, int W,
                     float* scratch, void* stream);
/* x = H_adj(y) */
int pf_degradation_H_adj(const pf_degradation* d, const float* y, float* x, int B, int C, int H, int W,
                         float* scratch, void* stream);

/* ---- PnP-Flow iteration pieces (pnpflow/methods/pnp_flow.py) ------------------------- */
/* z = x - coef[b] * H_adj(H(x) - y),  coef[b] = lr_t[b] / sigma^2
 * replaces grad_datafit + the update at pnp_flow.py:39-41, 109-112 (gaussian noise).
 * scratch: >= 2*B*C*H*W floats for GAUSSIAN_BLUR and SR_FILTERED (pf_degradation_H/H_adj: B*C*H*W resp. 2*B*C*H*W), may be NULL otherwise. */
int pf_grad_step(const pf_degradation* d, const float* x, const float* y, const float* coef, float* z,
                 int B, int C, int H, int W, float* scratch, void* stream);
/* Laplace noise model (pnp_flow.py:42-43): z = x - coef[b] * H_adj(2*heaviside(H(x) - y, 0) - 1), coef[b] = lr_t[b]/sigma */
int pf_grad_step_laplace(const pf_degradation* d, const float* x, const float* y, const float* coef, float* z,
                         int B, int C, int H, int W, float* scratch, void* stream);
/* z_tilde = t[b]*z + (1-t[b])*eps    (interpolation_step, pnp_flow.py:47-48)
 * eps = `noise` if non-NULL, else Philox4x32-10/Box-Muller (seed, stream_id). */
int pf_interpolate(const float* z, const float* t, const float* noise, uint64_t seed, uint64_t stream_id,
                   float* z_tilde, int B, int n_per_image, void* stream);
/* acc (=|+=) z_tilde + (1-t[b])*v ; final: acc = (acc + ...)*inv_count  (denoiser + average,
 * pnp_flow.py:50-52, 114-121).  mode: bit0 = first sample (overwrite), bit1 = last (scale). */
int pf_denoise_accumulate(float* acc, const float* z_tilde, const float* v, const float* t, int mode,
                          float num_samples, int B, int n_per_image, void* stream);
/* fills out[n] with engine normals (the same generator pf_interpolate uses). */
int pf_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* the same, starting at normal number `elem_offset` of the stream: out[i] = normal number elem_offset + i.  A shard of a
 * multi-GPU run that owns images [lo, hi) of the global batch draws elem_offset = lo*C*H*W, i.e. exactly the numbers a
 * single-device run at the global batch size draws for those images (replaces the one torch.randn_like(x) over the whole
 * batch of interpolation_step, pnpflow/methods/pnp_flow.py:47-48, when the batch is split over GPUs). */
int pf_fill_normal_at(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, void* stream);

/* per-image PSNR of postprocess(rec) vs postprocess(clean), data_range 1
 * (pnpflow/utils.py:560-577, 594-611) -> out[B] (device). */
int pf_psnr(const float* rec, const float* clean, float* out, int B, int n_per_image, void* stream);

/* per-image SSIM of postprocess(rec) vs postprocess(clean): ignite.metrics.SSIM(data_range=1.0) as the reference calls it
 * (pnpflow/utils.py:780-802) - 11x11 Gaussian window sigma 1.5, k1 0.01, k2 0.03, reflect padding, mean over (C,H,W) in
 * fp64 -> out[B] (device, double).  H, W > 5.  PARITY UNPINNED: ignite is absent from the build container. */
int pf_ssim(const float* rec, const float* clean, double* out, int B, int C, int H, int W, void* stream);

/* ---- LPIPS (AlexNet, v0.1), SURVEY 8f N2 ------------------------------------------------------------------------------------
 * What the reference logs next to PSNR / SSIM (pnpflow/utils.py:677-724): lpips.LPIPS(net='alex')(img0, img1, normalize=True).
 * The network (torchvision AlexNet features + lpips' five 1x1 "lin" heads) is restated in csrc/lpips.hip; weights are loaded under
 * the published names "features.{0,3,6,8,10}.{weight,bias}" (torchvision.models.alexnet) and "lin{0..4}" (the [1][C][1][1] conv of
 * lpips' NetLinLayer, as [C]).  PARITY UNPINNED: neither package nor their weight files are in the build image. */
typedef struct pf_lpips pf_lpips;
int pf_lpips_create(int device_id, pf_lpips** out);
void pf_lpips_destroy(pf_lpips* l);
const char* pf_lpips_last_error(const pf_lpips* l);
int pf_lpips_load_weight(pf_lpips* l, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* img0, img1: [B][3][H][W] fp32 on the device; out[B] (device) <- LPIPS distance per image pair.  normalize != 0 applies the
 * package's 2x - 1 first (the reference passes normalize=True on images that already are in [-1, 1]: utils.py:703-708). */
int pf_lpips_forward(pf_lpips* l, const float* img0, const float* img1, float* out, int B, int H, int W, int normalize, void* stream);

/* ---- the reference's native ops (NCSN++ "rectified" velocity net) ------------------------------------------- */
/* upfirdn2d (pnpflow/image_generation/op/upfirdn2d_kernel.cu:49-369; definition: op/upfirdn2d.py:142-187): per plane of
 * in[planes][in_h][in_w]: zero-insert upsample by (up_x, up_y), pad (negative = crop) by (pad_x0, pad_x1, pad_y0, pad_y1), true 2-D
 * convolution with kernel[kh][kw] (device), decimate by (down_x, down_y) -> out[planes][out_h][out_w],
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh) / down_y + 1 (same for w). */
int pf_upfirdn2d(const float* in, const float* kernel, float* out, int planes, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                 int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* fused_bias_act (op/fused_bias_act_kernel.cu:19-99): out[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale;
 * act 1 = linear, 3 = leaky ReLU(alpha); grad 0 = value, 1 = first derivative (sign taken from ref[i], the saved output),
 * 2 = zero; bias / ref may be NULL. */
int pf_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t n, int step_b, int size_b, int act, int grad,
                      float alpha, float scale, void* stream);

/* ---- attention core -------------------------------------------------------------------- */
/* out[B,T,C] = softmax(q k^T * C^-1/2, dim=-1) v for q|k|v stacked as qkv[B,T,3C] (fp32, token-major): the bmm /
 * softmax / bmm of SelfAttention.forward (pnpflow/models.py:152-158) in one launch, split-f16 MFMA (fp32-equivalent).
 * Shapes: T in {128, 256, 512, 768, ... 4096}, C in {128, 256}; anything else returns PF_ERR_INVALID (the engine then
 * uses three launches). */
int pf_attention_core(const float* qkv, float* out, int B, int T, int C, void* stream);

/* ---- vector-Jacobian product (OT-ODE) ------------------------------------------------ */
/* replaces torch.autograd.functional.vjp(lambda z: model(z, t), x, vec) at
 * pnpflow/methods/ot_ode.py:137-138.  pf_unet_forward_retain = a forward that keeps every
 * activation; pf_unet_backward = J^T vec at the last retained forward (input gradient only);
 * pf_unet_vjp = both (v: the forward output, g: J^T vec). */
int pf_unet_forward_retain(pf_engine* e, const float* x, const float* t, float* v, int B, void* stream);
int pf_unet_backward(pf_engine* e, const float* vec, float* g, int B, void* stream);
int pf_unet_vjp(pf_engine* e, const float* x, const float* t, const float* vec, float* v, float* g, int B, void* stream);

/* OT-ODE linear solve + adjoint (pnpflow/methods/ot_ode.py:72-130) and Euler update (:141-147).
 *   vec = H_adj( (rt2[b] H H^T + sigma2)^-1 (y - H(x + one_minus_t[b]*vt)) )
 *   x  += delta * (vt + coef[b] * (vec + one_minus_t[b]*g)),  coef = ((1-t)/t)*gamma
 * Closed form per pixel where H H^T is diagonal (denoising, masks, decimation: ot_ode.py:81-106); for
 * GAUSSIAN_BLUR the Fourier-domain solve of ot_ode.py:108-117 with a hand-written 2-D FFT, which needs
 * scratch >= 4*B*C*H*W + H + W floats (H, W <= 2048); scratch may be NULL for the other operators. */
int pf_ot_ode_vec(const pf_degradation* d, const float* x, const float* vt, const float* y, const float* one_minus_t,
                  const float* rt2, float sigma2, float* vec, int B, int C, int H, int W, float* scratch, void* stream);
int pf_ot_ode_update(float* x, const float* vt, const float* vec, const float* g, const float* one_minus_t,
                     const float* coef, float delta, int B, int n_per_image, void* stream);

/* Whole OT-ODE loop of one batch on the device (pnpflow/methods/ot_ode.py:63-147): iterations first..steps-1 of
 *   v_t = v_theta(x, t);  vec = H_adj((r_t^2 H H^T + sigma^2)^-1 (y - H(x + (1-t) v_t)));  g = J^T vec;
 *   x += delta * (v_t + coef * (vec + (1-t) g))
 * with the per-iteration scalars (host tables, fp32 values computed with the reference's own expressions - including its
 * `delta * iteration**2` quirk for superresolution, ot_ode.py:96) read on the device through an iteration counter, the buffers
 * pre-allocated, and one hipGraph per Euler step (retained forward -> solve -> hand-written backward -> update) captured once
 * and replayed.  x_inout: the initialisation t0*H_adj(y) + (1-t0)*noise on entry (ot_ode.py:27-28, 50-52), the result on exit. */
typedef struct pf_ot_ode_params {
    int32_t steps;                  /* steps_ode */
    int32_t first;                  /* int(steps_ode * start_time) */
    const float* host_t;            /* host [steps] */
    const float* host_one_minus_t;  /* host [steps] */
    const float* host_rt2;          /* host [steps] */
    const float* host_coef;         /* host [steps]: ((1-t)/t) * gamma_t */
    float sigma2;                   /* sigma_noise^2 */
    float delta;                    /* 1 / steps_ode */
    int32_t use_graph;
    int32_t reserved0;
    const uint8_t* host_cb_mask;    /* optional host [steps]: iterations after which iter_cb is called; NULL = every iteration */
} pf_ot_ode_params;
int pf_ot_ode_restore(pf_engine* e, const pf_degradation* d, const pf_ot_ode_params* prm, const float* y, float* x_inout, int B,
                      void* stream, void (*iter_cb)(int iteration, void* user), void* user);

/* ---- whole restoration loop --------------------------------------------------------- */
typedef struct pf_pnp_params {
    int32_t steps;            /* steps_pnp */
    int32_t num_samples;
    const float* host_t;      /* host [steps]  : t value of each iteration (fp32, as the reference computes it) */
    const float* host_coef;   /* host [steps]  : lr_t / sigma^2 of each iteration */
    uint64_t seed;            /* Philox key for the interpolation noise */
    uint64_t stream_base;     /* noise stream id of (iteration it, sample s) = stream_base + it*num_samples + s */
    const float* noise;       /* optional device [steps*num_samples][B*C*H*W] injected noise (parity runs) */
    int32_t use_graph;        /* capture one outer iteration in a hipGraph and replay it */
    int32_t noise_model;      /* 0 gaussian (pf_grad_step), 1 laplace (pf_grad_step_laplace) */
    int32_t batch_samples;    /* evaluate the num_samples velocities of an iteration as one U-Net pass over num_samples*B images */
    int32_t reserved0;
    uint64_t elem_offset;     /* first normal number of this shard inside every (iteration, sample) noise stream: lo*C*H*W for the
                                 shard that owns images [lo, hi) of a global batch (0 for a single-device run); see pf_fill_normal_at */
    const uint8_t* host_cb_mask; /* optional host [steps]: iter_cb is called (stream synchronised) only after iterations with a non-zero
                                 entry - the reference's logging iterations, pnp_flow.py:128-139; NULL = after every iteration */
} pf_pnp_params;

/* Runs pnp_flow.py:93 and 102-121 for one batch:  x0 = H_adj(1);  `steps` iterations.
 * y: measurement [B,C,Hy,Wy]; x_out: [B,C,H,W].  iter_cb, if non-NULL, is called on the
 * host after each iteration (stream synchronised) - used for the periodic metrics. */
typedef void (*pf_iter_callback)(int iteration, void* user);
int pf_pnp_flow_restore(pf_engine* e, const pf_degradation* d, const pf_pnp_params* prm,
                        const float* y, float* x_out, int B, void* stream,
                        pf_iter_callback iter_cb, void* user);

/* Numeric health of the forwards run so far: every GroupNorm finalisation checks the activation statistics it consumes; an
 * overflow / NaN anywhere upstream makes them non-finite and sets a device flag.  This call synchronises `stream`, returns
 * PF_ERR_NUMERIC if the flag was set (and clears it), PF_OK otherwise.  pf_pnp_flow_restore checks it before returning.
 * (The reference computes in plain fp32, pnpflow/models.py:94-113; the split-fp16 operands of the default precision mode are
 * range-guarded by a per-image power-of-two scale, this is the loud backstop.) */
int pf_engine_check_numerics(pf_engine* e, void* stream);

/* device bytes currently held by the engine (weights, activation plans, solver buffers): what torch.cuda.max_memory_allocated
 * cannot see of the reference's `compute_memory` bookkeeping (pnpflow/methods/pnp_flow.py:99-100, 141-146). */
int64_t pf_engine_memory_bytes(const pf_engine* e);

/* kernel-time accounting for bench.py: enables HIP-event timing of the U-Net conv-GEMM
 * launches on `stream`; read back accumulated (count, milliseconds). */
int pf_engine_profile(pf_engine* e, int enable);
int pf_engine_profile_read(pf_engine* e, int64_t* launches, double* ms_conv_gemm, double* flops_conv_gemm);

#ifdef __cplusplus
}
#endif
#endif /* PNPFLOW_HIP_H */
