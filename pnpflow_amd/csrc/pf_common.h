// Internal declarations shared by the engine (host) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace pf {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process may hold engines on several devices
// (pf_engine_create's device_index), so "set once" is tracked per device, not per process.  `done` is the call site's
// own static mask; devices past 63 simply set the attribute on every launch.
inline hipError_t set_max_dynamic_lds_once(const void* kernel, unsigned long long& done, int bytes = 160 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done |= 1ull << dev;
    return e;
}

// Compute units of the current device, queried once per device (the persistent kernels size their grids and their launch thresholds
// by it; 0 when the query fails - the persistent kernels are then refused)
inline int device_cu_count() {
    static std::atomic<int> cache[64] = {};      // (relaxed: every writer stores the same value for a device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 0) v = 0;
    if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}
// workgroups of a persistent launch: one per CU, a multiple of the 8 XCDs (the kernels map blockIdx & 7 to the XCD)
inline int persistent_grid() { return (device_cu_count() / 8) * 8; }

// One K-segment of the implicit GEMM: a source activation tensor (NHWC) plus the
// slice of the weights that multiplies it.
struct ConvSeg {
    const float* src;     // NHWC [B][Hs][Ws][cstride], channels [coff, coff+C) are used
    int C;                // channels contributed by this segment (GEMM K = taps*C)
    int cstride;          // floats per pixel in src
    int coff;             // first channel inside the pixel vector
    int xform;            // 0 raw, 1 GroupNorm, 2 GroupNorm+SiLU  (applied while staging)
    int taps;             // 9 (3x3, pad 1) or 1 (1x1 / plain GEMM)
    int gn_off;           // index of this segment's channel 0 in the GroupNorm channel space
    const double* stats;  // [B][C][2] per-channel (sum, sumsq) of src, or nullptr
    const float* w;       // weights
    int w_mode;           // 0: fragment-major repack [chunk][tap][kstep(2)][Cout][8] (see packed_conv)
                          // 1: generic strided operand: element (n,k) at b*w_bs + n*w_ns + k*w_ks
    int64_t w_bs, w_cs, w_ts, w_ns, w_ks;
    const void* w16;      // optional split-fp16 repack [chunk][tap][Cout][16 hi | 16 lo] (x256), see conv_mfma16.hip
    const void* w16h;     // optional hi-only repack [slice16][tap][Cout][16 hi] (x256) of the single-term mode's LDS-DMA kernel (conv_dma.hip)
    const void* a16;      // conv_dma.hip only: the pre-transformed fp16 operand of this segment (prep_split_kernel), or nullptr
    int oy, ox;           // conv_mfma16.hip, taps == 4 only: the segment's 2 x 2 window covers patch rows oy + {0, 1}, columns ox + {0, 1} of the 3 x 3 neighbourhood
};

struct ConvParams {
    ConvSeg seg[3];
    int nseg;
    int B, H, W;          // output spatial size
    int Hs, Ws;           // source spatial size (H*2 for stride 2; H/2 for fused nearest-up)
    int Cout;
    float* out;           // NHWC [B][H][W][out_cstride]
    int out_cstride;
    const float* addvec;  // bias (+ time-embedding projection): [b*addvec_bs + n], or nullptr
    int addvec_bs;
    const float* residual;  // NHWC, added in the epilogue (times res_scale), or nullptr
    int res_cstride;
    float res_scale;      // 1 for the DDPM blocks; 1/sqrt(2) for the NCSN++ blocks' (x + h)/sqrt(2) (layerspp.py:272-274)
    double* stats_out;    // [B][Cout][2] += per-channel (sum, sumsq) of the result, or nullptr
    float out_scale;      // applied to the accumulator before bias/residual
    // GroupNorm of the (concatenated) transformed segments
    int gn_C;             // total normalised channels (0 = no GN)
    int gn_cpg;           // channels per group
    float gn_eps;
    const float* gamma;   // [gn_C]
    const float* beta;
    // Per-image GroupNorm coefficients of this launch, finalised ONCE per launch by gn_coef_kernel (launch_gn_coef) instead of
    // by every workgroup: coef[(b*2 + 0)*coef_stride + c] = gamma[c]*rstd, coef[(b*2 + 1)*coef_stride + c] = beta[c] - mean*gamma[c]*rstd
    // over the gn_C normalised channels; nullptr only when gn_C == 0.
    const float* coef;
    int coef_stride;
    // Power-of-two operand scales of image b (split-fp16 kernels only; nullptr = 1): the staged activations of K-segment si are
    // multiplied by scale[8b + si] before they are split into fp16 hi/lo; the accumulator is kept in the units of the segment
    // being walked (multiplied by scale[8b + si'] * scale[8b + 4 + si] at a switch si -> si', exact) and by scale[8b + 4 + last]
    // in the epilogue - so raw (un-normalised) inputs of any magnitude keep fp32-grade relative accuracy and cannot overflow fp16.
    const float* scale;
    // GroupNorm(+SiLU) backward, first stage, fused into the epilogue (adjoint convs of the VJP on the split-fp16 kernel): the conv
    // result da (gradient w.r.t. the normalised/activated tensor) becomes dyhat = da * act'(u) * gamma in registers, and the
    // per-channel sums (sum dyhat, sum dyhat*yhat) are reduced in the same kernel - what unet_bwd.hip's gn_bwd_pre pass does with a
    // read and a write of the whole tensor.  gnb_x == nullptr: off.  Output channel n <-> GroupNorm channel gnb_coff + n.
    const float* gnb_x;          // forward input of the GroupNorm: NHWC, channels [0, Cout) of a tensor with gnb_xstride floats per pixel
    int gnb_xstride;
    const float* gnb_mu; const float* gnb_rs;         // [B][gnb_Ct]
    const float* gnb_gamma; const float* gnb_beta;    // [gnb_Ct]
    int gnb_Ct, gnb_coff, gnb_silu;
    double* gnb_sum;             // [B][gnb_Ct][2] += (sum dyhat, sum dyhat*yhat)
    int xcd_map;          // split-fp16 kernel, set by its launcher: 1 = workgroup -> (tile, N-block) mapping that keeps neighbouring tiles and the N-blocks of a tile on one XCD
    // round 6 (split-fp16 kernels): output channels >= pack_from_p1 - 1 (a multiple of 4; 0 = none) leave as PACKED fp16 pairs - per element the 32-bit word
    // hi | lo << 16 with hi = RNE16(v), lo = RNE16(v - hi), the split the fused attention core applies to its keys and values - in place of the fp32 value: the
    // stacked q,k,v conv packs its k and v channels, so the core's 32 query tiles per image unpack (two v_perm per pair) instead of each converting every key and value
    int pack_from_p1;
    // conv_mfma16.hip: source row pitch in pixels (of seg.cstride floats) when it is not Ws - a segment may then be a strided view of a larger tensor (every
    // second row and column of the fine-resolution gradient: the phase form of the upsampling conv's adjoint, engine.hip); 0 = Ws
    int src_row_pitch;
    // ... and the same for the result (and the residual it accumulates onto): row pitch in pixels of out_cstride floats; 0 = W.  With out = base + (py W + px) C,
    // out_cstride = 2 C, dst_row_pitch = 2 W' the launch writes output phase (py, px) of a tensor twice as large (the stride-2 conv's adjoint, engine.hip)
    int dst_row_pitch;
};

// fp32 -> packed (hi | lo << 16) fp16 pair, the exact split of attention.hip's split4 (the value is made opaque first: see the note there)
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned int pack_hilo(float v) {
    asm volatile("" : "+v"(v));
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    return (unsigned int)__builtin_bit_cast(unsigned short, h) | ((unsigned int)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ __forceinline__ float4 pack_hilo4(float4 v) {
    return make_float4(__uint_as_float(pack_hilo(v.x)), __uint_as_float(pack_hilo(v.y)), __uint_as_float(pack_hilo(v.z)), __uint_as_float(pack_hilo(v.w)));
}
#endif

constexpr int CONV_KC = 16;   // channels per K-chunk staged in LDS

// direct (VALU) convolutions at the image boundary of the network
struct EdgeConvParams {
    const float* in;      // begin: NCHW image [B][Cimg][H][W];   end: NHWC [B][H][W][C]
    float* out;           // begin: NHWC [B][H][W][C];            end: NCHW image
    const float* w;       // begin: [tap][cimg][C]  end: [tap][cimg][C]  (C innermost)
    const float* bias;
    int B, H, W, Cimg, C;
    // end conv only: GroupNorm+SiLU of the input
    const double* stats; const float* gamma; const float* beta; int gn_cpg; float gn_eps;
    double* stats_out;    // begin conv: per-channel stats of the output
    const void* w16;      // split-fp16 MFMA weight image [k-step 2][hi | lo][lane 64][8 halfs], x 2^8 (engine.hip): end conv in its GroupNorm + SiLU form (k = channel), begin conv forward (k = input channel * 9 + tap); nullptr: the VALU kernels
};

// fused attention core (attention.hip): qkv [B][T][3C] (q | k | v) -> out [B][T][C]
// round 6 (engine.hip attn_block): with proj_out folded into the value projection (out = x + P (v Wp^T) + bp: exact algebra) the core's epilogue
// adds `bias[c]` and `residual[b][t][c]`; both null = the plain core
// `stats_part` [B][T / 32][C][2] (or null): fp32 (sum, sum of squares) of every 32-query tile of the result, for launch_partial_stats
// `kv_packed`: the k and v channels of qkv hold packed fp16 pairs (ConvParams::pack_from_p1) instead of fp32 values
struct AttnParams { const float* qkv; float* out; int B, T, C; float scale; const float* bias; const float* residual; float* stats_part; int kv_packed; };
// GroupNorm statistics stats[b][C][2] of a tensor from per-tile partial sums part[B][nparts][C][2] (fp32), added in tile order in fp64 (plain stores): the
// folded attention core's result (fp64 atomics from its 32-query tiles - 655 k per launch - cost 30 us of a 110 us kernel)
hipError_t launch_partial_stats(const float* part, double* stats, int B, int nparts, int C, hipStream_t s);
bool attn_fused_supported(int T, int C);
hipError_t launch_attn_fused(const AttnParams& p, hipStream_t s);

struct TembParams {
    const float* t;         // [B]
    const float* w0; const float* b0;   // Linear(ch -> 4ch)     (out,in)
    const float* w1; const float* b1;   // Linear(4ch -> 4ch)
    const float* wp;        // all temb_proj weights stacked: [total_out][4ch]
    const float* bp;        // stacked (temb_proj.bias + conv1.bias): [total_out]
    float* out;             // [B][total_out]
    int B, ch, total_out;
};

// GroupNorm coefficient / operand-scale finalisation of one conv launch (unet_misc.hip): one block per image
struct GnCoefParams {
    const double* st[3];  // per K-segment of the consuming launch: per-channel (sum, sumsq) [B][C][2] of its source, or nullptr
    int C[3], xform[3], gn_off[3];
    int nseg, gn_C, gn_cpg, HW;      // HW = pixels of the SOURCE tensors (what the statistics were summed over)
    float eps;
    const float* gamma; const float* beta;
    float* coef; int coef_stride;    // see ConvParams::coef (written only when gn_C > 0)
    float* scale;                    // [B][8] (s of segment 0..2, pad, 1/s of segment 0..2, pad), or nullptr
    unsigned int* flags;             // [0] bit 0 is set when a statistic is not finite (an activation overflowed / NaN upstream); [1] = id + 1 of the first launch that saw it
    int id;                          // index of the consuming conv in the plan (diagnostics)
    // retained forwards (VJP): per-channel mean and 1 / sqrt(var + eps) of the GroupNorm, [B][gn_C] each - what the backward's GroupNorm stages need (they were
    // recomputed from the same statistics by a launch of their own per GroupNorm: 9 900 launches of 5 us per 95 Euler steps of C5); null otherwise
    float* mu_out; float* rs_out;
};
hipError_t launch_gn_coef(const GnCoefParams& p, int B, hipStream_t s);

// conv_dma.hip: prep pass (GroupNorm-apply + SiLU + operand scale + fp16 hi/lo split into the padded record layout) and the conv whose
// A operand arrives by LDS-DMA from it
struct PrepParams {
    const float* src[3]; void* dst[3];
    int C[3], cstride[3], coff[3], xform[3], gn_off[3];
    int nseg, B, Hs, Ws, terms;
    const float* coef; int coef_stride;      // ConvParams::coef of the consuming launch (gn_coef_kernel ran before)
    const float* scale;                      // ConvParams::scale, or nullptr
};
hipError_t launch_prep_split(const PrepParams& p, hipStream_t s);
bool conv_dma_supported(const ConvParams& p, int stride, int up, int terms);
bool conv_dma_phase_shape_ok(const ConvParams& p, int terms);      // the nearest-x2 upsampling conv described by p (its 9-tap form) can run in conv_dma's phase form (up = 2)
hipError_t launch_conv_dma(const ConvParams& p, int up, hipStream_t s, int terms);
size_t conv_dma_a16_bytes(int B, int C, int Hs, int Ws, int terms);

// conv_pp.hip: persistent two-team kernel of the full-resolution 32-channel level.  A launch is a list of 32-channel K-chunks (9-tap chunks
// with GroupNorm(+SiLU) staging first, then raw 1-tap chunks of a folded 1x1 shortcut), each with its own LDS weight image
// [k16-step = tap * 2 + j][hi | lo][k-half][Cout = 32][8 halfs] (values x 2^8, split like packed_conv16).
constexpr int PP_MAXCH = 24;     // conv_pp: <= 4 chunks of 32 channels; conv_sp: <= 24 chunks of 16 channels (cat[256, 128] -> 128)
struct PPChunk {
    const float* src;     // NHWC source tensor of the chunk's K-segment
    const void* wimg;     // weight image of this chunk in global memory: taps * 4 KiB (conv_sp: 36 / 72 KiB per 16-channel chunk of 64 / 128 output channels)
    int cstride, coff;    // floats per pixel of src; first channel of the chunk inside the pixel vector
    int xform;            // 1 GroupNorm, 2 GroupNorm + SiLU (9-tap chunks); 0 raw (1-tap chunks)
    int gn_c0;            // index of the chunk's channel 0 in the launch's GroupNorm coefficient vectors
    int seg;              // K-segment index (operand scale slot)
    int pad_;
};
struct PPParams {
    PPChunk ch[PP_MAXCH]; // n9 nine-tap chunks, then n1 one-tap chunks (conv_pp is instantiated per (n9, n1, residual); conv_sp walks n9 at run time)
    int n9, n1;
    int cout;                             // 32: conv_pp.hip; 64 / 128: conv_sp.hip
    int B, H, W, lx, ly;                  // tiles per row / column = 1 << lx / 1 << ly (8 x 16-pixel tiles)
    int rot;                              // workgroup g starts (g * rot) % (tiles of its range) tiles into its range (set by the launcher)
    float* out; const float* addvec; int addvec_bs; const float* residual; float res_scale; double* stats_out; float out_scale;
    const float* coef; int coef_stride; const float* scale;      // ConvParams::coef / ::scale of the launch
};
bool conv_pp_supported(const ConvParams& p, int stride, int up, int terms);
hipError_t launch_conv_pp(const PPParams& p, hipStream_t s, int terms = 3);      // terms 1: precision mode 2 (hi-only operands)
bool conv_sp_supported(const ConvParams& p, int stride, int up, int terms);      // conv_sp.hip: one wave per SIMD, software-pipelined (Cout = 64 / 128)
hipError_t launch_conv_sp(const PPParams& p, hipStream_t s, int terms = 3);      // terms 1: precision mode 2 (hi-only operands and weight images)

hipError_t launch_conv(const ConvParams& p, int stride, int up, hipStream_t s);
hipError_t launch_conv16(const ConvParams& p, int stride, int up, hipStream_t s, int terms = 3);   // split-fp16 MFMA variant (terms 3) / single fp16 MFMA (terms 1)
size_t conv_flops(const ConvParams& p);
hipError_t launch_begin_conv(const EdgeConvParams& p, hipStream_t s);
hipError_t launch_end_conv(const EdgeConvParams& p, hipStream_t s);
hipError_t launch_temb(const TembParams& p, hipStream_t s);
hipError_t launch_softmax_rows(float* data, int64_t rows, int cols, hipStream_t s);

// backward helpers (unet_bwd.hip)
hipError_t launch_gn_fwd_coeffs(const double* st0, int C0, const double* st1, int C1, int cpg, int HW, float eps, float* mu, float* rs, int B,
                                hipStream_t s);
hipError_t launch_gn_bwd_pre(float* g, const float* x, const float* mu, const float* rs, const float* gamma, const float* beta, double* bsum,
                             int B, int HW, int C, int coff, int Ct, int silu, hipStream_t s);
hipError_t launch_gn_bwd_coeffs(const double* bsum, int Ct, int cpg, int HW, float* m1, float* m2, int B, hipStream_t s);
hipError_t launch_gn_bwd_post(const float* dy, const float* x, const float* mu, const float* rs, const float* m1, const float* m2,
                              const float* add, float* out, int B, int HW, int C, int coff, int Ct, int accumulate, hipStream_t s, float add_scale = 1.0f);
hipError_t launch_transpose(const float* in, float* out, int B, int R, int Cc, hipStream_t s);
hipError_t launch_softmax_bwd(const float* A, float* dA, int64_t rows, int cols, float scale, hipStream_t s);
hipError_t launch_sumpool2(const float* in, float* out, int B, int H, int W, int C, int accumulate, hipStream_t s);

// pointwise / operator kernels (NCHW fp32 images)
enum { DEG_DENOISE = 0, DEG_BOX = 1, DEG_MASK = 2, DEG_SR = 3, DEG_BLUR = 4, DEG_SR_FILTER = 5 };
struct DegView {       // device-side view of pf_degradation
    int kind, half, sf, ntaps;
    const uint8_t* mask;
    const float* taps;
};
hipError_t launch_deg_H(const DegView& d, const float* x, float* y, int B, int C, int H, int W, float* scratch, hipStream_t s);
hipError_t launch_deg_Hadj(const DegView& d, const float* y, float* x, int B, int C, int H, int W, float* scratch, hipStream_t s);
hipError_t launch_grad_step(const DegView& d, const float* x, const float* y, const float* coef, float* z,
                            int B, int C, int H, int W, float* scratch, int laplace, hipStream_t s);
hipError_t launch_interpolate(const float* z, const float* t, const float* noise, uint64_t seed, uint64_t stream_id,
                              float* zt, int B, int n, hipStream_t s);
hipError_t launch_interp_iter(const float* z, const float* t, const float* noise, const unsigned long long* rng /* device [seed, stream_base, elem_offset] */,
                              const int* iter, int num_samples, int sample, float* zt, int B, int n, hipStream_t s);
hipError_t launch_denoise_accum(float* acc, const float* zt, const float* v, const float* t, int mode, float ns,
                                int B, int n, hipStream_t s);
hipError_t launch_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, hipStream_t s);
hipError_t launch_psnr(const float* rec, const float* clean, float* out, int B, int n, hipStream_t s);
hipError_t launch_fill(float* out, int64_t n, float v, hipStream_t s);
// FIR resampling / fused bias+activation of the NCSN++ velocity net (fir_ops.hip)
hipError_t launch_upfirdn2d(const float* in, const float* kernel, float* out, int planes, int in_h, int in_w, int kh, int kw,
                            int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, hipStream_t s);
hipError_t launch_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t n, int step_b, int size_b,
                                 int act, int grad, float alpha, float scale, hipStream_t s);
hipError_t launch_ssim(const float* rec, const float* clean, double* out, int B, int C, int H, int W, hipStream_t s);   // metrics.hip
hipError_t launch_vjp_normalise(const float* vec, float* vec_scaled, int64_t n, unsigned int* amax_bits, float* scale, hipStream_t s);
hipError_t launch_scale_inplace(float* x, int64_t n, const float* scale, hipStream_t s);

// OT-ODE per-pixel steps (pointwise.hip)
hipError_t launch_ot_ode_vec(const DegView& d, const float* x, const float* vt, const float* y, const float* one_minus_t,
                             const float* rt2, float sigma2, float* vec, int B, int C, int H, int W, hipStream_t s);
hipError_t launch_ot_ode_vec_blur(const DegView& d, const float* x, const float* vt, const float* y, const float* one_minus_t,
                                  const float* rt2, float sigma2, float* vec, int B, int C, int H, int W, float* scratch, hipStream_t s);
hipError_t launch_ot_ode_update(float* x, const float* vt, const float* vec, const float* g, const float* one_minus_t, const float* coef,
                                float delta, int B, int n, hipStream_t s);

// ---- NCSN++ ("rectified") velocity net, the ops that are not convs (ncsnpp_ops.hip) ----------------------------------------
// FIR resampling of an NHWC activation (upfirdn2d of up_or_down_sampling.py:204-259 on every channel), optionally of TWO views of
// the same source in one pass: `out_act` takes act(GroupNorm(src)) (the per-image sc/sh of `coef`, SiLU), `out_raw` the raw
// values (+ their per-channel statistics for the consumer's range guard)  -  ResnetBlockBigGANpp.forward, layerspp.py:238-254.
struct FirParams {
    const float* src;              // NHWC [B][Hs][Ws][C]
    float* out_act; float* out_raw;          // NHWC [B][H][W][C], either may be nullptr
    const float* coef; int coef_stride;      // ConvParams::coef layout; needed for out_act
    double* stats_raw;             // [B][C][2] += (sum, sumsq) of out_raw, or nullptr
    int B, Hs, Ws, H, W, C;
    int up, down, pad0, K;         // same along both axes
    float k2d[64];                 // [K][K], already normalised / gained (up_or_down_sampling.py:197-204)
};
hipError_t launch_fir_nhwc(const FirParams& p, hipStream_t s);

// time conditioning of NCSNpp.forward (ncsnpp.py:223-246) + every block's Dense_0(act(temb)) (layerspp.py:258-260) in one launch:
// out[b][j] = bp[j] + Wp[j] . silu(Linear1(silu(Linear0([sin, cos](2 pi W log(t[b] * t_scale))))))
struct NxTembParams {
    const float* t; float t_scale;
    const float* Wf;               // [nf] GaussianFourierProjection.W (layerspp.py:31-41)
    const float* w0; const float* b0;        // Linear(2nf -> 4nf)  (out, in)
    const float* w1; const float* b1;        // Linear(4nf -> 4nf)
    const float* wp; const float* bp;        // stacked Dense_0 weights [total_out][4nf], biases (+ Conv_0.bias)
    float* out; int B, nf, total_out;
};
hipError_t launch_nx_temb(const NxTembParams& p, hipStream_t s);

// image boundary: NCHW image -> zero-padded NHWC-32 operand of the MFMA conv, and back (the first Cimg channels of the NHWC-32
// output-skip pyramid, divided by sigma = t * t_scale when scale_by_sigma; ncsnpp.py:378-381)
hipError_t launch_img_to_nhwc32(const float* img, float* out, int B, int Cimg, int H, int W, hipStream_t s, const float* div = nullptr /* [B]: image b is divided by div[b] */);
hipError_t launch_nhwc32_to_img(const float* in, float* img, const float* t, float t_scale, int scale_by_sigma, int B, int Cimg, int H, int W,
                                hipStream_t s, float* sigma_out = nullptr /* [B] <- the divisor used (kept for the VJP) */);

}  // namespace pf
