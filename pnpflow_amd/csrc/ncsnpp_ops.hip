// The non-conv pieces of the NCSN++ ("rectified") velocity net: everything else of
// pnpflow/image_generation/models/ncsnpp.py runs on the MFMA conv kernels (engine_ncsnpp.inc).
//
//   fir_nhwc_kernel        upsample_2d / downsample_2d (up_or_down_sampling.py:204-259) of a channels-last activation; one pass
//                          produces both operands of a BigGAN block (layerspp.py:238-254): FIR(SiLU(GroupNorm_0(x))) and FIR(x).
//                          HBM-bound: reads x once (taps re-read from L2), writes each output once.
//   nx_temb_kernel         Gaussian Fourier features of log(sigma) -> 2-layer MLP -> every block's Dense_0 (ncsnpp.py:223-246)
//   img_to_nhwc32 / back   image boundary of the net
#include <algorithm>
#include <cstdlib>
#include "pf_common.h"

namespace pf {

__device__ __forceinline__ int nx_floor_div(int a, int b) { const int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
__device__ __forceinline__ int nx_pos_mod(int a, int b) { const int r = a % b; return r < 0 ? r + b : r; }
__device__ __forceinline__ float nx_silu(float u) { return u / (1.0f + __expf(-u)); }

constexpr int FIR_PPB = 512;      // output pixels per workgroup (one fp64 atomic per channel and workgroup for the statistics)

template <bool ACT, bool RAW>
__global__ __launch_bounds__(256) void fir_nhwc_kernel(const FirParams p) {
    __shared__ float s_k[64];
    __shared__ float s_sum[2 * 512];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cq = p.C / 4, lanes_p = 256 / cq;
    // flipped taps: out = sum P[oy*down + ky][ox*down + kx] * K[K-1-ky][K-1-kx]   (op/upfirdn2d.py:170-171)
    if (tid < p.K * p.K) s_k[tid] = p.k2d[(p.K - 1 - tid / p.K) * p.K + (p.K - 1 - tid % p.K)];
    if (RAW && p.stats_raw != nullptr) for (int i = tid; i < 2 * p.C; i += 256) s_sum[i] = 0.f;
    __syncthreads();
    const int q = tid % cq, pr = tid / cq;
    const int HWo = p.H * p.W;
    const int p0 = blockIdx.x * FIR_PPB, p1 = min(HWo, p0 + FIR_PPB);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACT && pr < lanes_p) {
        sc = *reinterpret_cast<const float4*>(p.coef + ((size_t)b * 2 + 0) * p.coef_stride + q * 4);
        sh = *reinterpret_cast<const float4*>(p.coef + ((size_t)b * 2 + 1) * p.coef_stride + q * 4);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (pr < lanes_p) {
        const float* src = p.src + (size_t)b * p.Hs * p.Ws * p.C + q * 4;
        for (int pix = p0 + pr; pix < p1; pix += lanes_p) {
            const int oy = pix / p.W, ox = pix % p.W;
            // only taps that meet a real (not zero-inserted) sample: (o*down + k - pad0) % up == 0
            const int by = oy * p.down - p.pad0, bx = ox * p.down - p.pad0;
            const int ky0 = nx_pos_mod(-by, p.up), kx0 = nx_pos_mod(-bx, p.up);
            float4 aa = make_float4(0.f, 0.f, 0.f, 0.f), ar = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ky = ky0; ky < p.K; ky += p.up) {
                const int iy = (by + ky) / p.up;
                if (by + ky < 0 || iy >= p.Hs) continue;
                for (int kx = kx0; kx < p.K; kx += p.up) {
                    const int ix = (bx + kx) / p.up;
                    if (bx + kx < 0 || ix >= p.Ws) continue;
                    const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)iy * p.Ws + ix) * p.C);
                    const float w = s_k[ky * p.K + kx];
                    if (RAW) { ar.x = fmaf(v.x, w, ar.x); ar.y = fmaf(v.y, w, ar.y); ar.z = fmaf(v.z, w, ar.z); ar.w = fmaf(v.w, w, ar.w); }
                    if (ACT) {
                        aa.x = fmaf(nx_silu(fmaf(v.x, sc.x, sh.x)), w, aa.x); aa.y = fmaf(nx_silu(fmaf(v.y, sc.y, sh.y)), w, aa.y);
                        aa.z = fmaf(nx_silu(fmaf(v.z, sc.z, sh.z)), w, aa.z); aa.w = fmaf(nx_silu(fmaf(v.w, sc.w, sh.w)), w, aa.w);
                    }
                }
            }
            const size_t o = ((size_t)b * HWo + pix) * p.C + q * 4;
            if (ACT) *reinterpret_cast<float4*>(p.out_act + o) = aa;
            if (RAW) {
                *reinterpret_cast<float4*>(p.out_raw + o) = ar;
                s1[0] += ar.x; s1[1] += ar.y; s1[2] += ar.z; s1[3] += ar.w;
                s2[0] += ar.x * ar.x; s2[1] += ar.y * ar.y; s2[2] += ar.z * ar.z; s2[3] += ar.w * ar.w;
            }
        }
    }
    if (RAW && p.stats_raw != nullptr) {
        if (pr < lanes_p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(&s_sum[(q * 4 + j) * 2], s1[j]); atomicAdd(&s_sum[(q * 4 + j) * 2 + 1], s2[j]); }
        }
        __syncthreads();
        for (int i = tid; i < 2 * p.C; i += 256) unsafeAtomicAdd(p.stats_raw + (size_t)b * p.C * 2 + i, (double)s_sum[i]);
    }
}

// The two shapes the net uses - 4 x 4 taps with (up 2, pad0 2) or (down 2, pad0 1), which are also each other's adjoints - with
// compile-time tap loops and no divisions: a workgroup walks whole output rows (row validity is uniform), a thread one (pixel,
// channel quad) per step.  Per output: 2 x 2 taps (up) / 4 x 4 taps (down), all requested before the first use.
template <bool ACT, bool RAW, int UP>
__global__ __launch_bounds__(256) void fir4_nhwc_kernel(const FirParams p) {
    constexpr int DOWN = UP == 2 ? 1 : 2, PAD0 = UP == 2 ? 2 : 1, NT = UP == 2 ? 2 : 4, SH = UP == 2 ? 1 : 0;
    __shared__ float s_k[16];
    __shared__ float s_sum[2 * 512];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cq = p.C / 4, lanes_p = 256 / cq;
    if (tid < 16) s_k[tid] = p.k2d[(3 - tid / 4) * 4 + (3 - tid % 4)];       // flipped taps (op/upfirdn2d.py:170-171)
    if (RAW && p.stats_raw != nullptr) for (int i = tid; i < 2 * p.C; i += 256) s_sum[i] = 0.f;
    __syncthreads();
    const int q = tid % cq, pr = tid / cq;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACT && pr < lanes_p) {
        sc = *reinterpret_cast<const float4*>(p.coef + ((size_t)b * 2 + 0) * p.coef_stride + q * 4);
        sh = *reinterpret_cast<const float4*>(p.coef + ((size_t)b * 2 + 1) * p.coef_stride + q * 4);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int rows = max(1, FIR_PPB / p.W);
    const int oy_lo = blockIdx.x * rows, oy_hi = min(p.H, oy_lo + rows);
    if (pr < lanes_p) {
        const float* src = p.src + (size_t)b * p.Hs * p.Ws * p.C + q * 4;
        for (int oy = oy_lo; oy < oy_hi; ++oy) {
            const int by = oy * DOWN - PAD0, ky0 = UP == 2 ? (by & 1) : 0;
            for (int ox = pr; ox < p.W; ox += lanes_p) {
                const int bx = ox * DOWN - PAD0, kx0 = UP == 2 ? (bx & 1) : 0;
                float4 v[NT][NT];
                unsigned inside = 0u;                      // bit a * NT + c: tap (a, c) lies inside the image
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const int iy = (by + ky0 + a * UP) >> SH;
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        const int ix = (bx + kx0 + c * UP) >> SH;
                        const bool ok = iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
                        v[a][c] = ok ? *reinterpret_cast<const float4*>(src + ((size_t)iy * p.Ws + ix) * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) inside |= 1u << (a * NT + c);
                    }
                }
                float4 aa = make_float4(0.f, 0.f, 0.f, 0.f), ar = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        const float w = s_k[(ky0 + a * UP) * 4 + kx0 + c * UP];
                        const float4 x = v[a][c];
                        if (ACT) {
                            if (inside & (1u << (a * NT + c))) {         // a tap outside the image contributes nothing (SiLU(sh) would)
                                aa.x = fmaf(nx_silu(fmaf(x.x, sc.x, sh.x)), w, aa.x); aa.y = fmaf(nx_silu(fmaf(x.y, sc.y, sh.y)), w, aa.y);
                                aa.z = fmaf(nx_silu(fmaf(x.z, sc.z, sh.z)), w, aa.z); aa.w = fmaf(nx_silu(fmaf(x.w, sc.w, sh.w)), w, aa.w);
                                if (RAW) { ar.x = fmaf(x.x, w, ar.x); ar.y = fmaf(x.y, w, ar.y); ar.z = fmaf(x.z, w, ar.z); ar.w = fmaf(x.w, w, ar.w); }
                            }
                        } else if (RAW) {
                            ar.x = fmaf(x.x, w, ar.x); ar.y = fmaf(x.y, w, ar.y); ar.z = fmaf(x.z, w, ar.z); ar.w = fmaf(x.w, w, ar.w);
                        }
                    }
                const size_t o = (((size_t)b * p.H + oy) * p.W + ox) * p.C + q * 4;
                if (ACT) *reinterpret_cast<float4*>(p.out_act + o) = aa;
                if (RAW) {
                    *reinterpret_cast<float4*>(p.out_raw + o) = ar;
                    s1[0] += ar.x; s1[1] += ar.y; s1[2] += ar.z; s1[3] += ar.w;
                    s2[0] += ar.x * ar.x; s2[1] += ar.y * ar.y; s2[2] += ar.z * ar.z; s2[3] += ar.w * ar.w;
                }
            }
        }
    }
    if (RAW && p.stats_raw != nullptr) {
        if (pr < lanes_p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(&s_sum[(q * 4 + j) * 2], s1[j]); atomicAdd(&s_sum[(q * 4 + j) * 2 + 1], s2[j]); }
        }
        __syncthreads();
        for (int i = tid; i < 2 * p.C; i += 256) unsafeAtomicAdd(p.stats_raw + (size_t)b * p.C * 2 + i, (double)s_sum[i]);
    }
}

template <int UP>
static void launch_fir4(const FirParams& p, hipStream_t s) {
    const int rows = std::max(1, FIR_PPB / p.W);
    const dim3 grid((p.H + rows - 1) / rows, p.B);
    if (p.out_act && p.out_raw) hipLaunchKernelGGL((fir4_nhwc_kernel<true, true, UP>), grid, dim3(256), 0, s, p);
    else if (p.out_act) hipLaunchKernelGGL((fir4_nhwc_kernel<true, false, UP>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((fir4_nhwc_kernel<false, true, UP>), grid, dim3(256), 0, s, p);
}

hipError_t launch_fir_nhwc(const FirParams& p, hipStream_t s) {
    if (p.C % 4 || p.C / 4 > 128 || p.C < 4 || p.K < 1 || p.K > 8 || p.up < 1 || p.down < 1 || (p.out_act == nullptr && p.out_raw == nullptr) ||
        (p.out_act != nullptr && p.coef == nullptr))
        return hipErrorInvalidValue;
    static const bool generic_only = getenv("PNPFLOW_HIP_FIR_GENERIC") != nullptr;
    if (!generic_only && p.K == 4 && p.up == 2 && p.down == 1 && p.pad0 == 2) { launch_fir4<2>(p, s); return hipGetLastError(); }
    if (!generic_only && p.K == 4 && p.up == 1 && p.down == 2 && p.pad0 == 1) { launch_fir4<1>(p, s); return hipGetLastError(); }
    const dim3 grid((p.H * p.W + FIR_PPB - 1) / FIR_PPB, p.B);
    if (p.out_act && p.out_raw) hipLaunchKernelGGL((fir_nhwc_kernel<true, true>), grid, dim3(256), 0, s, p);
    else if (p.out_act) hipLaunchKernelGGL((fir_nhwc_kernel<true, false>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((fir_nhwc_kernel<false, true>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- time conditioning ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nx_silu_acc(float u) { return u / (1.0f + expf(-u)); }

__global__ __launch_bounds__(256) void nx_temb_kernel(const NxTembParams p) {
    __shared__ float s_e[256];
    __shared__ float s_h[512];
    __shared__ float s_s[512];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nf = p.nf, e2 = 2 * nf, tch = 4 * nf;
    if (tid < nf) {
        // temb = GaussianFourierProjection(log(used_sigmas))  (ncsnpp.py:228-230; layerspp.py:39-41): x_proj = x * W * 2 * pi in the
        // reference's fp32 evaluation order; log in fp64 then rounded (the correctly rounded fp32 logarithm)
        const float sig = p.t[b] * p.t_scale;
        const float x = (float)log((double)sig);
        const float a = ((x * p.Wf[tid]) * 2.0f) * 3.14159265358979323846f;
        s_e[tid] = sinf(a);
        s_e[nf + tid] = cosf(a);
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b0[j];
        for (int k = 0; k < e2; ++k) acc = fmaf(p.w0[(size_t)j * e2 + k], s_e[k], acc);
        s_h[j] = nx_silu_acc(acc);                      // modules[2](act(modules[1](temb)))  (ncsnpp.py:241-244)
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b1[j];
        for (int k = 0; k < tch; ++k) acc = fmaf(p.w1[(size_t)j * tch + k], s_h[k], acc);
        s_s[j] = nx_silu_acc(acc);                      // every block applies act(temb) before Dense_0 (layerspp.py:260)
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j < p.total_out) {
        const float4* w = reinterpret_cast<const float4*>(p.wp + (size_t)j * tch);
        float acc = p.bp[j];
        for (int k = 0; k < tch / 4; ++k) {
            const float4 wv = w[k];
            acc = fmaf(wv.x, s_s[4 * k], acc); acc = fmaf(wv.y, s_s[4 * k + 1], acc);
            acc = fmaf(wv.z, s_s[4 * k + 2], acc); acc = fmaf(wv.w, s_s[4 * k + 3], acc);
        }
        p.out[(size_t)b * p.total_out + j] = acc;
    }
}

hipError_t launch_nx_temb(const NxTembParams& p, hipStream_t s) {
    if (p.nf < 4 || p.nf > 128 || (p.nf & 3)) return hipErrorInvalidValue;
    dim3 grid((p.total_out + 255) / 256, p.B);
    hipLaunchKernelGGL(nx_temb_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- image boundary ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void img_to_nhwc32_kernel(const float* __restrict__ img, float* __restrict__ out, int Cimg, int HW, const float* __restrict__ div) {
    const int b = blockIdx.y;
    const float dv = div != nullptr ? div[b] : 1.0f;
    const size_t n4 = (size_t)HW * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & 7); const size_t pix = i >> 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q == 0) {
            const float* src = img + (size_t)b * Cimg * HW + pix;
            v.x = src[0];
            if (Cimg > 1) v.y = src[HW];
            if (Cimg > 2) v.z = src[2 * (size_t)HW];
            if (Cimg > 3) v.w = src[3 * (size_t)HW];
            if (div != nullptr) { v.x /= dv; v.y /= dv; v.z /= dv; v.w /= dv; }
        }
        reinterpret_cast<float4*>(out + (size_t)b * HW * 32)[i] = v;
    }
}
hipError_t launch_img_to_nhwc32(const float* img, float* out, int B, int Cimg, int H, int W, hipStream_t s, const float* div) {
    if (Cimg < 1 || Cimg > 4) return hipErrorInvalidValue;
    const size_t n4 = (size_t)H * W * 8;
    hipLaunchKernelGGL(img_to_nhwc32_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096), B), dim3(256), 0, s, img, out, Cimg, H * W, div);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nhwc32_to_img_kernel(const float* __restrict__ in, float* __restrict__ img, const float* __restrict__ t,
                                                          float t_scale, int scale_by_sigma, int Cimg, int HW, float* __restrict__ sigma_out) {
    const int b = blockIdx.y;
    const float sig = scale_by_sigma ? t[b] * t_scale : 1.0f;
    if (sigma_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) sigma_out[b] = sig;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)b * HW + pix) * 32);
        float* dst = img + (size_t)b * Cimg * HW + pix;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int c = 0; c < Cimg; ++c) dst[(size_t)c * HW] = scale_by_sigma ? vv[c] / sig : vv[c];      // h / used_sigmas (ncsnpp.py:380-381)
    }
}
hipError_t launch_nhwc32_to_img(const float* in, float* img, const float* t, float t_scale, int scale_by_sigma, int B, int Cimg, int H, int W,
                                hipStream_t s, float* sigma_out) {
    if (Cimg < 1 || Cimg > 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nhwc32_to_img_kernel, dim3((unsigned)std::min((H * W + 255) / 256, 4096), B), dim3(256), 0, s, in, img, t, t_scale,
                       scale_by_sigma, Cimg, H * W, sigma_out);
    return hipGetLastError();
}

}  // namespace pf
