// gfx950 versions of the reference's only native code: the two StyleGAN2 ops its NCSN++ ("rectified") velocity net calls
// (SURVEY.md 8f N4).
//
//   upfirdn2d        pnpflow/image_generation/op/upfirdn2d_kernel.cu:49-369 (+ op/upfirdn2d.py:142-187, the pure-torch
//                    definition the oracle restates):  zero-insert upsampling by (up_x, up_y) -> pad / crop by
//                    (pad_x0, pad_x1, pad_y0, pad_y1) -> 2-D FIR filter (true convolution: the kernel is flipped) -> decimation
//                    by (down_x, down_y), on every (batch x channel) plane of an NCHW tensor.  Used by upsample_2d /
//                    downsample_2d / the fused conv variants of models/up_or_down_sampling.py:142,178,225,258.
//   fused_bias_act   pnpflow/image_generation/op/fused_bias_act_kernel.cu:19-99 (fused_leaky_relu of op/fused_act.py:84-96):
//                    y = act(x + bias[channel]) * scale, act in {linear, leaky ReLU}, plus the two gradient modes of the
//                    reference kernel (first derivative w.r.t. a reference output, and the identically-zero second derivative).
//
// Both are HBM-bound: upfirdn2d reads each input sample once per workgroup tile (the tile's input footprint is staged in LDS,
// with the FIR taps; at most ceil(kh/up_y)*ceil(kw/up_x) multiply-adds per output) and writes each output once;
// fused_bias_act is a 16-byte-per-lane stream (8 B per element algorithmic).
#include <algorithm>
#include "pf_common.h"

namespace pf {

struct UpfirdnParams {
    const float* in; const float* kernel; float* out;
    int planes, in_h, in_w, out_h, out_w, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int tin_h, tin_w;         // LDS tile of input samples (0: no staging, read global memory directly)
};

__device__ __forceinline__ int floor_div(int a, int b) { const int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
__device__ __forceinline__ int pos_mod(int a, int b) { const int r = a % b; return r < 0 ? r + b : r; }

constexpr int UF_TW = 32, UF_TH = 8;

__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UpfirdnParams p) {
    extern __shared__ float s_uf[];           // [kh*kw taps (flipped)] [tin_h][tin_w]
    float* s_k = s_uf;
    float* s_x = s_uf + p.kh * p.kw;
    const int tid = threadIdx.x;
    const int plane = blockIdx.z;
    const int ox0 = blockIdx.x * UF_TW, oy0 = blockIdx.y * UF_TH;
    // flipped taps: out = sum_{ky,kx} P[oy*down + ky][ox*down + kx] * K[kh-1-ky][kw-1-kx]   (op/upfirdn2d.py:170-171)
    for (int i = tid; i < p.kh * p.kw; i += 256) s_k[i] = p.kernel[(p.kh - 1 - i / p.kw) * p.kw + (p.kw - 1 - i % p.kw)];
    // input footprint of the tile: P rows [oy0*down, (oy0+TH-1)*down + kh-1] map to input rows (y - pad_y0) / up
    const int iy_lo = max(0, floor_div(oy0 * p.down_y - p.pad_y0 + p.up_y - 1, p.up_y));
    const int ix_lo = max(0, floor_div(ox0 * p.down_x - p.pad_x0 + p.up_x - 1, p.up_x));
    const float* src = p.in + (size_t)plane * p.in_h * p.in_w;
    if (p.tin_h > 0) {
        for (int i = tid; i < p.tin_h * p.tin_w; i += 256) {
            const int iy = iy_lo + i / p.tin_w, ix = ix_lo + i % p.tin_w;
            s_x[i] = (iy < p.in_h && ix < p.in_w) ? src[(size_t)iy * p.in_w + ix] : 0.f;
        }
    }
    __syncthreads();
    const int ox = ox0 + tid % UF_TW, oy = oy0 + tid / UF_TW;
    if (ox >= p.out_w || oy >= p.out_h) return;
    // only the taps that meet a real (non zero-inserted) sample: ky = ky0 + j*up_y with (oy*down_y + ky - pad_y0) % up_y == 0
    const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
    const int ky0 = pos_mod(-by, p.up_y), kx0 = pos_mod(-bx, p.up_x);
    float acc = 0.f;
    for (int ky = ky0; ky < p.kh; ky += p.up_y) {
        const int iy = (by + ky) / p.up_y;               // exact: by + ky is a multiple of up_y (may be negative -> outside)
        if (by + ky < 0 || iy >= p.in_h) continue;
        for (int kx = kx0; kx < p.kw; kx += p.up_x) {
            const int ix = (bx + kx) / p.up_x;
            if (bx + kx < 0 || ix >= p.in_w) continue;
            const float v = p.tin_h > 0 ? s_x[(iy - iy_lo) * p.tin_w + (ix - ix_lo)] : src[(size_t)iy * p.in_w + ix];
            acc = fmaf(v, s_k[ky * p.kw + kx], acc);
        }
    }
    p.out[((size_t)plane * p.out_h + oy) * p.out_w + ox] = acc;
}

hipError_t launch_upfirdn2d(const float* in, const float* kernel, float* out, int planes, int in_h, int in_w, int kh, int kw,
                            int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, hipStream_t s) {
    if (planes <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh * kw > 4096 || up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1)
        return hipErrorInvalidValue;
    UpfirdnParams p{};
    p.in = in; p.kernel = kernel; p.out = out; p.planes = planes; p.in_h = in_h; p.in_w = in_w; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    const int ph = in_h * up_y + pad_y0 + pad_y1, pw = in_w * up_x + pad_x0 + pad_x1;
    if (ph < kh || pw < kw) return hipErrorInvalidValue;
    p.out_h = (ph - kh) / down_y + 1; p.out_w = (pw - kw) / down_x + 1;       // op/upfirdn2d.py:184-185
    p.tin_h = ((UF_TH - 1) * down_y + kh - 1) / up_y + 2; p.tin_w = ((UF_TW - 1) * down_x + kw - 1) / up_x + 2;
    size_t lds = ((size_t)kh * kw + (size_t)p.tin_h * p.tin_w) * sizeof(float);
    if (lds > 60 * 1024) { p.tin_h = 0; p.tin_w = 0; lds = (size_t)kh * kw * sizeof(float); }      // huge footprints: direct reads
    const dim3 grid((p.out_w + UF_TW - 1) / UF_TW, (p.out_h + UF_TH - 1) / UF_TH, planes);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(upfirdn2d_kernel, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// y = act(x + b[(i / step_b) % size_b]) * scale;  act 1: linear, 3: leaky ReLU(alpha);  grad 0: value, 1: derivative w.r.t. `ref`'s
// sign (the saved forward output), 2: zero (fused_bias_act_kernel.cu:30-45)
__device__ __forceinline__ float fba_one(float x, float ref, int act, int grad, float alpha) {
    if (grad == 2) return 0.f;
    if (act == 3) return ((grad == 0 ? x : ref) > 0.f) ? x : x * alpha;
    return x;
}

__global__ __launch_bounds__(256) void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ ref,
                                                           float* __restrict__ out, int64_t n, int step_b, int size_b, int act, int grad,
                                                           float alpha, float scale, int vec) {
    if (vec) {        // step_b % 4 == 0: the 4 elements of a quad share their bias
        const int64_t n4 = n >> 2;
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
            float4 v = reinterpret_cast<const float4*>(x)[q];
            const float bb = b != nullptr ? b[((q * 4) / step_b) % size_b] : 0.f;
            const float4 r = ref != nullptr ? reinterpret_cast<const float4*>(ref)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            v = make_float4(fba_one(v.x + bb, r.x, act, grad, alpha) * scale, fba_one(v.y + bb, r.y, act, grad, alpha) * scale,
                            fba_one(v.z + bb, r.z, act, grad, alpha) * scale, fba_one(v.w + bb, r.w, act, grad, alpha) * scale);
            reinterpret_cast<float4*>(out)[q] = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float bb = b != nullptr ? b[(i / step_b) % size_b] : 0.f;
        out[i] = fba_one(x[i] + bb, ref != nullptr ? ref[i] : 0.f, act, grad, alpha) * scale;
    }
}

hipError_t launch_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t n, int step_b, int size_b,
                                 int act, int grad, float alpha, float scale, hipStream_t s) {
    if (n < 0 || (act != 1 && act != 3) || grad < 0 || grad > 2 || (bias != nullptr && (step_b <= 0 || size_b <= 0))) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const bool vec = (n & 3) == 0 && (bias == nullptr || step_b % 4 == 0) &&
                     ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)ref) & 15) == 0);
    const int64_t work = vec ? n / 4 : n;
    hipLaunchKernelGGL(fused_bias_act_kernel, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 8192)), dim3(256), 0, s, x, bias, ref, out, n,
                       step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1, act, grad, alpha, scale, vec ? 1 : 0);
    return hipGetLastError();
}

}  // namespace pf
