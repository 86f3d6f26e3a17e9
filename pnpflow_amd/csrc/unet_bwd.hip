// Backward (input-gradient only) helper kernels of the U-Net velocity field, used by the
// vector-Jacobian product that OT-ODE needs (reference: torch.autograd.functional.vjp through
// pnpflow/models.py:94-162, 442-495, called at pnpflow/methods/ot_ode.py:137-138).
// The dense parts of the backward (conv / 1x1 / attention matmul transposes) reuse
// conv_mfma_kernel with transposed-flipped weight repacks; this file holds the GroupNorm(+SiLU)
// backward, softmax backward, transposes and the 2x2 sum-pool (adjoint of nearest upsampling).
#include <algorithm>
#include <cstdlib>
#include "pf_common.h"

namespace pf {

// ---- forward GroupNorm coefficients per (b, channel) of a (possibly concatenated) input ------
//   mu[b][c], rs[b][c]  (replicated over the channels of a group)
__global__ __launch_bounds__(256) void gn_fwd_coeffs_kernel(const double* st0, int C0, const double* st1, int C1, int cpg, double n_per_ch,
                                                            float eps, float* mu, float* rs) {
    const int b = blockIdx.x, Ct = C0 + C1;
    for (int c = threadIdx.x; c < Ct; c += 256) {
        const int g = c / cpg;
        double s = 0.0, ss = 0.0;
        for (int j = g * cpg; j < (g + 1) * cpg; ++j) {
            const double* st = j < C0 ? st0 + ((size_t)b * C0 + j) * 2 : st1 + ((size_t)b * C1 + (j - C0)) * 2;
            s += st[0]; ss += st[1];
        }
        const double N = n_per_ch * cpg;
        const double mean = s / N;
        double var = ss / N - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mu[(size_t)b * Ct + c] = (float)mean;
        rs[(size_t)b * Ct + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ---- stage A of GN(+SiLU) backward: da -> dyhat = da * act'(u) * gamma, per-channel sums ------
//   x, g: NHWC [B][HW][C] slices of the concatenated channel space starting at coff (of Ct)
//   bsum[b][Ct][2] += (sum dyhat, sum dyhat*yhat)
__global__ __launch_bounds__(256) void gn_bwd_pre_kernel(float* g, const float* x, const float* mu, const float* rs, const float* gamma,
                                                         const float* beta, double* bsum, int HW, int C, int coff, int Ct, int silu,
                                                         int pix_per_block) {
    __shared__ float s_red[256 * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cq = C / 4, lanes_p = 256 / cq;
    const int q = tid % cq, pr = tid / cq;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pr < lanes_p) {
        const int cg = coff + q * 4;
        const float4 m4 = *reinterpret_cast<const float4*>(mu + (size_t)b * Ct + cg);
        const float4 r4 = *reinterpret_cast<const float4*>(rs + (size_t)b * Ct + cg);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + cg);
        const float4 be = *reinterpret_cast<const float4*>(beta + cg);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
        constexpr int UNR = 4;                    // pixels in flight per thread: 8 float4 loads before the first use
        for (int pix = p0 + pr; pix < p1; pix += lanes_p * UNR) {
            float4 xv[UNR], gv[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int pk = min(pix + k * lanes_p, p1 - 1);                       // clamped: unconditional loads
                const size_t o = ((size_t)b * HW + pk) * C + q * 4;
                xv[k] = *reinterpret_cast<const float4*>(x + o);
                gv[k] = *reinterpret_cast<const float4*>(g + o);
            }
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                if (pix + k * lanes_p < p1) {
                    const float xs[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
                    float gs[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float yh = (xs[j] - mm[j]) * rr[j];
                        float d = gs[j];
                        if (silu) {
                            const float u = yh * gg[j] + bb[j];
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                            d *= sg * (1.0f + u * (1.0f - sg));
                        }
                        d *= gg[j];
                        gs[j] = d;
                        s1[j] += d; s2[j] += d * yh;
                    }
                    *reinterpret_cast<float4*>(g + ((size_t)b * HW + pix + k * lanes_p) * C + q * 4) = make_float4(gs[0], gs[1], gs[2], gs[3]);
                }
            }
        }
    }
    for (int j = 0; j < 4; ++j) {
        __syncthreads();
        s_red[tid * 2] = s1[j]; s_red[tid * 2 + 1] = s2[j];
        __syncthreads();
        if (pr == 0 && tid < cq) {
            double a = 0, c2 = 0;
            for (int r = 0; r < lanes_p; ++r) { a += s_red[(r * cq + q) * 2]; c2 += s_red[(r * cq + q) * 2 + 1]; }
            double* st = bsum + ((size_t)b * Ct + coff + q * 4 + j) * 2;
            unsafeAtomicAdd(st, a);
            unsafeAtomicAdd(st + 1, c2);
        }
    }
}

// group means of the stage-A sums:  m1[b][c] = sum_group(dyhat)/N, m2[b][c] = sum_group(dyhat*yhat)/N
__global__ __launch_bounds__(256) void gn_bwd_coeffs_kernel(const double* bsum, int Ct, int cpg, double n_per_ch, float* m1, float* m2) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < Ct; c += 256) {
        const int g = c / cpg;
        double a = 0.0, d = 0.0;
        for (int j = g * cpg; j < (g + 1) * cpg; ++j) { a += bsum[((size_t)b * Ct + j) * 2]; d += bsum[((size_t)b * Ct + j) * 2 + 1]; }
        const double N = n_per_ch * cpg;
        m1[(size_t)b * Ct + c] = (float)(a / N);
        m2[(size_t)b * Ct + c] = (float)(d / N);
    }
}

// ---- stage B: out (=|+=) rs * (dyhat - m1 - yhat*m2) (+ add) -----------------------------------
// (HAS_ADD / ACC are compile-time: a load inside a run-time branch makes hipcc wait for it on the spot - vmcnt(0) - which
// serialised the 8-16 loads a thread keeps in flight; profiles/r02_kernel_trace_bench_c5.md: 3.2 TB/s before)
template <bool HAS_ADD, bool ACC>
__global__ __launch_bounds__(256) void gn_bwd_post_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ rs, const float* __restrict__ m1, const float* __restrict__ m2,
                                                          const float* __restrict__ add, float* out, int HW, int C, int coff, int Ct,
                                                          int pix_per_block, float add_scale) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cq = C / 4, lanes_p = 256 / cq;
    const int q = tid % cq, pr = tid / cq;
    if (pr >= lanes_p) return;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    const int cg = coff + q * 4;
    const float4 m4 = *reinterpret_cast<const float4*>(mu + (size_t)b * Ct + cg);
    const float4 r4 = *reinterpret_cast<const float4*>(rs + (size_t)b * Ct + cg);
    const float4 a4 = *reinterpret_cast<const float4*>(m1 + (size_t)b * Ct + cg);
    const float4 b4 = *reinterpret_cast<const float4*>(m2 + (size_t)b * Ct + cg);
    // dy (read once, dead afterwards) through non-temporal loads, the result through non-temporal stores: -0.6 % on the 256^2 VJP, twice
    // on one box (profiles/r03_ab_gn_bwd_post.txt); unroll depth 2 / 4 / 8: no difference
    constexpr int UNR = 4;
    for (int pix = p0 + pr; pix < p1; pix += lanes_p * UNR) {
        float4 xv[UNR], dv[UNR], av[UNR], ov[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int pk = min(pix + k * lanes_p, p1 - 1);
            const size_t o = ((size_t)b * HW + pk) * C + q * 4;
            xv[k] = *reinterpret_cast<const float4*>(x + o);
            { typedef float f4v __attribute__((ext_vector_type(4)));
              const f4v t_ = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(dy + o)); dv[k] = make_float4(t_.x, t_.y, t_.z, t_.w); }
            if constexpr (HAS_ADD) av[k] = *reinterpret_cast<const float4*>(add + o);
            if constexpr (ACC) ov[k] = *reinterpret_cast<const float4*>(out + o);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (pix + k * lanes_p < p1) {
                float4 r;
                r.x = r4.x * (dv[k].x - a4.x - (xv[k].x - m4.x) * r4.x * b4.x);
                r.y = r4.y * (dv[k].y - a4.y - (xv[k].y - m4.y) * r4.y * b4.y);
                r.z = r4.z * (dv[k].z - a4.z - (xv[k].z - m4.z) * r4.z * b4.z);
                r.w = r4.w * (dv[k].w - a4.w - (xv[k].w - m4.w) * r4.w * b4.w);
                if constexpr (HAS_ADD) { r.x = fmaf(av[k].x, add_scale, r.x); r.y = fmaf(av[k].y, add_scale, r.y); r.z = fmaf(av[k].z, add_scale, r.z); r.w = fmaf(av[k].w, add_scale, r.w); }
                if constexpr (ACC) { r.x += ov[k].x; r.y += ov[k].y; r.z += ov[k].z; r.w += ov[k].w; }
                { typedef float f4v __attribute__((ext_vector_type(4)));
                  f4v t_ = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(t_, reinterpret_cast<f4v*>(out + ((size_t)b * HW + pix + k * lanes_p) * C + q * 4)); }
            }
        }
    }
}


hipError_t launch_gn_fwd_coeffs(const double* st0, int C0, const double* st1, int C1, int cpg, int HW, float eps, float* mu, float* rs, int B,
                                hipStream_t s) {
    hipLaunchKernelGGL(gn_fwd_coeffs_kernel, dim3(B), dim3(256), 0, s, st0, C0, st1, C1, cpg, (double)HW, eps, mu, rs);
    return hipGetLastError();
}
hipError_t launch_gn_bwd_pre(float* g, const float* x, const float* mu, const float* rs, const float* gamma, const float* beta, double* bsum,
                             int B, int HW, int C, int coff, int Ct, int silu, hipStream_t s) {
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;
    int ppb = 256;
    while (ppb > 64 && (long)((HW + ppb - 1) / ppb) * B < 2048) ppb >>= 1;
    hipLaunchKernelGGL(gn_bwd_pre_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(256), 0, s, g, x, mu, rs, gamma, beta, bsum, HW, C, coff, Ct, silu, ppb);
    return hipGetLastError();
}
hipError_t launch_gn_bwd_coeffs(const double* bsum, int Ct, int cpg, int HW, float* m1, float* m2, int B, hipStream_t s) {
    hipLaunchKernelGGL(gn_bwd_coeffs_kernel, dim3(B), dim3(256), 0, s, bsum, Ct, cpg, (double)HW, m1, m2);
    return hipGetLastError();
}
hipError_t launch_gn_bwd_post(const float* dy, const float* x, const float* mu, const float* rs, const float* m1, const float* m2,
                              const float* add, float* out, int B, int HW, int C, int coff, int Ct, int accumulate, hipStream_t s, float add_scale) {
    if (C % 4 || C / 4 > 256) return hipErrorInvalidValue;
    // pixels per workgroup: >= 8 workgroups per CU where the tensor allows it (1024 left the second wave of workgroups 60 % full at
    // 256^2: C5 5.27 -> 5.54 images/s with 256, profiles/r02_ab_variants_same_box.txt)
    int ppb = 256;
    while (ppb > 64 && (long)((HW + ppb - 1) / ppb) * B < 2048) ppb >>= 1;
    const dim3 grid((HW + ppb - 1) / ppb, B);
    if (add != nullptr && accumulate) hipLaunchKernelGGL((gn_bwd_post_kernel<true, true>), grid, dim3(256), 0, s, dy, x, mu, rs, m1, m2, add, out, HW, C, coff, Ct, ppb, add_scale);
    else if (add != nullptr) hipLaunchKernelGGL((gn_bwd_post_kernel<true, false>), grid, dim3(256), 0, s, dy, x, mu, rs, m1, m2, add, out, HW, C, coff, Ct, ppb, add_scale);
    else if (accumulate) hipLaunchKernelGGL((gn_bwd_post_kernel<false, true>), grid, dim3(256), 0, s, dy, x, mu, rs, m1, m2, add, out, HW, C, coff, Ct, ppb, add_scale);
    else hipLaunchKernelGGL((gn_bwd_post_kernel<false, false>), grid, dim3(256), 0, s, dy, x, mu, rs, m1, m2, add, out, HW, C, coff, Ct, ppb, add_scale);
    return hipGetLastError();
}

// ---- batched transpose [B][R][Cc] -> [B][Cc][R] ---------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int R, int Cc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = in[((size_t)b * R + r0 + j) * Cc + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) out[((size_t)b * Cc + c0 + j) * R + r0 + tx] = tile[tx][j];
}
hipError_t launch_transpose(const float* in, float* out, int B, int R, int Cc, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, B), dim3(256), 0, s, in, out, R, Cc);
    return hipGetLastError();
}

// ---- softmax backward, in place over dA: dS = scale * A .* (dA - rowsum(dA .* A)) -----------------
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* A, float* dA, int64_t rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* a = A + row * cols;
    float* d = dA + row * cols;
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += a[i] * d[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    for (int i = lane; i < cols; i += 64) d[i] = scale * a[i] * (d[i] - s);
}
hipError_t launch_softmax_bwd(const float* A, float* dA, int64_t rows, int cols, float scale, hipStream_t s) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, A, dA, rows, cols, scale);
    return hipGetLastError();
}

// ---- adjoint of nearest-x2 upsampling: out[b][y][x][c] (=|+=) sum of the 2x2 block ---------------
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* in, float* out, int H, int W, int C, int accumulate) {
    const int b = blockIdx.y;
    const int cq = C / 4;
    const size_t n4 = (size_t)H * W * cq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int q = i % cq;
        const size_t pix = i / cq;
        const int x = pix % W, y = pix / W;
        const float* base = in + (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + q * 4;
        const float4 a = *reinterpret_cast<const float4*>(base), c = *reinterpret_cast<const float4*>(base + C);
        const float4 d = *reinterpret_cast<const float4*>(base + (size_t)2 * W * C), e = *reinterpret_cast<const float4*>(base + (size_t)2 * W * C + C);
        float4 r = make_float4(a.x + c.x + d.x + e.x, a.y + c.y + d.y + e.y, a.z + c.z + d.z + e.z, a.w + c.w + d.w + e.w);
        float* o = out + ((size_t)b * H * W + pix) * C + q * 4;
        if (accumulate) { const float4 ov = *reinterpret_cast<const float4*>(o); r.x += ov.x; r.y += ov.y; r.z += ov.z; r.w += ov.w; }
        *reinterpret_cast<float4*>(o) = r;
    }
}
hipError_t launch_sumpool2(const float* in, float* out, int B, int H, int W, int C, int accumulate, hipStream_t s) {
    if (C % 4) return hipErrorInvalidValue;
    const size_t n4 = (size_t)H * W * (C / 4);
    hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048), B), dim3(256), 0, s, in, out, H, W, C, accumulate);
    return hipGetLastError();
}

}  // namespace pf
