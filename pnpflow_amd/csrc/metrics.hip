// Image-quality metrics of the restoration path that need more than a per-pixel pass.
//
// SSIM as the reference computes it (pnpflow/utils.py:780-802): ignite.metrics.SSIM(data_range=1.0) with its defaults -
// 11x11 Gaussian window (sigma 1.5, outer product of the normalised 1-D taps), k1 = 0.01, k2 = 0.03, inputs reflect-padded
// by 5 pixels, per-channel filtering of x, y, x^2, y^2, xy, the SSIM map averaged over (C, H, W) per image in fp64.
// ignite is not installed in the build container, so this restatement follows ignite's published algorithm and is PARITY
// UNPINNED (said so in DESIGN.md); the oracle (oracle/pnpflow_oracle.py ssim_per_image) restates the same algorithm with
// torch ops and the GPU tests compare the two.
//
// HBM-bound: each image pair is read once (plus a 5-pixel halo per 32x16 tile: 1.9x), the five filtered fields never leave
// LDS (separable: 26x(16..32) horizontal pass into LDS, vertical pass in registers).
#include "pf_common.h"

namespace pf {

constexpr int SS_R = 5, SS_K = 11;          // window radius / taps
constexpr int SS_TW = 32, SS_TH = 8;        // output tile (256 threads = 32 x 8 pixels)
constexpr int SS_PW = SS_TW + 2 * SS_R, SS_PH = SS_TH + 2 * SS_R;

__device__ __forceinline__ int reflect_idx(int i, int n) {     // torch 'reflect' padding: the edge sample is not repeated
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ rec, const float* __restrict__ clean, double* __restrict__ out,
                                                  int C, int H, int W, float c1, float c2) {
    __shared__ float s_x[SS_PH][SS_PW + 1], s_y[SS_PH][SS_PW + 1];
    __shared__ float s_h[5][SS_PH][SS_TW + 1];
    __shared__ double s_red[4];
    const int tiles_x = (W + SS_TW - 1) / SS_TW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int ox0 = tx * SS_TW, oy0 = ty * SS_TH;
    const size_t plane = ((size_t)b * C + c) * H * W;
    // 1-D taps: exp(-0.5 (k/sigma)^2), k = -5..5, normalised (ignite _gaussian)
    float g[SS_K]; float gs = 0.f;
#pragma unroll
    for (int k = 0; k < SS_K; ++k) { const float d = (float)(k - SS_R) / 1.5f; g[k] = __expf(-0.5f * d * d); gs += g[k]; }
#pragma unroll
    for (int k = 0; k < SS_K; ++k) g[k] /= gs;

    for (int i = threadIdx.x; i < SS_PH * SS_PW; i += 256) {
        const int py = i / SS_PW, px = i % SS_PW;
        const int gy = reflect_idx(min(oy0 - SS_R + py, H - 1 + SS_R), H), gx = reflect_idx(min(ox0 - SS_R + px, W - 1 + SS_R), W);
        const size_t o = plane + (size_t)gy * W + gx;
        s_x[py][px] = (rec[o] + 1.0f) * 0.5f;          // postprocess (utils.py:560-577)
        s_y[py][px] = (clean[o] + 1.0f) * 0.5f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_PH * SS_TW; i += 256) {
        const int py = i / SS_TW, px = i % SS_TW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_K; ++k) {
            const float xv = s_x[py][px + k], yv = s_y[py][px + k];
            a0 = fmaf(g[k], xv, a0); a1 = fmaf(g[k], yv, a1);
            a2 = fmaf(g[k], xv * xv, a2); a3 = fmaf(g[k], yv * yv, a3); a4 = fmaf(g[k], xv * yv, a4);
        }
        s_h[0][py][px] = a0; s_h[1][py][px] = a1; s_h[2][py][px] = a2; s_h[3][py][px] = a3; s_h[4][py][px] = a4;
    }
    __syncthreads();
    const int lx = threadIdx.x % SS_TW, ly = threadIdx.x / SS_TW;
    double val = 0.0;
    if (ox0 + lx < W && oy0 + ly < H) {
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < SS_K; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] = fmaf(g[k], s_h[q][ly + k][lx], m[q]);
        const float mxx = m[0] * m[0], myy = m[1] * m[1], mxy = m[0] * m[1];
        const float sxx = m[2] - mxx, syy = m[3] - myy, sxy = m[4] - mxy;
        const float a1 = 2.0f * mxy + c1, a2 = 2.0f * sxy + c2, b1 = mxx + myy + c1, b2 = sxx + syy + c2;
        val = (double)((a1 * a2) / (b1 * b2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out + b, (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / ((double)C * H * W));
}

hipError_t launch_ssim(const float* rec, const float* clean, double* out, int B, int C, int H, int W, hipStream_t s) {
    if (H <= SS_R || W <= SS_R || B <= 0 || C <= 0 || B > 65535 || C > 65535) return hipErrorInvalidValue;     // reflect padding needs > 5 pixels
    hipError_t e = hipMemsetAsync(out, 0, (size_t)B * sizeof(double), s);
    if (e != hipSuccess) return e;
    const int tiles = ((W + SS_TW - 1) / SS_TW) * ((H + SS_TH - 1) / SS_TH);
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;        // (k1 * data_range)^2, (k2 * data_range)^2
    hipLaunchKernelGGL(ssim_kernel, dim3(tiles, C, B), dim3(256), 0, s, rec, clean, out, C, H, W, c1, c2);
    return hipGetLastError();
}

}  // namespace pf
