// HBM-bound per-pixel kernels of the PnP-Flow iteration and of the degradation
// operators (NCHW fp32 images).  Reference: pnpflow/methods/pnp_flow.py:39-52,109-121,
// pnpflow/degradations.py:15-127, pnpflow/utils.py:283-361, 560-611.
#include <algorithm>
#include <cstdlib>
#include "pf_common.h"

namespace pf {


// ---- mask value of the inpainting family at (b, y, x) -----------------------------------
__device__ __forceinline__ float mask_at(const DegView& d, int b, int y, int x, int H, int W) {
    if (d.kind == DEG_BOX) {
        const int c = H / 2;   // square_mask uses x.shape[2]//2 for both axes (utils.py:331)
        return (y >= c - d.half && y < c + d.half && x >= c - d.half && x < c + d.half) ? 0.f : 1.f;
    }
    if (d.kind == DEG_MASK) return (float)d.mask[((size_t)b * H + y) * W + x];
    return 1.f;
}

// float4 variants (W % 4 == 0): one thread = 4 consecutive pixels of a row; grid (x: float4 index inside the image, y: image), so
// no per-element division survives - the row/column of the quad come from one division per thread, the byte mask is read as one
// 32-bit word.  16 B per lane per access: the HBM-bound kernels of the iteration move 12 B (grad step), 8 B (interpolation) and
// 12-16 B (average) per pixel at the streaming rate.
__device__ __forceinline__ float4 mask4_at(const DegView& d, int b, int y, int x, int H, int W) {
    if (d.kind == DEG_BOX) {
        const int c = H / 2, lo = c - d.half, hi = c + d.half;
        const bool row = y >= lo && y < hi;
        return make_float4((row && x >= lo && x < hi) ? 0.f : 1.f, (row && x + 1 >= lo && x + 1 < hi) ? 0.f : 1.f,
                           (row && x + 2 >= lo && x + 2 < hi) ? 0.f : 1.f, (row && x + 3 >= lo && x + 3 < hi) ? 0.f : 1.f);
    }
    if (d.kind == DEG_MASK) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(d.mask + ((size_t)b * H + y) * W + x);
        return make_float4((float)(m & 0xff), (float)((m >> 8) & 0xff), (float)((m >> 16) & 0xff), (float)(m >> 24));
    }
    return make_float4(1.f, 1.f, 1.f, 1.f);
}

__global__ __launch_bounds__(256) void mask_apply4_kernel(DegView d, const float4* __restrict__ x, float4* __restrict__ y, int n4, int H, int W4) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int row = i / W4, px = (i - row * W4) * 4, py = row % H;
        const float4 m = mask4_at(d, b, py, px, H, W4 * 4), v = x[(size_t)b * n4 + i];
        y[(size_t)b * n4 + i] = make_float4(m.x * v.x, m.y * v.y, m.z * v.z, m.w * v.w);
    }
}

__global__ __launch_bounds__(256) void grad_step_mask4_kernel(DegView d, const float4* __restrict__ x, const float4* __restrict__ y,
                                                              const float* __restrict__ coef, float4* __restrict__ z, int n4, int H, int W4, int laplace) {
    const int b = blockIdx.y;
    const float cf = coef[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int row = i / W4, px = (i - row * W4) * 4, py = row % H;
        const float4 m = mask4_at(d, b, py, px, H, W4 * 4), xv = x[(size_t)b * n4 + i], yv = y[(size_t)b * n4 + i];
        float4 r = make_float4(m.x * xv.x - yv.x, m.y * xv.y - yv.y, m.z * xv.z - yv.z, m.w * xv.w - yv.w);
        if (laplace) r = make_float4(r.x > 0.f ? 1.f : -1.f, r.y > 0.f ? 1.f : -1.f, r.z > 0.f ? 1.f : -1.f, r.w > 0.f ? 1.f : -1.f);
        z[(size_t)b * n4 + i] = make_float4(xv.x - cf * (m.x * r.x), xv.y - cf * (m.y * r.y), xv.z - cf * (m.z * r.z), xv.w - cf * (m.w * r.w));
    }
}

// z = x - coef * zerofill(decimate(x) - y): a full-resolution quad holds at most 4/sf samples of the low-resolution image
__global__ __launch_bounds__(256) void grad_step_sr4_kernel(const float4* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef,
                                                            float4* __restrict__ z, int n4, int C, int H, int W4, int sf, int laplace) {
    const int b = blockIdx.y;
    const int W = W4 * 4, Hy = H / sf, Wy = W / sf;
    const float cf = coef[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int row = i / W4, px = (i - row * W4) * 4, py = row % H, pl = row / H;
        const float4 xv = x[(size_t)b * n4 + i];
        float v[4] = {xv.x, xv.y, xv.z, xv.w};
        if (py % sf == 0) {
            const float* yr = y + (((size_t)b * C + pl) * Hy + py / sf) * Wy;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((px + j) % sf == 0) {
                    float g = v[j] - yr[(px + j) / sf];
                    if (laplace) g = g > 0.f ? 1.f : -1.f;
                    v[j] = v[j] - cf * g;
                }
        }
        z[(size_t)b * n4 + i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// y = M*x (H and H_adj of the mask family, identity for denoising)
__global__ __launch_bounds__(256) void mask_apply_kernel(DegView d, const float* x, float* y, int C, int H, int W) {
    const int b = blockIdx.y;
    const int n = C * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H;
        y[(size_t)b * n + i] = mask_at(d, b, py, px, H, W) * x[(size_t)b * n + i];
    }
}

// Superresolution H: y[.., i, j] = x[.., sf*i, sf*j]   (utils.py:302-310)
__global__ __launch_bounds__(256) void decimate_kernel(const float* x, float* y, int planes, int H, int W, int sf) {
    const int Hy = H / sf, Wy = W / sf;
    const size_t n = (size_t)planes * Hy * Wy;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int j = i % Wy, r = (i / Wy) % Hy;
        const size_t pl = i / ((size_t)Wy * Hy);
        y[i] = x[(pl * H + (size_t)r * sf) * W + (size_t)j * sf];
    }
}

// Superresolution H_adj: zero-fill upsample (utils.py:283-299)
__global__ __launch_bounds__(256) void zerofill_kernel(const float* y, float* x, int planes, int H, int W, int sf) {
    const int Wy = W / sf, Hy = H / sf;
    const size_t n = (size_t)planes * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H;
        const size_t pl = i / ((size_t)W * H);
        float v = 0.f;
        if (py % sf == 0 && px % sf == 0) v = y[(pl * Hy + py / sf) * Wy + px / sf];
        x[i] = v;
    }
}

// One pass of the separable circular Gaussian (degradations.py:55-89: the FFT product
// with the rolled 61x61 filter is a circular convolution with outer(g,g)).
//   dir 0: along x, dir 1: along y;  sign +1: convolution (H), -1: correlation (H_adj)
//   mode 0: out = val;  1: out = val - aux[i];  2: out = aux[i] - coef[b]*val;  3: out = sgn(val - aux[i]) in {-1,+1}
//   (sgn = 2*heaviside(.,0)-1, the Laplace data-fit gradient of pnp_flow.py:42-43)
// LDS-tiled (round 1 re-read 61 taps per output from global memory with a modulo per tap): a workgroup stages the lines it
// filters once - dir 0: BL_ROWS whole image rows, dir 1: a strip of BL_COLS columns over the full height - with the circular
// wrap resolved at staging time (the line is stored with `r` wrapped samples on either side), so the tap loop is
// ntaps fused multiply-adds on consecutive LDS words.  Algorithmic HBM traffic: one read + one write of the plane per pass
// (+ aux in modes 1-3); the summation order over the taps is the one of round 1 (k ascending), results unchanged.
constexpr int BL_MAXW = 2048 + 128;       // longest staged line (image side <= 2048, ntaps <= 127)

__global__ __launch_bounds__(256) void blur_rows_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps,
                                                       int ntaps, int rows, int W, int sign, int mode,
                                                       const float* __restrict__ aux, const float* __restrict__ coef, int rows_per_image) {
    extern __shared__ float s_line[];            // [rows_here][W + 2r] then the taps
    const int r = ntaps / 2, LW = W + 2 * r + 4;           // 4 extra samples: the 4-output sliding window reads up to line[W + r + 2]
    const int rows_here = min((int)blockDim.y, rows - (int)blockIdx.x * (int)blockDim.y);
    float* s_g = s_line + blockDim.y * LW;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y;
    for (int k = tid; k < ntaps; k += nthr) s_g[k] = taps[k];
    const size_t row0 = (size_t)blockIdx.x * blockDim.y;
    for (int i = tid; i < rows_here * LW; i += nthr) {
        const int rr = i / LW, j = i % LW;
        int xx = (j - r) % W; xx += xx < 0 ? W : 0;
        s_line[rr * LW + j] = in[(row0 + rr) * W + xx];
    }
    __syncthreads();
    if ((int)threadIdx.y >= rows_here) return;
    const size_t grow = row0 + threadIdx.y;
    const int b = (int)(grow / rows_per_image);
    const float* line = s_line + threadIdx.y * LW + r;       // line[x] = in[row][x], valid for x in [-r, W + r)
    // each thread filters 4 consecutive outputs with a sliding window: one LDS read feeds 4 multiply-adds (the tap order per
    // output is unchanged: k ascending)
    for (int px = threadIdx.x * 4; px < W; px += blockDim.x * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (sign > 0) {
            // out[px + j] = sum_k g[k] * line[px + j - (k - r)]: window w[j] = line[px + j + r - k]
            float w1 = line[px + 1 + r], w2 = line[px + 2 + r], w3 = line[px + 3 + r];
            for (int k = 0; k < ntaps; ++k) {
                const float w0 = line[px + r - k], gk = s_g[k];
                acc[0] = fmaf(gk, w0, acc[0]); acc[1] = fmaf(gk, w1, acc[1]); acc[2] = fmaf(gk, w2, acc[2]); acc[3] = fmaf(gk, w3, acc[3]);
                w3 = w2; w2 = w1; w1 = w0;
            }
        } else {
            // out[px + j] = sum_k g[k] * line[px + j + (k - r)]: window w[j] = line[px + j - r + k]
            float w0 = line[px - r], w1 = line[px + 1 - r], w2 = line[px + 2 - r];
            for (int k = 0; k < ntaps; ++k) {
                const float w3 = line[px + 3 - r + k], gk = s_g[k];
                acc[0] = fmaf(gk, w0, acc[0]); acc[1] = fmaf(gk, w1, acc[1]); acc[2] = fmaf(gk, w2, acc[2]); acc[3] = fmaf(gk, w3, acc[3]);
                w0 = w1; w1 = w2; w2 = w3;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (px + j >= W) break;
            float a = acc[j];
            const size_t o = grow * W + px + j;
            if (mode == 1) a = a - aux[o];
            else if (mode == 2) a = aux[o] - coef[b] * a;
            else if (mode == 3) a = (a - aux[o]) > 0.f ? 1.f : -1.f;
            out[o] = a;
        }
    }
}

constexpr int BL_COLS = 32;
__global__ __launch_bounds__(256) void blur_cols_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps,
                                                       int ntaps, int H, int W, int sign, int mode,
                                                       const float* __restrict__ aux, const float* __restrict__ coef, int planes_per_image) {
    extern __shared__ float s_col[];             // [H + 2r][BL_COLS] then the taps
    const int r = ntaps / 2, LH = H + 2 * r + 4;          // 4 extra rows for the 4-output sliding window
    float* s_g = s_col + LH * BL_COLS;
    const int tid = threadIdx.x;
    for (int k = tid; k < ntaps; k += 256) s_g[k] = taps[k];
    const int plane = blockIdx.y, x0 = blockIdx.x * BL_COLS;
    const float* src = in + (size_t)plane * H * W;
    for (int i = tid; i < LH * BL_COLS; i += 256) {
        const int j = i / BL_COLS, cx = i % BL_COLS;
        int yy = (j - r) % H; yy += yy < 0 ? H : 0;
        s_col[i] = (x0 + cx < W) ? src[(size_t)yy * W + x0 + cx] : 0.f;
    }
    __syncthreads();
    const int b = plane / planes_per_image;
    const int cx = tid % BL_COLS;
    if (x0 + cx >= W) return;
    // 4 consecutive rows per thread, sliding window down the staged column (one LDS read per tap feeds 4 multiply-adds)
    for (int py = (tid / BL_COLS) * 4; py < H; py += (256 / BL_COLS) * 4) {
        const float* col = s_col + (py + r) * BL_COLS + cx;       // col[d * BL_COLS] = in[py + d][x]
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (sign > 0) {
            float w1 = col[(1 + r) * BL_COLS], w2 = col[(2 + r) * BL_COLS], w3 = col[(3 + r) * BL_COLS];
            for (int k = 0; k < ntaps; ++k) {
                const float w0 = col[(r - k) * BL_COLS], gk = s_g[k];
                acc[0] = fmaf(gk, w0, acc[0]); acc[1] = fmaf(gk, w1, acc[1]); acc[2] = fmaf(gk, w2, acc[2]); acc[3] = fmaf(gk, w3, acc[3]);
                w3 = w2; w2 = w1; w1 = w0;
            }
        } else {
            float w0 = col[(-r) * BL_COLS], w1 = col[(1 - r) * BL_COLS], w2 = col[(2 - r) * BL_COLS];
            for (int k = 0; k < ntaps; ++k) {
                const float w3 = col[(3 - r + k) * BL_COLS], gk = s_g[k];
                acc[0] = fmaf(gk, w0, acc[0]); acc[1] = fmaf(gk, w1, acc[1]); acc[2] = fmaf(gk, w2, acc[2]); acc[3] = fmaf(gk, w3, acc[3]);
                w0 = w1; w1 = w2; w2 = w3;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (py + j >= H) break;
            float a = acc[j];
            const size_t o = ((size_t)plane * H + py + j) * W + x0 + cx;
            if (mode == 1) a = a - aux[o];
            else if (mode == 2) a = aux[o] - coef[b] * a;
            else if (mode == 3) a = (a - aux[o]) > 0.f ? 1.f : -1.f;
            out[o] = a;
        }
    }
}

static inline bool aligned16(const void* a, const void* b, const void* c = nullptr, const void* d = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

static inline dim3 grid_for(int n_per_image, int B) {
    int gx = (n_per_image + 255) / 256;
    if (gx > 1024) gx = 1024;
    return dim3(gx, B);
}

// One pass per application of the separable circular filter (round 3): a workgroup stages the (TS + 2r)^2 neighbourhood of a TS x TS output
// tile once (wrap resolved while staging), filters it along x into a second LDS array and along y into the outputs.  Against the
// row pass + column pass pair this halves the HBM traffic of an application (no intermediate tensor) and the launches; the halo
// re-reads (2.1x at r = 7, TS = 32) are L2 hits.  Odd tap counts with 2r < min(H, W) and r <= 24 only (the Gaussian's visible taps: 15
// at sigma 1, 43 at sigma 3); anything else keeps the two-pass path.  mode / sign as above; the taps are flipped once for the
// convolution so that both directions are the correlation  out[i] = sum_k g'[k] in[i - r + k].
template <int TS>
__global__ __launch_bounds__(256) void blur2d_fused_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ taps,
                                                          int ntaps, int H, int W, int sign, int mode,
                                                          const float* __restrict__ aux, const float* __restrict__ coef, int planes_per_image) {
    extern __shared__ float s_f[];
    const int r = ntaps / 2, PW = TS + 2 * r, PWP = PW + 1;        // +1: row pitch off the bank period
    float* s_in = s_f;                         // [PW][PWP]
    float* s_h = s_in + PW * PWP + 4;          // [PW + 3][TS + 1]   (filtered along x; the sliding windows read up to 3 elements / rows past the last tap)
    float* s_g = s_h + (PW + 3) * (TS + 1);
    const int tid = threadIdx.x;
    for (int k = tid; k < ntaps; k += 256) s_g[k] = taps[sign > 0 ? ntaps - 1 - k : k];
    const int plane = blockIdx.z, x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const float* src = in + (size_t)plane * H * W;
    for (int i = tid; i < PW * PW; i += 256) {
        const int py = i / PW, px = i - py * PW;
        int gy = y0 - r + py, gx = x0 - r + px;
        gy += gy < 0 ? H : 0; gy -= gy >= H ? H : 0; gy -= gy >= H ? H : 0;      // (a tile past the image edge wraps twice at most: TS + r < 2H)
        gx += gx < 0 ? W : 0; gx -= gx >= W ? W : 0; gx -= gx >= W ? W : 0;
        s_in[py * PWP + px] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    // four consecutive outputs per thread with a sliding window: one LDS read per tap feeds four multiply-adds
    for (int i = tid; i < PW * (TS / 4); i += 256) {
        const int py = i / (TS / 4), x = (i - py * (TS / 4)) * 4;
        const float* row = s_in + py * PWP + x;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float w0 = row[0], w1 = row[1], w2 = row[2];
        for (int k = 0; k < ntaps; ++k) {
            const float w3 = row[k + 3], gk = s_g[k];
            a0 = fmaf(gk, w0, a0); a1 = fmaf(gk, w1, a1); a2 = fmaf(gk, w2, a2); a3 = fmaf(gk, w3, a3);
            w0 = w1; w1 = w2; w2 = w3;
        }
        float* hp = s_h + py * (TS + 1) + x;
        hp[0] = a0; hp[1] = a1; hp[2] = a2; hp[3] = a3;
    }
    __syncthreads();
    const int b = plane / planes_per_image;
    for (int i = tid; i < (TS / 4) * TS; i += 256) {
        const int yq = i / TS, x = i - yq * TS, y = yq * 4;
        const float* col = s_h + y * (TS + 1) + x;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        float w0 = col[0], w1 = col[TS + 1], w2 = col[2 * (TS + 1)];
        for (int k = 0; k < ntaps; ++k) {
            const float w3 = col[(k + 3) * (TS + 1)], gk = s_g[k];
            a[0] = fmaf(gk, w0, a[0]); a[1] = fmaf(gk, w1, a[1]); a[2] = fmaf(gk, w2, a[2]); a[3] = fmaf(gk, w3, a[3]);
            w0 = w1; w1 = w2; w2 = w3;
        }
        if (x0 + x >= W) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (y0 + y + j >= H) break;
            float v = a[j];
            const size_t o = ((size_t)plane * H + y0 + y + j) * W + x0 + x;
            if (mode == 1) v = v - aux[o];
            else if (mode == 2) v = aux[o] - coef[b] * v;
            else if (mode == 3) v = (v - aux[o]) > 0.f ? 1.f : -1.f;
            out[o] = v;
        }
    }
}

static hipError_t blur2(const DegView& d, const float* in, float* tmp, float* out, int B, int C, int H, int W, int sign,
                        int mode, const float* aux, const float* coef, hipStream_t s) {
    if (W + d.ntaps > BL_MAXW || H + d.ntaps > BL_MAXW) return hipErrorInvalidValue;
    static const bool fused_env = !(getenv("PNPFLOW_HIP_BLUR_FUSED") && atoi(getenv("PNPFLOW_HIP_BLUR_FUSED")) == 0);
    if (fused_env && (d.ntaps & 1) && d.ntaps / 2 <= 24 && d.ntaps < H && d.ntaps < W && in != out &&
        (d.ntaps / 2 <= 8 ? 32 : 64) + d.ntaps / 2 <= 2 * std::min(H, W)) {      // (the staging loop resolves at most two wraps)
        const int r = d.ntaps / 2;
        if (r <= 8) {
            constexpr int TS = 32;
            const int PW = TS + 2 * r;
            const size_t lds = ((size_t)PW * (PW + 1) + 4 + (size_t)(PW + 3) * (TS + 1) + 128) * sizeof(float);
            hipLaunchKernelGGL(blur2d_fused_kernel<TS>, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B * C), dim3(256), lds, s, in, out, d.taps, d.ntaps, H, W, sign, mode,
                               aux, coef, C);
        } else {
            constexpr int TS = 64;
            const int PW = TS + 2 * r;
            const size_t lds = ((size_t)PW * (PW + 1) + 4 + (size_t)(PW + 3) * (TS + 1) + 128) * sizeof(float);
            static unsigned long long attr64 = 0ull;
            { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(blur2d_fused_kernel<TS>), attr64, 160 * 1024); if (e != hipSuccess) return e; }
            hipLaunchKernelGGL(blur2d_fused_kernel<TS>, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B * C), dim3(256), lds, s, in, out, d.taps, d.ntaps, H, W, sign, mode,
                               aux, coef, C);
        }
        return hipGetLastError();
    }
    const int r = d.ntaps / 2;
    // rows: 64-wide row groups, 256 threads per workgroup
    const int tx = 64, ty = 4, rows = B * C * H;
    const size_t lds_r = ((size_t)ty * (W + 2 * r + 4) + 128) * sizeof(float);
    hipLaunchKernelGGL(blur_rows_kernel, dim3((rows + ty - 1) / ty), dim3(tx, ty), lds_r, s, in, tmp, d.taps, d.ntaps, rows, W, sign, 0,
                       (const float*)nullptr, (const float*)nullptr, C * H);
    const size_t lds_c = ((size_t)(H + 2 * r + 4) * BL_COLS + 128) * sizeof(float);
    static unsigned long long attr_set = 0ull;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(blur_cols_kernel), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    if (lds_c > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(blur_cols_kernel, dim3((W + BL_COLS - 1) / BL_COLS, B * C), dim3(256), lds_c, s, (const float*)tmp, out, d.taps, d.ntaps,
                       H, W, sign, mode, aux, coef, C);
    return hipGetLastError();
}

hipError_t launch_deg_H(const DegView& d, const float* x, float* y, int B, int C, int H, int W, float* scratch, hipStream_t s) {
    switch (d.kind) {
        case DEG_DENOISE: case DEG_BOX: case DEG_MASK:
            if (W % 4 == 0 && aligned16(x, y) && ((uintptr_t)d.mask & 3) == 0) hipLaunchKernelGGL(mask_apply4_kernel, grid_for(C * H * W / 4, B), dim3(256), 0, s, d, (const float4*)x, (float4*)y, C * H * W / 4, H, W / 4);
            else hipLaunchKernelGGL(mask_apply_kernel, grid_for(C * H * W, B), dim3(256), 0, s, d, x, y, C, H, W);
            return hipGetLastError();
        case DEG_SR: {
            if (d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            const size_t n = (size_t)B * C * (H / d.sf) * (W / d.sf);
            hipLaunchKernelGGL(decimate_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, x, y, B * C, H, W, d.sf);
            return hipGetLastError();
        }
        case DEG_BLUR:
            if (!scratch || d.ntaps < 1 || d.ntaps > 127) return hipErrorInvalidValue;
            return blur2(d, x, scratch, y, B, C, H, W, +1, 0, nullptr, nullptr, s);
        case DEG_SR_FILTER: {      // y = decimate(filter (*) x)
            if (!scratch || d.ntaps < 1 || d.ntaps > 127 || d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            float* s0 = scratch; float* s1 = scratch + (size_t)B * C * H * W;
            hipError_t e = blur2(d, x, s0, s1, B, C, H, W, +1, 0, nullptr, nullptr, s);
            if (e != hipSuccess) return e;
            const size_t n = (size_t)B * C * (H / d.sf) * (W / d.sf);
            hipLaunchKernelGGL(decimate_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, (const float*)s1, y, B * C, H, W, d.sf);
            return hipGetLastError();
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_deg_Hadj(const DegView& d, const float* y, float* x, int B, int C, int H, int W, float* scratch, hipStream_t s) {
    switch (d.kind) {
        case DEG_DENOISE: case DEG_BOX: case DEG_MASK:
            if (W % 4 == 0 && aligned16(x, y) && ((uintptr_t)d.mask & 3) == 0) hipLaunchKernelGGL(mask_apply4_kernel, grid_for(C * H * W / 4, B), dim3(256), 0, s, d, (const float4*)y, (float4*)x, C * H * W / 4, H, W / 4);
            else hipLaunchKernelGGL(mask_apply_kernel, grid_for(C * H * W, B), dim3(256), 0, s, d, y, x, C, H, W);
            return hipGetLastError();
        case DEG_SR: {
            if (d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            const size_t n = (size_t)B * C * H * W;
            hipLaunchKernelGGL(zerofill_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, y, x, B * C, H, W, d.sf);
            return hipGetLastError();
        }
        case DEG_BLUR:
            if (!scratch || d.ntaps < 1 || d.ntaps > 127) return hipErrorInvalidValue;
            return blur2(d, y, scratch, x, B, C, H, W, -1, 0, nullptr, nullptr, s);
        case DEG_SR_FILTER: {      // x = filter^T (*) zerofill(y)
            if (!scratch || d.ntaps < 1 || d.ntaps > 127 || d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            float* s0 = scratch; float* s1 = scratch + (size_t)B * C * H * W;
            const size_t n = (size_t)B * C * H * W;
            hipLaunchKernelGGL(zerofill_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, y, s0, B * C, H, W, d.sf);
            return blur2(d, s0, s1, x, B, C, H, W, -1, 0, nullptr, nullptr, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- fused data-fidelity gradient step: z = x - coef[b] * H_adj(H x - y) -------------------
// (pnp_flow.py:39-41 with lr = sigma^2*lr_pnp, :109-112; sigma^2 is folded into coef.)
__global__ __launch_bounds__(256) void grad_step_mask_kernel(DegView d, const float* x, const float* y, const float* coef,
                                                             float* z, int C, int H, int W, int laplace) {
    const int b = blockIdx.y;
    const int n = C * H * W;
    const float cf = coef[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H;
        const float m = mask_at(d, b, py, px, H, W);
        const size_t o = (size_t)b * n + i;
        const float xv = x[o];
        float r = m * xv - y[o];
        if (laplace) r = r > 0.f ? 1.f : -1.f;          // 2*heaviside(Hx-y, 0) - 1
        z[o] = xv - cf * (m * r);
    }
}

__global__ __launch_bounds__(256) void grad_step_sr_kernel(const float* x, const float* y, const float* coef, float* z,
                                                           int C, int H, int W, int sf, int laplace) {
    const int b = blockIdx.y;
    const int n = C * H * W, Hy = H / sf, Wy = W / sf;
    const float cf = coef[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H, pl = i / (W * H);
        const size_t o = (size_t)b * n + i;
        const float xv = x[o];
        float g = 0.f;
        if (py % sf == 0 && px % sf == 0) {
            g = xv - y[(((size_t)b * C + pl) * Hy + py / sf) * Wy + px / sf];
            if (laplace) g = g > 0.f ? 1.f : -1.f;
        }
        z[o] = xv - cf * g;
    }
}

// out (full resolution) = zerofill( decimate(hx) - y )   (or its sign: laplace)
__global__ __launch_bounds__(256) void sr_residual_kernel(const float* hx, const float* y, float* out, int C, int H, int W, int sf, int laplace) {
    const int b = blockIdx.y;
    const int n = C * H * W, Hy = H / sf, Wy = W / sf;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H, pl = i / (W * H);
        const size_t o = (size_t)b * n + i;
        float g = 0.f;
        if (py % sf == 0 && px % sf == 0) {
            g = hx[o] - y[(((size_t)b * C + pl) * Hy + py / sf) * Wy + px / sf];
            if (laplace) g = g > 0.f ? 1.f : -1.f;
        }
        out[o] = g;
    }
}

hipError_t launch_grad_step(const DegView& d, const float* x, const float* y, const float* coef, float* z,
                            int B, int C, int H, int W, float* scratch, int laplace, hipStream_t s) {
    dim3 g = grid_for(C * H * W, B);
    switch (d.kind) {
        case DEG_DENOISE: case DEG_BOX: case DEG_MASK:
            if (W % 4 == 0 && aligned16(x, y, z) && (d.kind != DEG_MASK || ((uintptr_t)d.mask & 3) == 0))
                hipLaunchKernelGGL(grad_step_mask4_kernel, grid_for(C * H * W / 4, B), dim3(256), 0, s, d, (const float4*)x, (const float4*)y, coef, (float4*)z,
                                   C * H * W / 4, H, W / 4, laplace);
            else hipLaunchKernelGGL(grad_step_mask_kernel, g, dim3(256), 0, s, d, x, y, coef, z, C, H, W, laplace);
            return hipGetLastError();
        case DEG_SR:
            if (d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            if (W % 4 == 0 && aligned16(x, z))
                hipLaunchKernelGGL(grad_step_sr4_kernel, grid_for(C * H * W / 4, B), dim3(256), 0, s, (const float4*)x, y, coef, (float4*)z, C * H * W / 4, C, H, W / 4,
                                   d.sf, laplace);
            else hipLaunchKernelGGL(grad_step_sr_kernel, g, dim3(256), 0, s, x, y, coef, z, C, H, W, d.sf, laplace);
            return hipGetLastError();
        case DEG_BLUR: {
            if (!scratch || d.ntaps < 1 || d.ntaps > 127) return hipErrorInvalidValue;
            float* s0 = scratch;
            float* s1 = scratch + (size_t)B * C * H * W;
            hipError_t e = blur2(d, x, s0, s1, B, C, H, W, +1, laplace ? 3 : 1, y, nullptr, s);     // s1 = Hx - y  (or its sign)
            if (e != hipSuccess) return e;
            return blur2(d, s1, s0, z, B, C, H, W, -1, 2, x, coef, s);                 // z = x - coef*H_adj(s1)
        }
        case DEG_SR_FILTER: {
            if (!scratch || d.ntaps < 1 || d.ntaps > 127 || d.sf <= 0 || H % d.sf || W % d.sf) return hipErrorInvalidValue;
            float* s0 = scratch;
            float* s1 = scratch + (size_t)B * C * H * W;
            hipError_t e = blur2(d, x, s0, s1, B, C, H, W, +1, 0, nullptr, nullptr, s);             // s1 = filter (*) x
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(sr_residual_kernel, g, dim3(256), 0, s, (const float*)s1, y, s0, C, H, W, d.sf, laplace);
            return blur2(d, s0, s1, z, B, C, H, W, -1, 2, x, coef, s);                 // z = x - coef*filter^T (*) s0
        }
    }
    return hipErrorInvalidValue;
}

// ---- engine RNG: Philox4x32-10 + Box-Muller (restated in oracle/pnpflow_oracle.py) ----------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                               uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float4 normal4(uint64_t q, uint64_t seed, uint64_t stream) {
    uint32_t r[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float k = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = ((float)r[0] + 0.5f) * k, u1 = ((float)r[1] + 0.5f) * k;
    const float u2 = ((float)r[2] + 0.5f) * k, u3 = ((float)r[3] + 0.5f) * k;
    const float rad0 = sqrtf(-2.0f * logf(u0)), rad1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    return make_float4(rad0 * c0, rad0 * s0, rad1 * c1, rad1 * s1);
}

// normals number [first, first+4) of stream `stream` (normal number e = component e%4 of Philox counter e/4): a shard of a
// multi-GPU run passes first = elem_offset + 4*q, so that it draws exactly the numbers the single-device run at the global
// batch size draws for the same images
__device__ __forceinline__ void normals_at(uint64_t first, uint64_t seed, uint64_t stream, float e[4]) {
    const uint64_t q0 = first >> 2; const int r = (int)(first & 3);
    const float4 a = normal4(q0, seed, stream);
    if (r == 0) { e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; return; }
    const float4 b2 = normal4(q0 + 1, seed, stream);
    const float v[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = v[r + j];
}

__global__ __launch_bounds__(256) void fill_normal_kernel(float* out, int64_t n, uint64_t seed, uint64_t stream, uint64_t elem_offset) {
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        float zz[4];
        normals_at(elem_offset + (uint64_t)q * 4, seed, stream, zz);
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = zz[j];
    }
}

hipError_t launch_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, hipStream_t s) {
    const int64_t nq = (n + 3) / 4;
    hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)std::min<int64_t>((nq + 255) / 256, 4096)), dim3(256), 0, s, out, n, seed, stream_id, elem_offset);
    return hipGetLastError();
}

// z_tilde = t*z + (1-t)*eps   (pnp_flow.py:47-48).  The batch is one flat noise stream:
// element e of the [B*n] tensor uses normal number e of stream `stream_id`.
__global__ __launch_bounds__(256) void interpolate_kernel(const float* z, const float* t, const float* noise, uint64_t seed,
                                                          uint64_t stream, float* zt, int n, int64_t total) {
    const int64_t nq = (total + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        float e[4];
        if (noise != nullptr) {
            for (int j = 0; j < 4; ++j) e[j] = (q * 4 + j < total) ? noise[q * 4 + j] : 0.f;
        } else {
            normals_at((uint64_t)q * 4, seed, stream, e);
        }
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < total) {
                const float tb = t[i / n];
                zt[i] = tb * z[i] + e[j] * (1.0f - tb);
            }
        }
    }
}

hipError_t launch_interpolate(const float* z, const float* t, const float* noise, uint64_t seed, uint64_t stream_id,
                              float* zt, int B, int n, hipStream_t s) {
    const int64_t total = (int64_t)B * n, nq = (total + 3) / 4;
    hipLaunchKernelGGL(interpolate_kernel, dim3((unsigned)std::min<int64_t>((nq + 255) / 256, 4096)), dim3(256), 0, s,
                       z, t, noise, seed, stream_id, zt, n, total);
    return hipGetLastError();
}

// Same, with the noise stream / injected-noise slice of (iteration, sample) resolved from a
// device-side iteration counter, so one captured hipGraph serves every outer iteration.
__global__ __launch_bounds__(256) void interp_iter_kernel(const float* z, const float* t, const float* noise, const unsigned long long* rng,
                                                          const int* iter, int num_samples, int sample,
                                                          float* zt, int n, int64_t total) {
    const uint64_t seed = rng[0], stream_base = rng[1], elem_offset = rng[2];
    const int64_t slot = (int64_t)(*iter) * num_samples + sample;
    const uint64_t stream = stream_base + (uint64_t)slot;
    const float* nz = noise != nullptr ? noise + slot * total : nullptr;
    if ((n & 3) == 0 && gridDim.y > 1) {
        // one image per blockIdx.y, 4 consecutive elements per thread: t[b] is a scalar, loads and stores are 16 B per lane
        const int b = blockIdx.y, n4 = n >> 2;
        const float tb = t[b], ob = 1.0f - tb;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
            const int64_t i = (int64_t)b * n + (int64_t)q * 4;
            float e[4];
            if (nz != nullptr) { const float4 ev = *reinterpret_cast<const float4*>(nz + i); e[0] = ev.x; e[1] = ev.y; e[2] = ev.z; e[3] = ev.w; }
            else normals_at(elem_offset + (uint64_t)i, seed, stream, e);
            const float4 zv = *reinterpret_cast<const float4*>(z + i);
            *reinterpret_cast<float4*>(zt + i) = make_float4(tb * zv.x + e[0] * ob, tb * zv.y + e[1] * ob, tb * zv.z + e[2] * ob, tb * zv.w + e[3] * ob);
        }
        return;
    }
    const int64_t nq = (total + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        float e[4];
        if (nz != nullptr) {
            for (int j = 0; j < 4; ++j) e[j] = (q * 4 + j < total) ? nz[q * 4 + j] : 0.f;
        } else {
            normals_at(elem_offset + (uint64_t)q * 4, seed, stream, e);
        }
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < total) {
                const float tb = t[i / n];
                zt[i] = tb * z[i] + e[j] * (1.0f - tb);
            }
        }
    }
}

hipError_t launch_interp_iter(const float* z, const float* t, const float* noise, const unsigned long long* rng,
                              const int* iter, int num_samples, int sample, float* zt, int B, int n, hipStream_t s) {
    const int64_t total = (int64_t)B * n, nq = (total + 3) / 4;
    const bool vec = (n & 3) == 0 && B > 1 && B <= 65535 && aligned16(z, zt, noise);
    const dim3 grid = vec ? grid_for(n / 4, B) : dim3((unsigned)std::min<int64_t>((nq + 255) / 256, 4096));
    hipLaunchKernelGGL(interp_iter_kernel, grid, dim3(256), 0, s, z, t, noise, rng, iter, num_samples, sample, zt, n, total);
    return hipGetLastError();
}

// acc (=|+=) z_tilde + (1-t)*v ; last sample: acc /= num_samples  (pnp_flow.py:50-52,114-121)
__global__ __launch_bounds__(256) void denoise_accum_kernel(float* acc, const float* zt, const float* v, const float* t,
                                                            int mode, float ns, int n, int64_t total) {
    if ((n & 3) == 0 && gridDim.y > 1) {
        const int b = blockIdx.y, n4 = n >> 2;
        const float ob = 1.0f - t[b];
        for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
            const int64_t i = (int64_t)b * n + (int64_t)q * 4;
            const float4 a = *reinterpret_cast<const float4*>(zt + i), w = *reinterpret_cast<const float4*>(v + i);
            float4 r = make_float4(a.x + ob * w.x, a.y + ob * w.y, a.z + ob * w.z, a.w + ob * w.w);
            if (!(mode & 1)) { const float4 o = *reinterpret_cast<const float4*>(acc + i); r = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w); }
            if (mode & 2) r = make_float4(r.x / ns, r.y / ns, r.z / ns, r.w / ns);
            *reinterpret_cast<float4*>(acc + i) = r;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float tb = t[i / n];
        float val = zt[i] + (1.0f - tb) * v[i];
        if (!(mode & 1)) val = acc[i] + val;
        if (mode & 2) val = val / ns;
        acc[i] = val;
    }
}

hipError_t launch_denoise_accum(float* acc, const float* zt, const float* v, const float* t, int mode, float ns,
                                int B, int n, hipStream_t s) {
    const int64_t total = (int64_t)B * n;
    const bool vec = (n & 3) == 0 && B > 1 && B <= 65535 && aligned16(acc, zt, v);
    const dim3 grid = vec ? grid_for(n / 4, B) : dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(denoise_accum_kernel, grid, dim3(256), 0, s, acc, zt, v, t, mode, ns, n, total);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fill_kernel(float* out, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = v;
}

hipError_t launch_fill(float* out, int64_t n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, out, n, v);
    return hipGetLastError();
}

// ---- OT-ODE per-pixel steps (pnpflow/methods/ot_ode.py:72-130, 141-147) -------------------------
// vec = H_adj( (r_t^2 H H^T + sigma^2)^-1 (y - H(x + (1-t) v_t)) ) in closed form for the operators
// whose H H^T is diagonal: identity, masks (m in {0,1}), decimation (diag(D D^T) = 1).
__global__ __launch_bounds__(256) void ot_ode_vec_kernel(DegView d, const float* x, const float* vt, const float* y, const float* omt,
                                                         const float* rt2, float sigma2, float* vec, int C, int H, int W) {
    const int b = blockIdx.y;
    const int n = C * H * W;
    const float o = omt[b], r2 = rt2[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int px = i % W, py = (i / W) % H, pl = i / (W * H);
        const size_t idx = (size_t)b * n + i;
        const float x1 = x[idx] + o * vt[idx];                         // x1_hat (ot_ode.py:74)
        float v;
        if (d.kind == DEG_SR) {
            v = 0.f;
            if (py % d.sf == 0 && px % d.sf == 0) {
                const int Hy = H / d.sf, Wy = W / d.sf;
                const float dd = y[(((size_t)b * C + pl) * Hy + py / d.sf) * Wy + px / d.sf] - x1;
                v = (1.0f / (r2 + sigma2)) * dd;                       // ot_ode.py:95-106 (rt2 carries the reference's quirk)
            }
        } else if (d.kind == DEG_DENOISE) {
            v = (y[idx] - x1) / (r2 + sigma2);                         // ot_ode.py:89-93
        } else {
            const float m = mask_at(d, b, py, px, H, W);
            v = m * ((1.0f / (m * r2 + sigma2)) * (y[idx] - m * x1));  // ot_ode.py:81-87, then H_adj (:130)
        }
        vec[idx] = v;
    }
}

hipError_t launch_ot_ode_vec(const DegView& d, const float* x, const float* vt, const float* y, const float* one_minus_t, const float* rt2,
                             float sigma2, float* vec, int B, int C, int H, int W, hipStream_t s) {
    if (d.kind == DEG_BLUR) return hipErrorInvalidValue;   // Fourier-domain solve: launch_ot_ode_vec_blur (fft2.hip)
    if (d.kind == DEG_SR && (d.sf <= 0 || H % d.sf || W % d.sf)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ot_ode_vec_kernel, grid_for(C * H * W, B), dim3(256), 0, s, d, x, vt, y, one_minus_t, rt2, sigma2, vec, C, H, W);
    return hipGetLastError();
}

// x += delta * (vt + coef[b] * (vec + (1-t[b]) * g)),  coef = ((1-t)/t) * gamma     (ot_ode.py:141-147)
__global__ __launch_bounds__(256) void ot_ode_update_kernel(float* x, const float* vt, const float* vec, const float* g, const float* omt,
                                                            const float* coef, float delta, int n, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / n);
        const float gg = vec[i] + omt[b] * g[i];
        x[i] = x[i] + delta * (vt[i] + coef[b] * gg);
    }
}

hipError_t launch_ot_ode_update(float* x, const float* vt, const float* vec, const float* g, const float* one_minus_t, const float* coef,
                                float delta, int B, int n, hipStream_t s) {
    const int64_t total = (int64_t)B * n;
    hipLaunchKernelGGL(ot_ode_update_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, x, vt, vec, g,
                       one_minus_t, coef, delta, n, total);
    return hipGetLastError();
}

// ---- power-of-two normalisation of the VJP input (keeps the backward's fp16-split operands in range) -----
__global__ void zero_u32_kernel(unsigned int* p) { *p = 0u; }
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, int64_t n, unsigned int* amax_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));     // non-negative floats order like their bit patterns
}
// scale[0] = 2^-ceil(log2(amax)) (1 if amax is 0 or not finite), scale[1] = 1/scale[0]
__global__ void pow2_scale_kernel(const unsigned int* amax_bits, float* scale) {
    const float a = __uint_as_float(*amax_bits);
    float s = 1.f;
    if (a > 0.f && a < 3.0e38f) { int e; frexpf(a, &e); s = ldexpf(1.f, -e); }     // a = m*2^e, m in [0.5,1)  ->  a*s in [0.5,1)
    scale[0] = s; scale[1] = 1.f / s;
}
__global__ __launch_bounds__(256) void scale_kernel(const float* in, float* out, int64_t n, const float* scale) {
    const float s = *scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i] * s;
}
hipError_t launch_vjp_normalise(const float* vec, float* vec_scaled, int64_t n, unsigned int* amax_bits, float* scale, hipStream_t s) {
    hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, s, amax_bits);      // a kernel, not a memset node (see engine.hip zero_fill)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned g = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(g), dim3(256), 0, s, vec, n, amax_bits);
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, s, (const unsigned int*)amax_bits, scale);
    hipLaunchKernelGGL(scale_kernel, dim3(g), dim3(256), 0, s, vec, vec_scaled, n, (const float*)scale);
    return hipGetLastError();
}
hipError_t launch_scale_inplace(float* x, int64_t n, const float* scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, (const float*)x, x, n, scale);
    return hipGetLastError();
}

// per-image PSNR, data range 1, after postprocess (x+1)/2 (utils.py:560-577, 610)
__global__ __launch_bounds__(1024) void psnr_kernel(const float* rec, const float* clean, float* out, int n) {
    __shared__ double s_red[1024];
    const int b = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float a = (rec[(size_t)b * n + i] + 1.0f) * 0.5f, c = (clean[(size_t)b * n + i] + 1.0f) * 0.5f;
        const float dlt = a - c;
        acc += (double)(dlt * dlt);
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = (float)(10.0 * log10(1.0 / (s_red[0] / (double)n)));
}

hipError_t launch_psnr(const float* rec, const float* clean, float* out, int B, int n, hipStream_t s) {
    hipLaunchKernelGGL(psnr_kernel, dim3(B), dim3(1024), 0, s, rec, clean, out, n);
    return hipGetLastError();
}

}  // namespace pf
