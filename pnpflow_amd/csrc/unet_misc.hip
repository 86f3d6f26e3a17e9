// Small kernels of the U-Net velocity field that are not dense contractions:
// image-boundary convs (3->ch and ch->3, VALU), time embedding, softmax rows,
// stand-alone per-channel statistics.  Reference: pnpflow/models.py:253-299, 442-495.
#include "pf_common.h"

namespace pf {

__device__ __forceinline__ float silu_acc(float x) { return x / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------
// begin_conv: NCHW image -> NHWC [B][H][W][C], 3x3 pad 1 (models.py:358, 451).
// One thread per output pixel, all C=32 output channels in registers; weights in LDS.
// ------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void begin_conv_kernel(const EdgeConvParams p) {
    __shared__ float s_w[9 * 3 * C + C];
    const int nw = 9 * p.Cimg * C;
    for (int i = threadIdx.x; i < nw; i += 256) s_w[i] = p.w[i];
    for (int i = threadIdx.x; i < C; i += 256) s_w[9 * 3 * C + i] = p.bias[i];
    __syncthreads();
    const int HW = p.H * p.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (pix >= HW) return;
    const int y = pix / p.W, x = pix % p.W;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = s_w[9 * 3 * C + c];
    for (int ci = 0; ci < p.Cimg; ++ci) {
        const float* img = p.in + ((size_t)b * p.Cimg + ci) * HW;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            float v = 0.f;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = img[yy * p.W + xx];
            const float* w = s_w + (tap * p.Cimg + ci) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fmaf(v, w[c], acc[c]);
        }
    }
    float4* o = reinterpret_cast<float4*>(p.out + ((size_t)b * HW + pix) * C);
#pragma unroll
    for (int c = 0; c < C / 4; ++c) o[c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
}

hipError_t launch_begin_conv(const EdgeConvParams& p, hipStream_t s) {
    if (p.C != 32 || p.Cimg > 3) return hipErrorInvalidValue;
    dim3 grid((p.H * p.W + 255) / 256, p.B);
    hipLaunchKernelGGL(begin_conv_kernel<32>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// end_conv: GroupNorm -> SiLU -> 3x3 conv C->Cimg, NHWC in, NCHW image out
// (models.py:428-433, 492).  16x16 output tile per workgroup; the normalised+activated
// 18x18xC patch lives in LDS ([pix][C+4]).
// ------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void end_conv_kernel(const EdgeConvParams p) {
    constexpr int CP = C + 4, PW = 18, PP = PW * PW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_patch = reinterpret_cast<float*>(smem_raw);  // [PP][CP]
    float* s_w = s_patch + PP * CP;                        // [9][3][C]
    float* s_sc = s_w + 9 * 3 * C;                         // [C]
    float* s_sh = s_sc + C;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 15) / 16;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * p.Cimg * C; i += 256) s_w[i] = p.w[i];
    const bool raw = p.stats == nullptr;   // raw: plain 3x3 conv (used as the adjoint of begin_conv)
    if (tid < C && !raw) {
        const int g = tid / p.gn_cpg;
        double s = 0.0, ss = 0.0;
        for (int j = g * p.gn_cpg; j < (g + 1) * p.gn_cpg; ++j) {
            const double* st = p.stats + ((size_t)b * C + j) * 2;
            s += st[0]; ss += st[1];
        }
        const double N = (double)p.gn_cpg * p.H * p.W;
        const double mean = s / N;
        double var = ss / N - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
        const float sc = p.gamma[tid] * rstd;
        s_sc[tid] = sc;
        s_sh[tid] = p.beta[tid] - (float)mean * sc;
    }
    __syncthreads();
    const int oy0 = ty * 16, ox0 = tx * 16;
    for (int idx = tid; idx < PP * (C / 4); idx += 256) {
        const int pix = idx / (C / 4), q = idx % (C / 4);
        const int gy = oy0 - 1 + pix / PW, gx = ox0 - 1 + pix % PW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            v = *reinterpret_cast<const float4*>(p.in + ((size_t)(b * p.H + gy) * p.W + gx) * C + q * 4);
            if (!raw) {
                const float4 sc = *reinterpret_cast<const float4*>(s_sc + q * 4);
                const float4 sh = *reinterpret_cast<const float4*>(s_sh + q * 4);
                v.x = silu_acc(v.x * sc.x + sh.x); v.y = silu_acc(v.y * sc.y + sh.y);
                v.z = silu_acc(v.z * sc.z + sh.z); v.w = silu_acc(v.w * sc.w + sh.w);
            }
        }
        *reinterpret_cast<float4*>(s_patch + pix * CP + q * 4) = v;
    }
    __syncthreads();
    const int ly = tid / 16, lx = tid % 16;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* ap = s_patch + ((ly + tap / 3) * PW + lx + tap % 3) * CP;
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(ap + q * 4);
            for (int co = 0; co < p.Cimg; ++co) {
                const float4 w = *reinterpret_cast<const float4*>(s_w + (tap * p.Cimg + co) * C + q * 4);
                acc[co] = fmaf(a.x, w.x, acc[co]); acc[co] = fmaf(a.y, w.y, acc[co]);
                acc[co] = fmaf(a.z, w.z, acc[co]); acc[co] = fmaf(a.w, w.w, acc[co]);
            }
        }
    }
    if (oy < p.H && ox < p.W)
        for (int co = 0; co < p.Cimg; ++co)
            p.out[((size_t)(b * p.Cimg + co) * p.H + oy) * p.W + ox] = acc[co] + p.bias[co];
}

hipError_t launch_end_conv(const EdgeConvParams& p, hipStream_t s) {
    if (p.C != 32 || p.Cimg > 3) return hipErrorInvalidValue;
    constexpr int C = 32;
    const size_t lds = (size_t)(18 * 18 * (C + 4) + 9 * 3 * C + 2 * C) * sizeof(float);
    dim3 grid(p.B * ((p.H + 15) / 16) * ((p.W + 15) / 16));
    hipLaunchKernelGGL(end_conv_kernel<C>, grid, dim3(256), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Time embedding + every ResidualBlock's temb_proj in one launch
// (models.py:253-299 and :101): out[b][j] = bp[j] + Wp[j] . silu(MLP(sinusoidal(t[b]))).
// bp already contains conv1.bias so the conv epilogue adds a single vector.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temb_kernel(const TembParams p) {
    __shared__ float s_e[64];
    __shared__ float s_h[512];
    __shared__ float s_s[512];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ch = p.ch, tch = 4 * p.ch, half = ch / 2;
    if (tid < half) {
        // freq_i = exp(-i * ln(1e4)/(half-1))  (models.py:270-273), t used raw
        const float f = expf((float)tid * -(logf(10000.0f) / (float)(half - 1)));
        const float a = p.t[b] * f;
        s_e[tid] = sinf(a);
        s_e[half + tid] = cosf(a);
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b0[j];
        for (int k = 0; k < ch; ++k) acc = fmaf(p.w0[j * ch + k], s_e[k], acc);
        s_h[j] = silu_acc(acc);
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b1[j];
        for (int k = 0; k < tch; ++k) acc = fmaf(p.w1[j * tch + k], s_h[k], acc);
        s_s[j] = silu_acc(acc);   // every consumer applies act(temb) first (models.py:101)
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

hipError_t launch_temb(const TembParams& p, hipStream_t s) {
    if (p.ch > 64 || 4 * p.ch > 512 || (p.ch & 1)) return hipErrorInvalidValue;
    dim3 grid((p.total_out + 255) / 256, p.B);
    hipLaunchKernelGGL(temb_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// softmax over the last dim, in place; one wave per row (models.py:154-155).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* data, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = data + row * cols;
    float m = -INFINITY;
    for (int i = lane; i < cols; i += 64) m = fmaxf(m, r[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) {
        const float e = expf(r[i] - m);
        r[i] = e;
        s += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < cols; i += 64) r[i] *= inv;
}

hipError_t launch_softmax_rows(float* data, int64_t rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, data, rows, cols);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// per-channel (sum, sumsq) of an NHWC tensor: stats[b][c][2] += ...   (HBM-bound)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* x, double* stats, int HW, int C, int pix_per_block) {
    __shared__ double s_red[256 * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cq = C / 4;                 // float4 groups per pixel
    const int lanes_p = 256 / cq;         // pixels processed concurrently (C <= 1024)
    const int q = tid % cq, pr = tid / cq;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (pr < lanes_p) {
        for (int pix = p0 + pr; pix < p1; pix += lanes_p) {
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + pix) * C + q * 4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
        }
    }
    for (int j = 0; j < 4; ++j) {
        __syncthreads();
        s_red[tid * 2] = s[j]; s_red[tid * 2 + 1] = ss[j];
        __syncthreads();
        if (pr == 0 && tid < cq) {
            double a = 0, c2 = 0;
            for (int r = 0; r < lanes_p; ++r) { a += s_red[(r * cq + q) * 2]; c2 += s_red[(r * cq + q) * 2 + 1]; }
            double* st = stats + ((size_t)b * C + q * 4 + j) * 2;
            unsafeAtomicAdd(st, a);
            unsafeAtomicAdd(st + 1, c2);
        }
    }
}

hipError_t launch_channel_stats(const float* x, double* stats, int B, int HW, int C, hipStream_t s) {
    if (C % 4 != 0 || C / 4 > 256) return hipErrorInvalidValue;
    const int ppb = 1024;
    dim3 grid((HW + ppb - 1) / ppb, B);
    hipLaunchKernelGGL(channel_stats_kernel, grid, dim3(256), 0, s, x, stats, HW, C, ppb);
    return hipGetLastError();
}

}  // namespace pf
