// Fourier-domain solve of OT-ODE's linear system for the circular Gaussian blur
// (pnpflow/methods/ot_ode.py:108-117):
//     sol = ifft2( fft2(d) / (r_t^2 |fft2(filter)|^2 + sigma^2) )
// Hand-written line-batched FFT in LDS (no hipFFT): one workgroup transforms LB lines of N complex points,
// radix-2 Stockham autosort when N is a power of two, a direct O(N^2) DFT otherwise (N <= 2048).
// The 2-D transform is rows -> columns; the column kernel applies the spectral division and goes straight
// back (forward FFT, scale, inverse FFT of the same LDS-resident columns), so the solve is 3 launches:
//     rows  forward  (real d = y - H(x1) in, complex spectrum out)
//     cols  forward + divide + inverse (in place)
//     rows  inverse  (complex in, real sol out, 1/(H W) folded in)
// The blur filter is separable (outer(g,g), utils.py:273-280) so |fft2(filter)|^2(u,v) = P_H[u] * P_W[v] with
// P_N[u] = |sum_j g[j] exp(-2 pi i u (j-r)/N)|^2, evaluated in fp64 by a tiny kernel.
#include "pf_common.h"

namespace pf {
namespace {

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// tw[m] = exp(-2 pi i m / N), m in [0, N)
__device__ inline void fill_twiddles(float2* tw, int N) {
    for (int m = threadIdx.x; m < N; m += blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)m / (float)N, &s, &c);
        tw[m] = make_float2(c, -s);
    }
}

// Transforms LB lines (line l at a[l*N .. l*N+N)) in LDS; returns the buffer that holds the result.
// inverse: conjugated twiddles, no normalisation.
__device__ float2* fft_lines(float2* a, float2* b, const float2* tw, int N, int log2n, int LB, int inverse) {
    const float sgn = inverse ? -1.f : 1.f;
    if (log2n >= 0) {
        const int half = N >> 1;
        for (int s = 0; s < log2n; ++s) {
            const int p = 1 << s;
            const int tstep = half >> s;                         // exp(-i pi k / p) = tw[k * (N/2)/p]
            for (int idx = threadIdx.x; idx < LB * half; idx += blockDim.x) {
                const int l = idx / half, j = idx - l * half;
                const int k = j & (p - 1);
                float2 w = tw[k * tstep]; w.y *= sgn;
                const float2 u0 = a[l * N + j];
                const float2 u1 = cmul(w, a[l * N + j + half]);
                const int o = ((j - k) << 1) + k;
                b[l * N + o] = make_float2(u0.x + u1.x, u0.y + u1.y);
                b[l * N + o + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
            __syncthreads();
            float2* t = a; a = b; b = t;
        }
        return a;
    }
    for (int idx = threadIdx.x; idx < LB * N; idx += blockDim.x) {
        const int l = idx / N, k = idx - l * N;
        float2 acc = make_float2(0.f, 0.f);
        int m = 0;
        for (int n = 0; n < N; ++n) {
            float2 w = tw[m]; w.y *= sgn;
            const float2 v = cmul(w, a[l * N + n]);
            acc.x += v.x; acc.y += v.y;
            m += k; m -= m >= N ? N : 0;
        }
        b[l * N + k] = acc;
    }
    __syncthreads();
    return b;
}

// rows forward: z[plane][row][:] = FFT_W( y - hx )
__global__ __launch_bounds__(256) void fft_rows_fwd_kernel(const float* y, const float* hx, float2* z, int H, int W, int log2w, int LB) {
    extern __shared__ float2 smem[];
    float2* a = smem; float2* b = a + LB * W; float2* tw = b + LB * W;
    fill_twiddles(tw, W);
    const size_t plane = (size_t)blockIdx.y * H * W;
    const int row0 = blockIdx.x * LB;
    for (int idx = threadIdx.x; idx < LB * W; idx += 256) {
        const int l = idx / W, k = idx - l * W;
        float v = 0.f;
        if (row0 + l < H) { const size_t o = plane + (size_t)(row0 + l) * W + k; v = y[o] - hx[o]; }
        a[idx] = make_float2(v, 0.f);
    }
    __syncthreads();
    const float2* r = fft_lines(a, b, tw, W, log2w, LB, 0);
    for (int idx = threadIdx.x; idx < LB * W; idx += 256) {
        const int l = idx / W, k = idx - l * W;
        if (row0 + l < H) z[plane + (size_t)(row0 + l) * W + k] = r[idx];
    }
}

// columns: forward FFT_H, divide by (rt2 * P_H[u] * P_W[v] + sigma2), inverse FFT_H - in place
__global__ __launch_bounds__(256) void fft_cols_solve_kernel(float2* z, const float* pw_h, const float* pw_w, const float* rt2, float sigma2,
                                                             int C, int H, int W, int log2h, int LB) {
    extern __shared__ float2 smem[];
    float2* a = smem; float2* b = a + LB * H; float2* tw = b + LB * H;
    fill_twiddles(tw, H);
    const size_t plane = (size_t)blockIdx.y * H * W;
    const float r2 = rt2[blockIdx.y / C];
    const int col0 = blockIdx.x * LB;
    for (int idx = threadIdx.x; idx < LB * H; idx += 256) {
        const int k = idx / LB, l = idx - k * LB;                 // adjacent lanes -> adjacent columns
        a[l * H + k] = col0 + l < W ? z[plane + (size_t)k * W + col0 + l] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    float2* r = fft_lines(a, b, tw, H, log2h, LB, 0);
    float2* o = r == a ? b : a;
    for (int idx = threadIdx.x; idx < LB * H; idx += 256) {
        const int l = idx / H, u = idx - l * H;
        const float pv = col0 + l < W ? pw_w[col0 + l] : 0.f;
        const float inv = r2 * (pw_h[u] * pv) + sigma2;           // ot_ode.py:113-115
        r[idx] = make_float2(r[idx].x / inv, r[idx].y / inv);     // :116
    }
    __syncthreads();
    const float2* q = fft_lines(r, o, tw, H, log2h, LB, 1);
    for (int idx = threadIdx.x; idx < LB * H; idx += 256) {
        const int k = idx / LB, l = idx - k * LB;
        if (col0 + l < W) z[plane + (size_t)k * W + col0 + l] = q[l * H + k];
    }
}

// rows inverse: sol[plane][row][:] = Re( IFFT_W( z ) ) / (H W)
__global__ __launch_bounds__(256) void fft_rows_inv_kernel(const float2* z, float* sol, int H, int W, int log2w, int LB) {
    extern __shared__ float2 smem[];
    float2* a = smem; float2* b = a + LB * W; float2* tw = b + LB * W;
    fill_twiddles(tw, W);
    const size_t plane = (size_t)blockIdx.y * H * W;
    const int row0 = blockIdx.x * LB;
    for (int idx = threadIdx.x; idx < LB * W; idx += 256) {
        const int l = idx / W, k = idx - l * W;
        a[idx] = row0 + l < H ? z[plane + (size_t)(row0 + l) * W + k] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const float2* r = fft_lines(a, b, tw, W, log2w, LB, 1);
    const float norm = 1.0f / ((float)H * (float)W);
    for (int idx = threadIdx.x; idx < LB * W; idx += 256) {
        const int l = idx / W, k = idx - l * W;
        if (row0 + l < H) sol[plane + (size_t)(row0 + l) * W + k] = r[idx].x * norm;
    }
}

// P_N[u] = |DFT_N of the rolled, zero-padded 1-D taps|^2  (degradations.py:62-68: peak rolled to index 0)
__global__ void blur_power_spectrum_kernel(const float* taps, int ntaps, int N, float* pw) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= N) return;
    const int r = ntaps / 2;
    double re = 0.0, im = 0.0;
    for (int j = 0; j < ntaps; ++j) {
        long long m = ((long long)u * (j - r)) % N;               // exp(-2 pi i u (j-r) / N)
        if (m < 0) m += N;
        double s, c;
        sincospi(2.0 * (double)m / (double)N, &s, &c);
        re += (double)taps[j] * c; im -= (double)taps[j] * s;
    }
    pw[u] = (float)(re * re + im * im);
}

__global__ __launch_bounds__(256) void x1_hat_kernel(const float* x, const float* vt, const float* omt, float* out, int n, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        out[i] = x[i] + omt[i / n] * vt[i];                       // ot_ode.py:74
}

inline int ilog2_exact(int n) { int l = 0; while ((1 << l) < n) ++l; return (1 << l) == n ? l : -1; }
inline int lines_per_wg(int N) { int lb = 2048 / N; return lb > 8 ? 8 : (lb < 1 ? 1 : lb); }

}  // namespace

// vec = H_adj( sol ),  sol = ifft2( fft2( y - H(x + (1-t) v_t) ) / (r_t^2 |fft2 filter|^2 + sigma^2) )
// scratch: 4*B*C*H*W + H + W floats
hipError_t launch_ot_ode_vec_blur(const DegView& d, const float* x, const float* vt, const float* y, const float* one_minus_t,
                                  const float* rt2, float sigma2, float* vec, int B, int C, int H, int W, float* scratch, hipStream_t s) {
    if (!scratch || d.kind != DEG_BLUR || H > 2048 || W > 2048 || d.ntaps > H || d.ntaps > W) return hipErrorInvalidValue;
    const size_t S = (size_t)B * C * H * W;
    float* buf0 = scratch; float* buf1 = scratch + S; float2* z = reinterpret_cast<float2*>(scratch + 2 * S);
    float* pw_h = scratch + 4 * S; float* pw_w = pw_h + H;
    hipLaunchKernelGGL(blur_power_spectrum_kernel, dim3((H + 63) / 64), dim3(64), 0, s, d.taps, d.ntaps, H, pw_h);
    hipLaunchKernelGGL(blur_power_spectrum_kernel, dim3((W + 63) / 64), dim3(64), 0, s, d.taps, d.ntaps, W, pw_w);
    hipLaunchKernelGGL(x1_hat_kernel, dim3((unsigned)std::min<size_t>((S + 255) / 256, 4096)), dim3(256), 0, s, x, vt, one_minus_t, buf0,
                       C * H * W, (int64_t)S);
    hipError_t r = launch_deg_H(d, buf0, buf1, B, C, H, W, reinterpret_cast<float*>(z), s);          // buf1 = H(x1_hat)
    if (r != hipSuccess) return r;
    const int lbw = lines_per_wg(W), lbh = lines_per_wg(H);
    const size_t lds_w = (size_t)(2 * lbw + 1) * W * sizeof(float2), lds_h = (size_t)(2 * lbh + 1) * H * sizeof(float2);
    hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3((H + lbw - 1) / lbw, B * C), dim3(256), lds_w, s, y, (const float*)buf1, z, H, W, ilog2_exact(W), lbw);
    hipLaunchKernelGGL(fft_cols_solve_kernel, dim3((W + lbh - 1) / lbh, B * C), dim3(256), lds_h, s, z, (const float*)pw_h, (const float*)pw_w, rt2,
                       sigma2, C, H, W, ilog2_exact(H), lbh);
    hipLaunchKernelGGL(fft_rows_inv_kernel, dim3((H + lbw - 1) / lbw, B * C), dim3(256), lds_w, s, (const float2*)z, buf0, H, W, ilog2_exact(W), lbw);
    r = hipGetLastError();
    if (r != hipSuccess) return r;
    return launch_deg_Hadj(d, buf0, vec, B, C, H, W, buf1, s);                                        // vec = H_adj(sol), ot_ode.py:130
}

}  // namespace pf
