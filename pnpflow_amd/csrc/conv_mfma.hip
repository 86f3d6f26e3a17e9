// Implicit-GEMM convolution / GEMM on the gfx950 matrix cores (exact-fp32 MFMA).
//
// One kernel family serves every dense contraction of the U-Net velocity field
// (reference: pnpflow/models.py:58-162, 442-495):
//   * 3x3 convs (stride 1, stride 2, fused nearest-x2 upsample) with GroupNorm(+SiLU)
//     applied to the input while it is staged into LDS, bias + time-embedding broadcast
//     + residual fused in the epilogue, two-source input for the skip concatenation,
//     the 1x1 shortcut folded in as extra K-segments;
//   * 1x1 convs (attention q/k/v, proj_out) and the attention matmuls q^T k and v A^T as
//     1-tap "convs" with per-sample weights.
// The epilogue also emits per-channel (sum, sumsq) of the produced tensor so that the
// next GroupNorm never re-reads it from HBM.
//
// Tiling (wave64, 4 waves / workgroup):  output tile = (2*MW rows x 16 cols) pixels x
// (NW*NT*32) channels; wave (wm, wn) owns 2 rows x 16 cols = 32 pixels (the M of a
// 32x32x2 MFMA) x NT*32 channels.  K is walked in chunks of KC=16 input channels: the
// (halo) input patch of the chunk and the 9 (or 1) weight taps are staged once in LDS
// ([pixel][KC+4] / [tap][n][KC+4] fp32, the +4 pad makes the ds_read_b128 fragment
// reads bank-conflict free) and reused by all taps; the next chunk's global loads are
// in flight (registers) while the MFMAs of the current chunk issue.
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_fast(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

template <int MW, int NW, int NT, int S, int UP>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr int KC = CONV_KC, KCP = KC + 4, KQ = KC / 4;
    constexpr int TH = 2 * MW, TW = 16;
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PP = PH * PW;
    constexpr int BN = NW * NT * 32;
    constexpr int A_F4 = PP * KQ, A_PER = (A_F4 + 255) / 256;
    constexpr int W_F4 = 9 * BN * KQ, W_PER = (W_F4 + 255) / 256;
    static_assert(MW * NW == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_patch = reinterpret_cast<float*>(smem_raw);   // [PP][KCP]
    float* s_w = s_patch + PP * KCP;                        // [9][BN][KCP]
    float* s_sc = s_w + 9 * BN * KCP;                       // [gn_C] GroupNorm scale
    float* s_sh = s_sc + ((p.gn_C + 3) & ~3);               // [gn_C] GroupNorm shift

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave % MW, wn = wave / MW;
    const int Hv = UP ? 2 * p.Hs : p.Hs, Wv = UP ? 2 * p.Ws : p.Ws;

    // ---- GroupNorm scale/shift of this sample from the producers' per-channel stats ----
    for (int c = tid; c < p.gn_C; c += 256) {
        const int g = c / p.gn_cpg;
        double s = 0.0, ss = 0.0;
        for (int j = g * p.gn_cpg; j < (g + 1) * p.gn_cpg; ++j) {
            for (int si = 0; si < p.nseg; ++si) {
                const ConvSeg& sg = p.seg[si];
                if (sg.xform != 0 && j >= sg.gn_off && j < sg.gn_off + sg.C) {
                    const double* st = sg.stats + ((size_t)b * sg.C + (j - sg.gn_off)) * 2;
                    s += st[0]; ss += st[1];
                }
            }
        }
        const double N = (double)p.gn_cpg * (double)p.Hs * (double)p.Ws;
        const double mean = s / N;
        double var = ss / N - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
        const float sc = p.gamma[c] * rstd;
        s_sc[c] = sc;
        s_sh[c] = p.beta[c] - (float)mean * sc;
    }

    float4 ra[A_PER];
    float4 rw[W_PER];

    auto prefetch = [&](int si, int ch) {
        const ConvSeg& sg = p.seg[si];
        const int c0 = ch * KC;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int idx = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < A_F4) {
                const int pix = idx / KQ, q = idx % KQ;
                const int py = pix / PW, px = pix % PW;
                const int gy = oy0 * S - 1 + py, gx = ox0 * S - 1 + px;
                const int c = c0 + q * 4;
                if (gy >= 0 && gy < Hv && gx >= 0 && gx < Wv && c < sg.C) {
                    const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                    v = *reinterpret_cast<const float4*>(sg.src + ((size_t)(b * p.Hs + sy) * p.Ws + sx) * sg.cstride + sg.coff + c);
                }
            }
            ra[i] = v;
        }
        const float* wb = sg.w + (size_t)b * sg.w_bs;
        if (sg.w_mode == 0) {
            const int total = sg.taps * BN * KQ;
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int idx = tid + i * 256;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total) {
                    const int tap = idx / (BN * KQ), rem = idx % (BN * KQ);
                    const int n = n0 + rem / KQ, q = rem % KQ;
                    if (n < p.Cout && c0 + q * 4 < sg.C)
                        v = *reinterpret_cast<const float4*>(wb + (size_t)ch * sg.w_cs + (size_t)tap * sg.w_ts + (size_t)n * sg.w_ns + q * 4);
                }
                rw[i] = v;
            }
        } else {
            constexpr int total = KC * (BN / 4);
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int idx = tid + i * 256;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total) {
                    const int kk = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
                    if (n < p.Cout && c0 + kk < sg.C)
                        v = *reinterpret_cast<const float4*>(wb + (size_t)(c0 + kk) * sg.w_ks + n);
                }
                rw[i] = v;
            }
        }
    };

    auto store_lds = [&](int si, int ch) {
        const ConvSeg& sg = p.seg[si];
        const int c0 = ch * KC;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int idx = tid + i * 256;
            if (idx < A_F4) {
                const int pix = idx / KQ, q = idx % KQ;
                float4 v = ra[i];
                if (sg.xform != 0) {
                    const int py = pix / PW, px = pix % PW;
                    const int gy = oy0 * S - 1 + py, gx = ox0 * S - 1 + px;
                    const int c = c0 + q * 4;
                    if (gy >= 0 && gy < Hv && gx >= 0 && gx < Wv && c < sg.C) {
                        const float4 sc = *reinterpret_cast<const float4*>(s_sc + sg.gn_off + c);
                        const float4 sh = *reinterpret_cast<const float4*>(s_sh + sg.gn_off + c);
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
                        v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        if (sg.xform == 2) {
                            v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                        }
                    }
                }
                *reinterpret_cast<float4*>(s_patch + pix * KCP + q * 4) = v;
            }
        }
        if (sg.w_mode == 0) {
            const int total = sg.taps * BN * KQ;
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int idx = tid + i * 256;
                if (idx < total) {
                    const int row = idx / KQ, q = idx % KQ;   // row = tap*BN + n
                    *reinterpret_cast<float4*>(s_w + row * KCP + q * 4) = rw[i];
                }
            }
        } else {
            constexpr int total = KC * (BN / 4);
#pragma unroll
            for (int i = 0; i < W_PER; ++i) {
                const int idx = tid + i * 256;
                if (idx < total) {
                    const int kk = idx / (BN / 4), n = (idx % (BN / 4)) * 4;
                    s_w[(n + 0) * KCP + kk] = rw[i].x; s_w[(n + 1) * KCP + kk] = rw[i].y;
                    s_w[(n + 2) * KCP + kk] = rw[i].z; s_w[(n + 3) * KCP + kk] = rw[i].w;
                }
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    const int prow = l31 >> 4, pcol = l31 & 15;
    int si = 0, ch = 0;
    prefetch(0, 0);
    while (true) {
        __syncthreads();          // all waves finished reading the previous chunk (and s_sc is written)
        store_lds(si, ch);
        __syncthreads();
        int nsi = si, nch = ch + 1;
        if (nch * KC >= p.seg[si].C) { nsi = si + 1; nch = 0; }
        const bool more = nsi < p.nseg;
        if (more) prefetch(nsi, nch);

        const int ntaps = p.seg[si].taps;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int ky = ntaps == 9 ? tap / 3 : 1, kx = ntaps == 9 ? tap % 3 : 1;
            const int ppix = ((wm * 2 + prow) * S + ky) * PW + pcol * S + kx;
            const float* ap = s_patch + ppix * KCP + hi * 4;
            const float* bp = s_w + (tap * BN + wn * NT * 32 + l31) * KCP + hi * 4;
#pragma unroll
            for (int ks = 0; ks < KC / 8; ++ks) {
                const float4 a = *reinterpret_cast<const float4*>(ap + ks * 8);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + nt * 32 * KCP + ks * 8);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, acc[nt], 0, 0, 0);
                }
            }
        }
        if (!more) break;
        si = nsi; ch = nch;
    }

    // ---- epilogue: scale, bias(+temb), residual, store NHWC, per-channel statistics ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + l31;
        const bool nok = n < p.Cout;
        const float add = (p.addvec != nullptr && nok) ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int oy = oy0 + wm * 2 + (row >> 4), ox = ox0 + (row & 15);
            if (nok && oy < p.H && ox < p.W) {
                const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                float v = acc[nt][r] * p.out_scale + add;
                if (p.residual != nullptr) v += p.residual[pix * p.res_cstride + n];
                p.out[pix * p.out_cstride + n] = v;
                s1 += v; s2 += v * v;
            }
        }
        if (p.stats_out != nullptr) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (hi == 0 && nok) {
                double* st = p.stats_out + ((size_t)b * p.Cout + n) * 2;
                unsafeAtomicAdd(st, (double)s1);
                unsafeAtomicAdd(st + 1, (double)s2);
            }
        }
    }
}

template <int MW, int NW, int NT, int S, int UP>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int KCP = CONV_KC + 4;
    constexpr int TH = 2 * MW, TW = 16;
    constexpr int PP = ((TH - 1) * S + 3) * ((TW - 1) * S + 3);
    constexpr int BN = NW * NT * 32;
    const size_t lds = (size_t)(PP * KCP + 9 * BN * KCP + 2 * ((p.gn_C + 3) & ~3)) * sizeof(float);
    static size_t lds_set = 0;
    auto kern = conv_mfma_kernel<MW, NW, NT, S, UP>;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lds_set = 160 * 1024;
    }
    const int tiles = p.B * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    dim3 grid(tiles, (p.Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int S, int UP>
static hipError_t launch_sel(const ConvParams& p, hipStream_t stream) {
    if (p.Cout <= 32) return launch_cfg<4, 1, 1, S, UP>(p, stream);
    if (p.Cout <= 64) return launch_cfg<4, 1, 2, S, UP>(p, stream);
    const long wgs128 = (long)p.B * ((p.H + 3) / 4) * ((p.W + 15) / 16) * ((p.Cout + 127) / 128);
    if (wgs128 >= 1024) return launch_cfg<2, 2, 2, S, UP>(p, stream);
    return launch_cfg<2, 2, 1, S, UP>(p, stream);
}

hipError_t launch_conv(const ConvParams& p, int stride, int up, hipStream_t stream) {
    if (stride == 2) return launch_sel<2, 0>(p, stream);
    if (up) return launch_sel<1, 1>(p, stream);
    return launch_sel<1, 0>(p, stream);
}

size_t conv_flops(const ConvParams& p) {
    size_t k = 0;
    for (int i = 0; i < p.nseg; ++i) k += (size_t)p.seg[i].taps * p.seg[i].C;
    return 2 * (size_t)p.B * p.H * p.W * p.Cout * k;
}

}  // namespace pf
