// Implicit-GEMM convolution on the gfx950 16-bit matrix cores with fp32-equivalent accuracy.
//
// Same GEMM view, tiling, fusion and epilogue as conv_mfma.hip, but the contraction runs on
// v_mfma_f32_32x32x16_f16 (16x the rate of the fp32 MFMA) with every operand carried as an
// unevaluated fp16 pair  a = a_hi + a_lo,  w = w_hi + w_lo  and the product expanded into three
// MFMAs  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi  accumulated in fp32 (the dropped a_lo*w_lo term is
// 2^-22 relative).  fp16 products are exact in fp32, so the result matches the fp32-MFMA kernel to
// fp32-rounding level (same parity tolerances in tests/), at 3/16 of its matrix-pipe time.
//   * activations: split after GroupNorm/SiLU while staging; LDS row of a pixel = [16 hi | 16 lo | pad]
//     halfs = 80 B (the same conflict-free 20-dword stride as the fp32 patch);
//   * weights: split on the host after an exact 2^8 pre-scale (keeps w_lo a normal fp16 number; the
//     epilogue multiplies by 2^-8), packed [chunk][tap][Cout][16 hi | 16 lo] and staged through LDS
//     (at this MFMA rate the B fragments would need 2/3 of the L1 bandwidth if read per wave).
#include <cstdlib>
#include <hip/hip_fp16.h>
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_fast16(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

template <int MT, int NT, int WM, int WN, int S, int UP>
__global__ __launch_bounds__(256) void conv_mfma16_kernel(const ConvParams p) {
    constexpr int KC = CONV_KC, KQ = KC / 4;
    constexpr int ROW = 20;                      // dwords per LDS row: 8 (hi) + 8 (lo) + 4 (pad)
    constexpr int TH = 2 * MT * WM, TW = 16;
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PP = PH * PW;
    constexpr int BN = WN * NT * 32;
    constexpr int A_F4 = PP * KQ, A_PER = (A_F4 + 255) / 256;
    constexpr int W_U4 = 9 * BN * 4, W_PER = (W_U4 + 255) / 256;   // 64 B (4 x uint4) per (tap, n)
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* s_patch = reinterpret_cast<uint32_t*>(smem_raw);   // [PP][ROW]
    uint32_t* s_w = s_patch + PP * ROW;                           // [9][BN][ROW]
    float* s_sc = reinterpret_cast<float*>(s_w + 9 * BN * ROW);
    float* s_sh = s_sc + ((p.gn_C + 3) & ~3);

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int Hv = UP ? 2 * p.Hs : p.Hs, Wv = UP ? 2 * p.Ws : p.Ws;
    const int qi = tid % KQ, q4 = qi * 4;

    int a_pix[A_PER], a_lds[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int idx = tid + i * 256;
        a_pix[i] = -1; a_lds[i] = -1;
        if (idx < A_F4) {
            const int pix = idx / KQ;
            const int py = pix / PW, px = pix % PW;
            const int gy = oy0 * S - 1 + py, gx = ox0 * S - 1 + px;
            a_lds[i] = pix * ROW + qi * 2;          // dword offset of this thread's 4 hi halfs (lo at +8)
            if (gy >= 0 && gy < Hv && gx >= 0 && gx < Wv && !(UP == 2 && ((gy | gx) & 1))) {
                const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                a_pix[i] = (b * p.Hs + sy) * p.Ws + sx;
            }
        }
    }
    // weight staging descriptors: item idx -> (tap, n, q): global uint4 index and LDS dword offset
    int w_src[W_PER], w_lds[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        const int idx = tid + i * 256;
        const int tap = idx / (BN * 4), rem = idx % (BN * 4), n = rem / 4, q = rem % 4;
        w_lds[i] = idx < W_U4 ? (tap * BN + n) * ROW + q * 4 : -1;
        w_src[i] = (tap * p.Cout + min(n0 + n, p.Cout - 1)) * 4 + q;     // in uint4 units within a chunk block
    }

    float4 ra[A_PER];
    uint4 rw[W_PER];
    auto prefetch = [&](int si, int ch) {
        const ConvSeg& sg = p.seg[si];
        const int c = min(ch * KC + q4, sg.C - 4);
        const float* base = sg.src + sg.coff + c;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ra[i] = *reinterpret_cast<const float4*>(base + (size_t)max(a_pix[i], 0) * sg.cstride);
        const uint4* wb = reinterpret_cast<const uint4*>(sg.w16) + (size_t)ch * sg.taps * p.Cout * 4;
        const int nitems = sg.taps * BN * 4;
#pragma unroll
        for (int i = 0; i < W_PER; ++i) rw[i] = wb[min(tid + i * 256, nitems - 1) == tid + i * 256 ? w_src[i] : 0];
    };
    auto store_lds = [&](int si, int ch) {
        const ConvSeg& sg = p.seg[si];
        const int c = ch * KC + q4;
        const bool cok = c < sg.C;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sg.xform != 0 && cok) {
            sc = *reinterpret_cast<const float4*>(s_sc + sg.gn_off + c);
            sh = *reinterpret_cast<const float4*>(s_sh + sg.gn_off + c);
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            if (a_lds[i] >= 0) {
                float4 v = ra[i];
                if (sg.xform != 0) {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
                    v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (sg.xform == 2) {
                        v.x = silu_fast16(v.x); v.y = silu_fast16(v.y); v.z = silu_fast16(v.z); v.w = silu_fast16(v.w);
                    }
                }
                if (!(cok && a_pix[i] >= 0)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                f16x4 h, l;
                h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
                l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
                l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
                *reinterpret_cast<f16x4*>(s_patch + a_lds[i]) = h;
                *reinterpret_cast<f16x4*>(s_patch + a_lds[i] + 8) = l;
            }
        }
        const int nitems = sg.taps * BN * 4;
#pragma unroll
        for (int i = 0; i < W_PER; ++i)
            if (w_lds[i] >= 0 && tid + i * 256 < nitems) *reinterpret_cast<uint4*>(s_w + w_lds[i]) = rw[i];
    };

    prefetch(0, 0);

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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int prow = l31 >> 4, pcol = l31 & 15;
    const int nbase = n0 + wn * NT * 32 + l31;

    int si = 0, ch = 0;
    while (true) {
        const ConvSeg& sg = p.seg[si];
        __syncthreads();
        store_lds(si, ch);
        __syncthreads();
        int nsi = si, nch = ch + 1;
        if (nch * KC >= sg.C) { nsi = si + 1; nch = 0; }
        const bool more = nsi < p.nseg;
        if (more) prefetch(nsi, nch);

        const int ntaps = sg.taps;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int ky = ntaps == 9 ? tap / 3 : 1, kx = ntaps == 9 ? tap % 3 : 1;
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ppix = (((wm * MT + mt) * 2 + prow) * S + ky) * PW + pcol * S + kx;
                ah[mt] = *reinterpret_cast<const f16x8*>(s_patch + ppix * ROW + hi * 4);
                al[mt] = *reinterpret_cast<const f16x8*>(s_patch + ppix * ROW + 8 + hi * 4);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = tap * BN + (wn * NT + nt) * 32 + l31;
                bh[nt] = *reinterpret_cast<const f16x8*>(s_w + row * ROW + hi * 4);
                bl[nt] = *reinterpret_cast<const f16x8*>(s_w + row * ROW + 8 + hi * 4);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
        if (!more) break;
        si = nsi; ch = nch;
    }

    // ---- epilogue (identical to conv_mfma.hip apart from the 2^-8 weight pre-scale) ----------
    float* s_red = reinterpret_cast<float*>(s_patch);
    const float oscale = p.out_scale * (1.0f / 256.0f);
    if (p.stats_out != nullptr) __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32;
        const bool nok = n < p.Cout;
        const float add = (p.addvec != nullptr && nok) ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int oy = oy0 + (wm * MT + mt) * 2 + (row >> 4), ox = ox0 + (row & 15);
                if (nok && oy < p.H && ox < p.W) {
                    const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                    float v = acc[mt][nt][r] * oscale + add;
                    if (p.residual != nullptr) v += p.residual[pix * p.res_cstride + n];
                    p.out[pix * p.out_cstride + n] = v;
                    s1 += v; s2 += v * v;
                }
            }
        }
        if (p.stats_out != nullptr) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (hi == 0) {
                const int col = (wn * NT + nt) * 32 + l31;
                s_red[(wm * BN + col) * 2] = s1; s_red[(wm * BN + col) * 2 + 1] = s2;
            }
        }
    }
    if (p.stats_out != nullptr) {
        __syncthreads();
        if (tid < BN * 2) {
            const int col = tid >> 1, which = tid & 1;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) tot += s_red[(w * BN + col) * 2 + which];
            const int n = n0 + col;
            if (n < p.Cout) unsafeAtomicAdd(p.stats_out + ((size_t)b * p.Cout + n) * 2 + which, (double)tot);
        }
    }
}

template <int MT, int NT, int WM, int WN, int S, int UP>
static hipError_t launch_cfg16(const ConvParams& p, hipStream_t stream) {
    constexpr int ROW = 20;
    constexpr int TH = 2 * MT * WM, TW = 16;
    constexpr int PP = ((TH - 1) * S + 3) * ((TW - 1) * S + 3);
    constexpr int BN = WN * NT * 32;
    const size_t lds = (size_t)(PP * ROW + 9 * BN * ROW + 2 * ((p.gn_C + 3) & ~3)) * 4;
    static bool attr_set = false;
    auto kern = conv_mfma16_kernel<MT, NT, WM, WN, S, UP>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = p.B * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    dim3 grid(tiles, (p.Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

static long wg_count16(const ConvParams& p, int TH, int BN) {
    return (long)p.B * ((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * ((p.Cout + BN - 1) / BN);
}

template <int S, int UP>
static hipError_t launch_sel16(const ConvParams& p, hipStream_t stream) {
    static const long MIN_WGS = getenv("PNPFLOW_HIP_MIN_WGS") ? atol(getenv("PNPFLOW_HIP_MIN_WGS")) : 512;
    if constexpr (S == 1) {
        if (p.Cout <= 32) return launch_cfg16<2, 1, 4, 1, S, UP>(p, stream);
        if (wg_count16(p, 16, 64) >= MIN_WGS) return launch_cfg16<2, 2, 4, 1, S, UP>(p, stream);
        if (wg_count16(p, 8, 64) >= MIN_WGS) return launch_cfg16<1, 2, 4, 1, S, UP>(p, stream);
        return launch_cfg16<1, 1, 2, 2, S, UP>(p, stream);
    } else {
        if (p.Cout <= 32) return launch_cfg16<1, 1, 4, 1, S, UP>(p, stream);
        if (p.Cout <= 64) return launch_cfg16<1, 2, 4, 1, S, UP>(p, stream);
        return launch_cfg16<1, 1, 2, 2, S, UP>(p, stream);
    }
}

// Only fragment-major packed weights (w_mode 0) with a 16-bit repack are supported; the caller falls
// back to the fp32 kernel (launch_conv) for the generic strided operands of attention.
hipError_t launch_conv16(const ConvParams& p, int stride, int up, hipStream_t stream) {
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].w_mode != 0 || p.seg[i].w16 == nullptr) return hipErrorInvalidValue;
    if (stride == 2) return launch_sel16<2, 0>(p, stream);
    if (up == 2) return launch_sel16<1, 2>(p, stream);
    if (up) return launch_sel16<1, 1>(p, stream);
    return launch_sel16<1, 0>(p, stream);
}

}  // namespace pf
