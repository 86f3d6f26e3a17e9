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
// Tiling (wave64, 4 waves = WM x WN per workgroup): workgroup tile = (2*MT*WM rows x 16
// cols) pixels x (WN*NT*32) output channels; wave (wm, wn) owns MT x NT MFMA tiles of
// 32 pixels (2 rows x 16 cols) x 32 channels, i.e. MT*NT independent fp32 accumulators.
// K is walked in chunks of KC=16 input channels:
//   A (activations): the halo patch of the chunk is normalised/activated once and staged
//      in LDS as [pixel][KC+4] fp32 (the +4 pad makes the ds_read_b128 fragment reads
//      conflict free); all 9 taps re-read it with shifted addresses.  The next chunk's
//      global loads are in flight (registers) while the MFMAs of the current chunk issue.
//   B (weights): fragments are read straight from L1/L2 into registers, two k-steps ahead
//      of their use, from a fragment-major repack [chunk][tap][kstep][Cout][8] in which
//      one wave-level fragment (32 channels x 8 k) is one contiguous 1 KiB line group.
//      (fp32 MFMA consumes 2 KiB of operands per 1024 MFMA cycles per wave: the weights
//      stay L2/L1 resident and need no LDS staging, which leaves LDS to the patch and
//      lets 3-6 workgroups share a CU.)
#include <cstdlib>
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_exp_f32 / v_rcp_f32 (1 ulp each); __frcp_rn would be a correctly rounded division: 10 instructions per element
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int MT, int NT, int WM, int WN, int S, int UP, int BM, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    // KC = input channels per K-chunk: 16 for 3x3 segments; 64 for launches made only of 1-tap segments
    // (1x1 convs, attention matmuls), whose chunks would otherwise be 2 k-steps between two barriers
    constexpr int KCP = KC + 4, KQ = KC / 4, KS = KC / 8;
    constexpr int TH = 2 * MT * WM, TW = 16;
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PP = PH * PW;
    constexpr int BN = WN * NT * 32;
    constexpr bool A_PREFETCH = false;   // measured: prefetching the LDS fragments one k-step ahead costs 8 % (more VGPRs, clumped ds_reads)
    constexpr int A_F4 = PP * KQ, A_PER = (A_F4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_patch = reinterpret_cast<float*>(smem_raw);   // [PP][KCP]
    constexpr int EPI = 4 * 32 * 36 + WM * BN * 2;         // floats needed by the epilogue (4 transpose tiles + statistics)
    float* s_sc = s_patch + (PP * KCP > EPI ? PP * KCP : EPI);   // [gn_C] GroupNorm scale
    float* s_sh = s_sc + ((p.gn_C + 3) & ~3);               // [gn_C] GroupNorm shift

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int Hv = UP ? 2 * p.Hs : p.Hs, Wv = UP ? 2 * p.Ws : p.Ws;
    const int q4 = (tid % KQ) * 4;      // channel offset of this thread's float4 inside a chunk

    // ---- per-thread staging descriptors, identical for every K-chunk ----------------------
    int a_pix[A_PER];    // source pixel index (b*Hs+sy)*Ws+sx, or -1: zero padding / outside
    int a_lds[A_PER];    // float offset of the float4 in s_patch, or -1: nothing to store
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int idx = tid + i * 256;
        a_pix[i] = -1; a_lds[i] = -1;
        if (idx < A_F4) {
            const int pix = idx / KQ;
            const int py = pix / PW, px = pix % PW;
            const int gy = oy0 * S - 1 + py, gx = ox0 * S - 1 + px;
            a_lds[i] = pix * KCP + q4;
            // UP == 1: nearest x2 upsample of the source; UP == 2: zero-insertion x2 (adjoint of a stride-2 conv)
            if (gy >= 0 && gy < Hv && gx >= 0 && gx < Wv && !(UP == 2 && ((gy | gx) & 1))) {
                const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                a_pix[i] = (b * p.Hs + sy) * p.Ws + sx;
            }
        }
    }

    // All global loads below are unconditional (indices are clamped into the tensor and the zero
    // padding is applied when the value is consumed): a load inside a divergent branch makes hipcc
    // wait for it right away, which would serialise the software pipeline.
    float4 ra[A_PER];
    auto prefetch = [&](int si, int ch) {
        const ConvSeg& sg = p.seg[si];
        const int c = min(ch * KC + q4, sg.C - 4);
        const float* base = sg.src + sg.coff + c;
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            ra[i] = *reinterpret_cast<const float4*>(base + (size_t)max(a_pix[i], 0) * sg.cstride);
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
                        v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                    }
                }
                if (!(cok && a_pix[i] >= 0)) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER norm+activation
                *reinterpret_cast<float4*>(s_patch + a_lds[i]) = v;
            }
        }
    };

    // first chunk's activations go in flight before anything else
    prefetch(0, 0);

    // ---- GroupNorm scale/shift of this sample: finalised once per launch by gn_coef_kernel (unet_misc.hip) ----
    if (p.gn_C > 0) {
        const float* cb = p.coef + (size_t)b * 2 * p.coef_stride;
        for (int c = tid; c < p.gn_C; c += 256) { s_sc[c] = cb[c]; s_sh[c] = cb[p.coef_stride + c]; }
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int prow = l31 >> 4, pcol = l31 & 15;
    const int nbase = n0 + wn * NT * 32 + l31;      // this lane's output channel in N-tile 0

    // B fragment (one float4 = 4 consecutive k of one output channel) for k-step `s` of a chunk.
    // Out-of-range channels / k are clamped to valid addresses: such columns are masked in the
    // epilogue and such k meet zeros in the A operand.
    int nclamp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) nclamp[nt] = min(nbase + nt * 32, p.Cout - 1);
    auto load_b = [&](const ConvSeg& sg, int ch, int s, float4 (&dst)[NT]) {
        const int tap = s / KS, ks = s % KS;
        if constexpr (BM == 0) {
            // fragment-major repack: [chunk16][tap][kstep(2)][Cout][8]; for a 1-tap segment that is simply [kstep][Cout][8]
            const size_t kidx = (KC == 16) ? (size_t)(ch * sg.taps + tap) * 2 + ks : (size_t)ch * KS + ks;
            const float* wp = sg.w + kidx * ((size_t)p.Cout * 8) + hi * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[nt] = *reinterpret_cast<const float4*>(wp + (size_t)nclamp[nt] * 8);
        } else {   // generic strided operand (attention): element (n, k) at w + b*w_bs + n*w_ns + k*w_ks
            const int k0 = ch * KC + ks * 8 + hi * 4;
            const float* wp = sg.w + (size_t)b * sg.w_bs;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float* wn_ = wp + (size_t)nclamp[nt] * sg.w_ns;
                dst[nt] = make_float4(wn_[(size_t)min(k0 + 0, sg.C - 1) * sg.w_ks], wn_[(size_t)min(k0 + 1, sg.C - 1) * sg.w_ks],
                                      wn_[(size_t)min(k0 + 2, sg.C - 1) * sg.w_ks], wn_[(size_t)min(k0 + 3, sg.C - 1) * sg.w_ks]);
            }
        }
    };

    int si = 0, ch = 0;
    while (true) {
        const ConvSeg& sg = p.seg[si];
        const bool first = (si == 0 && ch == 0);
        __syncthreads();          // every wave finished reading the previous chunk (and s_sc is written)
        store_lds(si, ch);
        __syncthreads();
        int nsi = si, nch = ch + 1;
        if (nch * KC >= sg.C) { nsi = si + 1; nch = 0; }
        const bool more = nsi < p.nseg;

        const int nsteps = sg.taps * KS;
        float4 b0[NT], b1[NT], b2[NT];
        if (nsteps > 0) { load_b(sg, ch, 0, b0); load_b(sg, ch, min(1, nsteps - 1), b1); }
        if (more) prefetch(nsi, nch);                  // issued after the first two B fragments: the
                                                        // in-order vmcnt wait for them does not drag these along
        // one k-step: 8 input channels of one tap.  `bc`/`ac` hold this step's B / A fragments, `bl`
        // receives the B fragments of step s+2 and `al` the A fragments of step s+1 (register rings
        // with static names: a rotating copy would make the compiler wait for what it just issued).
        auto load_a = [&](int s, float4 (&dst)[MT]) {
            const int tap = s / KS, ks = s % KS;
            const int ky = sg.taps == 9 ? tap / 3 : 1, kx = sg.taps == 9 ? tap % 3 : 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ppix = (((wm * MT + mt) * 2 + prow) * S + ky) * PW + pcol * S + kx;
                dst[mt] = *reinterpret_cast<const float4*>(s_patch + ppix * KCP + ks * 8 + hi * 4);
            }
        };
        auto k_step = [&](int s, float4 (&bc)[NT], float4 (&bl)[NT], float4 (&ac)[MT], float4 (&al)[MT]) {
            load_b(sg, ch, min(s + 2, nsteps - 1), bl);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of its use (hipcc sinks it otherwise)
            if (A_PREFETCH) load_a(min(s + 1, nsteps - 1), al); else load_a(s, ac);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt].x, bc[nt].x, acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt].y, bc[nt].y, acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt].z, bc[nt].z, acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt].w, bc[nt].w, acc[mt][nt], 0, 0, 0);
        };
        float4 a0[MT], a1[MT];
        if (A_PREFETCH && nsteps > 0) load_a(0, a0);
        // nsteps is 18 (3x3, KC 16), 2 (1 tap, KC 16) or 8 (1 tap, KC 64): groups of 6 + a tail of 2
        int s = 0;
        for (; s + 6 <= nsteps; s += 6) {
            k_step(s, b0, b2, a0, a1);     k_step(s + 1, b1, b0, a1, a0); k_step(s + 2, b2, b1, a0, a1);
            k_step(s + 3, b0, b2, a1, a0); k_step(s + 4, b1, b0, a0, a1); k_step(s + 5, b2, b1, a1, a0);
        }
        if (nsteps - s == 2) { k_step(s, b0, b2, a0, a1); k_step(s + 1, b1, b0, a1, a0); }
        if (!more) break;
        si = nsi; ch = nch;
    }

    // ---- epilogue: scale, bias(+temb), residual, store NHWC, per-channel statistics ----
    // The MFMA accumulator layout gives each lane 16 pixels of ONE channel (4-byte accesses).  Each 32x32
    // tile is transposed through a per-wave LDS scratch so that every lane owns 4 consecutive channels of
    // a pixel: residual loads and output stores are 16 B per lane, 1 KiB of contiguous NHWC rows per
    // wave instruction.
    __syncthreads();                                   // every wave is done reading the patch
    constexpr int TP = 36;                             // scratch row pitch in floats (32 + 4 pad)
    float* s_tr = s_patch + wave * (32 * TP);
    float* s_red = s_patch + 4 * 32 * TP;              // [WM][BN][2] behind the 4 scratch tiles
    const int cq = lane & 7;                           // this lane's channel quad inside a 32-channel tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ncol = n0 + (wn * NT + nt) * 32;
        const int n = ncol + l31;
        const float add = (p.addvec != nullptr && n < p.Cout) ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f;
        const int n4 = ncol + cq * 4;
        const bool nok4 = n4 < p.Cout;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                s_tr[row * TP + l31] = acc[mt][nt][r] * p.out_scale + add;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = (lane >> 3) + 8 * i;
                const int oy = oy0 + (wm * MT + mt) * 2 + (px >> 4), ox = ox0 + (px & 15);
                float4 v = *reinterpret_cast<const float4*>(s_tr + px * TP + cq * 4);
                if (nok4 && oy < p.H && ox < p.W) {
                    const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                    if (p.residual != nullptr) {
                        const float4 rv = *reinterpret_cast<const float4*>(p.residual + pix * p.res_cstride + n4);
                        const float rsc = p.res_scale;
                        v.x = fmaf(rv.x, rsc, v.x); v.y = fmaf(rv.y, rsc, v.y); v.z = fmaf(rv.z, rsc, v.z); v.w = fmaf(rv.w, rsc, v.w);
                    }
                    *reinterpret_cast<float4*>(p.out + pix * p.out_cstride + n4) = v;
                    s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                    s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (p.stats_out != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
            }
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = (wn * NT + nt) * 32 + cq * 4 + j;
                    s_red[(wm * BN + col) * 2] = s1[j]; s_red[(wm * BN + col) * 2 + 1] = s2[j];
                }
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

template <int MT, int NT, int WM, int WN, int S, int UP, int BM, int KC>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int KCP = KC + 4;
    constexpr int TH = 2 * MT * WM, TW = 16;
    constexpr int PP = ((TH - 1) * S + 3) * ((TW - 1) * S + 3);
    constexpr int BN = WN * NT * 32;
    constexpr int EPI = 4 * 32 * 36 + WM * BN * 2;
    const size_t lds = (size_t)((PP * KCP > EPI ? PP * KCP : EPI) + 2 * ((p.gn_C + 3) & ~3)) * sizeof(float);
    static unsigned long long attr_set = 0ull;
    auto kern = conv_mfma_kernel<MT, NT, WM, WN, S, UP, BM, KC>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    const int tiles = p.B * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    dim3 grid(tiles, (p.Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

static long wg_count(const ConvParams& p, int TH, int BN) {
    return (long)p.B * ((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * ((p.Cout + BN - 1) / BN);
}

template <int S, int UP, int BM, int KC>
static hipError_t launch_sel(const ConvParams& p, hipStream_t stream) {
    constexpr long MIN_WGS = 512;   // >= 2 workgroups per CU (a constant: finer tiles measured 12 % slower per unit of work, round 3)
    if (S == 1) {
        if (p.Cout <= 32) return launch_cfg<2, 1, 4, 1, S, UP, BM, KC>(p, stream);                                    // 16x16 px x 32
        if (p.Cout <= 64 && wg_count(p, 16, 64) >= MIN_WGS) return launch_cfg<2, 2, 4, 1, S, UP, BM, KC>(p, stream);  // 16x16 px x 64
        if (p.Cout > 64 && wg_count(p, 8, 128) >= MIN_WGS) return launch_cfg<2, 2, 2, 2, S, UP, BM, KC>(p, stream);   // 8x16 px x 128
        if (p.Cout > 64 && wg_count(p, 4, 128) >= MIN_WGS) return launch_cfg<1, 2, 2, 2, S, UP, BM, KC>(p, stream);   // 4x16 px x 128
        if (wg_count(p, 8, 64) >= MIN_WGS) return launch_cfg<1, 2, 4, 1, S, UP, BM, KC>(p, stream);                   // 8x16 px x 64
        return launch_cfg<1, 1, 2, 2, S, UP, BM, KC>(p, stream);                                                      // 4x16 px x 64
    } else {
        if (p.Cout <= 32) return launch_cfg<1, 1, 4, 1, S, UP, BM, KC>(p, stream);                                    // 8x16 px x 32
        if (p.Cout <= 64) return launch_cfg<1, 2, 4, 1, S, UP, BM, KC>(p, stream);                                    // 8x16 px x 64
        return launch_cfg<1, 1, 2, 2, S, UP, BM, KC>(p, stream);                                                      // 4x16 px x 64
    }
}

hipError_t launch_conv(const ConvParams& p, int stride, int up, hipStream_t stream) {
    bool generic = false;
    for (int i = 0; i < p.nseg; ++i) generic |= p.seg[i].w_mode != 0;
    if (generic) {
        for (int i = 0; i < p.nseg; ++i) if (p.seg[i].w_mode == 0) return hipErrorInvalidValue;   // not mixed
        if (stride != 1 || up) return hipErrorInvalidValue;
    }
    bool all_1tap = true;
    // 64-channel chunks: the generic operand clamps its k index, the packed weights end at C/16 slices and need whole chunks
    for (int i = 0; i < p.nseg; ++i) all_1tap &= p.seg[i].taps == 1 && (generic ? p.seg[i].C >= 64 : p.seg[i].C % 64 == 0);
    if (all_1tap && stride == 1 && !up) return generic ? launch_sel<1, 0, 1, 64>(p, stream) : launch_sel<1, 0, 0, 64>(p, stream);
    if (generic) return launch_sel<1, 0, 1, 16>(p, stream);
    if (stride == 2) return launch_sel<2, 0, 0, 16>(p, stream);
    if (up == 2) return launch_sel<1, 2, 0, 16>(p, stream);
    if (up) return launch_sel<1, 1, 0, 16>(p, stream);
    return launch_sel<1, 0, 0, 16>(p, stream);
}

size_t conv_flops(const ConvParams& p) {
    size_t k = 0;
    for (int i = 0; i < p.nseg; ++i) k += (size_t)p.seg[i].taps * p.seg[i].C;
    return 2 * (size_t)p.B * p.H * p.W * p.Cout * k;
}

}  // namespace pf
