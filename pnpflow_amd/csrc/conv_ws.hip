// Wave-specialised, persistent implicit-GEMM convolution on the gfx950 16-bit matrix cores (fp32-equivalent split-fp16
// operands: see conv_mfma16.hip for the arithmetic; same GEMM view, same fusion, same results).
//
// Why a second kernel.  conv_mfma16_kernel runs every phase of a workgroup in lockstep - GroupNorm prologue, global ->
// GN/SiLU -> split -> LDS staging, k-loop, epilogue - and relies on 2-3 co-resident workgroups per CU to overlap them
// (profiles/r02_phase_trace_baseline.txt: the k-loop is 38-76 % of a workgroup's life, the matrix pipe is busy ~30-40 %).
// Here the phases are assigned to different waves of ONE persistent workgroup per CU and overlap by construction:
//
//   waves 4-7  PRODUCERS  global loads of the halo patch two K-chunks ahead (two register sets), GroupNorm scale/shift +
//                         SiLU + power-of-two operand scale + fp16 hi/lo split, ds_write into a double-buffered LDS patch
//   waves 0-3  CONSUMERS  A fragments from LDS, B fragments (pre-split weights) straight from L2 into a register ring,
//                         3 x v_mfma_f32_32x32x16_f16 per product, epilogue (bias/temb/residual, NHWC float4 stores,
//                         GroupNorm statistics of the output)
//
// One s_barrier per K-chunk ("step") is the only synchronisation: in step s the producers write chunk s into patch buffer
// s&1 while the consumers contract chunk s-1 from buffer (s-1)&1.  The workgroup walks its tiles back to back (tile t of
// workgroup w = w + k*gridDim.x, mapped so that the N-blocks of one pixel tile land on the same XCD), so the consumers'
// epilogue of tile j overlaps the producers' loads/staging of tile j+1, and nothing is re-derived per tile (the GroupNorm
// coefficients come from the per-launch gn_coef micro-kernel).  With two waves per SIMD each wave may use 256 registers:
// the consumer tile is MT x NT = 4 x 2 accumulators (one A fragment pair feeds 6 MFMAs, one B pair 12).
//
// Scope: stride 1, no resampling, packed split-fp16 weights, 3x3 main segments with optional folded 1x1 shortcut segments
// (every ResidualBlock conv of the U-Net); everything else stays on conv_mfma16_kernel / conv_mfma_kernel.
#include <cstdlib>
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_ws(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// position of a workgroup's walk over (tile, K-chunk)
struct WsCursor {
    int v;            // virtual tile index (see decode); >= VT: exhausted
    int b, oy0, ox0, n0;
    int si, ch;
};

template <int MT, int NT, int WM, int WN, int KC>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const ConvParams p, const int PT, const int NB) {
    constexpr int ROW = KC + 4;                  // dwords per LDS row of a pixel: KC/2 (hi halfs) + KC/2 (lo halfs) + 4 (pad)
    constexpr int KQ = KC / 4, KH = KC / 2, KS = KC / 16;
    constexpr int TH = 2 * MT * WM, TW = 16;
    constexpr int PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int BN = WN * NT * 32;
    constexpr int A_F4 = PP * KQ, A_PER = (A_F4 + 255) / 256;
    constexpr int TP = 36;                       // transpose scratch pitch (floats)
    static_assert(WM * WN == 4, "4 consumer waves");
    static_assert(A_PER <= 16, "validity mask");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* const s_patch = reinterpret_cast<uint32_t*>(smem_raw);                  // [2][PP][ROW]
    float* const s_trs = reinterpret_cast<float*>(s_patch + 2 * PP * ROW);            // [4][32][TP]   per consumer wave
    float* const s_redb = s_trs + 4 * 32 * TP;                                        // [2][WM][BN][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int G = gridDim.x;
    const int VT = ((PT + 7) / 8) * 8 * NB;      // virtual tiles: v -> (pixel tile, N block); pt >= PT are holes
    int NCH = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < p.nseg) NCH += (p.seg[i].C + KC - 1) / KC;

    auto decode = [&](WsCursor& c) {             // c.v -> tile coordinates; returns false for a hole
        const int grp = c.v / (8 * NB), r = c.v % (8 * NB);
        const int nb = r / 8, pt = grp * 8 + (r % 8);
        if (pt >= PT) return false;
        int q = pt;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        c.b = q / tiles_y; c.oy0 = ty * TH; c.ox0 = tx * TW; c.n0 = nb * BN;
        return true;
    };
    auto first_tile = [&](WsCursor& c) {
        c.v = blockIdx.x; c.si = 0; c.ch = 0;
        while (c.v < VT && !decode(c)) c.v += G;
    };
    auto advance = [&](WsCursor& c) {            // next K-chunk; at the end of a tile, the workgroup's next tile
        c.ch += 1;
        if (c.ch * KC >= p.seg[c.si].C) { c.si += 1; c.ch = 0; }
        if (c.si >= p.nseg) {
            c.si = 0; c.v += G;
            while (c.v < VT && !decode(c)) c.v += G;
        }
    };
    int ntile = 0;
    { WsCursor c; c.v = blockIdx.x; c.b = c.oy0 = c.ox0 = c.n0 = c.si = c.ch = 0; for (; c.v < VT; c.v += G) ntile += decode(c) ? 1 : 0; }
    const int N = ntile * NCH;                   // steps 0..N: N+1 barriers, the same count in every wave

    if (producer) {
        // ================================================ PRODUCERS ================================================
        const int ptid = tid - 256;
        const int qi = ptid % KQ, q4 = qi * 4;
        float4 ra0[A_PER], ra1[A_PER];           // two K-chunks in flight
        float4 sc0, sh0, sc1, sh1;
        int meta0 = 0, meta1 = 0;                // bits 0-15: pixel validity of the staged float4s, 16-17: xform, 18: channel quad inside the segment
        float as0 = 1.f, as1 = 1.f;              // operand scale of the chunk's segment
        WsCursor L; first_tile(L);               // load cursor

        auto load = [&](float4 (&ra)[A_PER], float4& sc, float4& sh, int& meta, float& as) {
            const ConvSeg& sg = p.seg[L.si];
            const int c = L.ch * KC + q4;
            const bool cok = c < sg.C;
            const float* base = sg.src + sg.coff + min(c, sg.C - 4);
            int m = (sg.xform << 16) | (cok ? (1 << 18) : 0);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int idx = ptid + i * 256;
                const int pix = idx / KQ, py = pix / PW, px = pix % PW;
                const int gy = L.oy0 - 1 + py, gx = L.ox0 - 1 + px;
                const bool ok = idx < A_F4 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                m |= ok ? (1 << i) : 0;
                const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);      // unconditional, clamped loads
                ra[i] = *reinterpret_cast<const float4*>(base + ((size_t)(L.b * p.H + cy) * p.W + cx) * sg.cstride);
            }
            sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.gn_C > 0) {                    // uniform
                const float* cb = p.coef + (size_t)L.b * 2 * p.coef_stride + min(sg.gn_off + c, p.gn_C - 4);
                const float4 a = *reinterpret_cast<const float4*>(cb), d = *reinterpret_cast<const float4*>(cb + p.coef_stride);
                if (sg.xform != 0 && cok) { sc = a; sh = d; }
            }
            as = p.scale != nullptr ? p.scale[8 * L.b + L.si] : 1.0f;
            meta = m;
            advance(L);
        };
        auto write = [&](const float4 (&ra)[A_PER], const float4 sc, const float4 sh, const int meta, const float as, uint32_t* buf) {
            const int xform = (meta >> 16) & 3;
            const bool cok = (meta >> 18) & 1;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                if (ptid + i * 256 < A_F4) {
                    const int a_lds = ((ptid + i * 256) / KQ) * ROW + qi * 2;      // dword offset of this thread's 4 hi halfs (lo at +KH)
                    float4 v = ra[i];
                    if (xform != 0) {
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
                        v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        if (xform == 2) { v.x = silu_ws(v.x); v.y = silu_ws(v.y); v.z = silu_ws(v.z); v.w = silu_ws(v.w); }
                    }
                    v.x *= as; v.y *= as; v.z *= as; v.w *= as;
                    if (!(cok && ((meta >> i) & 1))) v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies AFTER norm + activation
                    // opaque: the hi that is stored and the hi that is subtracted must be the same rounding of the same value
                    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                    f16x4 h, l;
                    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
                    l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
                    l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
                    *reinterpret_cast<f16x4*>(buf + a_lds) = h;
                    *reinterpret_cast<f16x4*>(buf + a_lds + KH) = l;
                }
            }
        };

        if (N > 0) load(ra0, sc0, sh0, meta0, as0);
        if (N > 1) load(ra1, sc1, sh1, meta1, as1);
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s <= N; s += 2) {
            if (s < N) {
                write(ra0, sc0, sh0, meta0, as0, s_patch);
                if (s + 2 < N) load(ra0, sc0, sh0, meta0, as0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (s + 1 <= N) {
                if (s + 1 < N) {
                    write(ra1, sc1, sh1, meta1, as1, s_patch + PP * ROW);
                    if (s + 3 < N) load(ra1, sc1, sh1, meta1, as1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
        }
        return;
    }

    // ==================================================== CONSUMERS ====================================================
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int prow = l31 >> 4, pcol = l31 & 15;
    const int cq = lane & 7;
    float* const s_tr = s_trs + wave * (32 * TP);

    f32x16 acc[MT][NT];
    WsCursor Cc; first_tile(Cc);                 // chunk being contracted
    WsCursor Cn = Cc;                            // the chunk after it (B-fragment look-ahead)
    int nclamp[NT], nclamp_n[NT];
    auto set_nclamp = [&](const WsCursor& c, int (&nc)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) nc[nt] = min(c.n0 + wn * NT * 32 + l31 + nt * 32, p.Cout - 1);
    };
    // B fragments of k16-step s of chunk (si, ch): 8 hi + 8 lo halfs per lane (clamped channel index: unconditional loads)
    auto load_b = [&](int si, int ch, int s, const int (&nc)[NT], uint4 (&dh)[NT], uint4 (&dl)[NT]) {
        const ConvSeg& sg = p.seg[si];
        const int tap = s / KS, j = s % KS;
        const size_t kidx = ((size_t)ch * KS + j) * sg.taps + tap;
        const uint4* wp = reinterpret_cast<const uint4*>(sg.w16) + kidx * ((size_t)p.Cout * 4) + hi;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { dh[nt] = wp[(size_t)nc[nt] * 2]; dl[nt] = wp[(size_t)p.Cout * 2 + (size_t)nc[nt] * 2]; }      // [hi | lo][Cout][16] blocks (round 3 repack)
    };
    // B register ring: RD - 1 k-steps ahead of use.  4 x 2 accumulator tiles leave room for a 2-deep ring only (one k-step
    // there is 24 MFMAs = 768 matrix-pipe cycles, longer than an L2 round trip)
    constexpr int RD = (MT * NT > 4) ? 2 : 3;
    uint4 bh0[NT], bl0[NT], bh1[NT], bl1[NT], bh2[RD == 3 ? NT : 1], bl2[RD == 3 ? NT : 1];
    if (N > 0) {
        set_nclamp(Cc, nclamp);
        const int ns0 = p.seg[Cc.si].taps * KS;
        load_b(Cc.si, Cc.ch, 0, nclamp, bh0, bl0);
        if constexpr (RD == 3) load_b(Cc.si, Cc.ch, min(1, ns0 - 1), nclamp, bh1, bl1);
        advance(Cn);
    }
    int pend_par = -1, pend_b = 0, pend_n0 = 0, tile_no = 0;      // statistics of the previous tile, waiting in s_redb[pend_par]

    auto flush_stats = [&]() {
        if (pend_par >= 0 && p.stats_out != nullptr) {
            const float* sr = s_redb + pend_par * (WM * BN * 2);
            for (int t = tid; t < BN * 2; t += 256) {
                const int col = t >> 1, which = t & 1;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) tot += sr[(w * BN + col) * 2 + which];
                const int n = pend_n0 + col;
                if (n < p.Cout) unsafeAtomicAdd(p.stats_out + ((size_t)pend_b * p.Cout + n) * 2 + which, (double)tot);
            }
        }
        pend_par = -1;
    };

    auto epilogue = [&](const WsCursor& c) {
        // each 32x32 accumulator tile is transposed through the wave's LDS scratch so that a lane owns 4 consecutive channels
        // of a pixel: residual loads and output stores are 16 B per lane, 1 KiB of contiguous NHWC rows per wave instruction
        const int b = c.b, oy0 = c.oy0, ox0 = c.ox0, n0 = c.n0;
        const int last = p.nseg - 1;
        const float oscale = p.out_scale * (1.0f / 256.0f) * (p.scale != nullptr ? p.scale[8 * b + 4 + last] : 1.0f);
        float* s_red = s_redb + (tile_no & 1) * (WM * BN * 2);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ncol = n0 + (wn * NT + nt) * 32;
            const int n = ncol + l31;
            const float add = (p.addvec != nullptr && n < p.Cout) ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f;
            const int n4 = ncol + cq * 4;
            const bool nok4 = n4 < p.Cout;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            constexpr int RG = (MT * NT > 4) ? 1 : 2;    // residual values are requested for RG M-tiles at a time, ahead of their stores
            float4 rv[RG][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt % RG == 0 && p.residual != nullptr) {
#pragma unroll
                    for (int g = 0; g < RG; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int px = (lane >> 3) + 8 * i;
                            const int oy = min(oy0 + (wm * MT + mt + g) * 2 + (px >> 4), p.H - 1), ox = min(ox0 + (px & 15), p.W - 1);
                            const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                            rv[g][i] = *reinterpret_cast<const float4*>(p.residual + pix * p.res_cstride + min(n4, p.Cout - 4));
                        }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s_tr[row * TP + l31] = acc[mt][nt][r] * oscale + add;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = (lane >> 3) + 8 * i;    // pixel of the 32-pixel M-tile (2 rows x 16 cols)
                    const int oy = oy0 + (wm * MT + mt) * 2 + (px >> 4), ox = ox0 + (px & 15);
                    float4 v = *reinterpret_cast<const float4*>(s_tr + px * TP + cq * 4);
                    if (nok4 && oy < p.H && ox < p.W) {
                        const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                        if (p.residual != nullptr) {
                            const float rsc = p.res_scale;
                            v.x = fmaf(rv[mt % RG][i].x, rsc, v.x); v.y = fmaf(rv[mt % RG][i].y, rsc, v.y); v.z = fmaf(rv[mt % RG][i].z, rsc, v.z); v.w = fmaf(rv[mt % RG][i].w, rsc, v.w);
                        }
                        *reinterpret_cast<float4*>(p.out + pix * p.out_cstride + n4) = v;
                        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                        s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();            // the scratch is rewritten by the next tile
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
        pend_par = tile_no & 1; pend_b = b; pend_n0 = n0;      // reduced over the M-waves + added to the global statistics after the next barrier
        tile_no += 1;
    };

    // one step of the consumers: contract chunk Cc from `buf`; prefetch the first two B fragments of chunk Cn before the barrier
    auto compute = [&](const uint32_t* buf) {
        const ConvSeg& sg = p.seg[Cc.si];
        if (Cc.si == 0 && Cc.ch == 0) {
            flush_stats();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        } else if (Cc.ch == 0 && p.scale != nullptr) {
            // the accumulator changes units: from the previous segment's operand scale to this one's (powers of two: exact)
            const float ratio = p.scale[8 * Cc.b + Cc.si] * p.scale[8 * Cc.b + 4 + Cc.si - 1];
            if (ratio != 1.0f) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= ratio;
            }
        }
        const int nsteps = sg.taps * KS;
        auto k_step = [&](int s, uint4 (&ch_)[NT], uint4 (&cl_)[NT], uint4 (&nh_)[NT], uint4 (&nl_)[NT]) {
            load_b(Cc.si, Cc.ch, min(s + RD - 1, nsteps - 1), nclamp, nh_, nl_);
            __builtin_amdgcn_sched_barrier(0);
            const int tap = s / KS, j = s % KS;
            const int ky = sg.taps == 9 ? tap / 3 : 1, kx = sg.taps == 9 ? tap % 3 : 1;
            // M-tiles in groups of MG: the A fragments of a group are live only across its 3*MG*NT MFMAs (MT = 4, NT = 2 would
            // otherwise hold 32 registers of A fragments next to 128 accumulators and the B ring)
            constexpr int MG = (MT * NT > 4) ? 2 : MT;
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += MG) {
                f16x8 ah[MG], al[MG];
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    const int ppix = ((wm * MT + m0 + g) * 2 + prow + ky) * PW + pcol + kx;
                    ah[g] = *reinterpret_cast<const f16x8*>(buf + ppix * ROW + j * 8 + hi * 4);
                    al[g] = *reinterpret_cast<const f16x8*>(buf + ppix * ROW + KH + j * 8 + hi * 4);
                }
#pragma unroll
                for (int g = 0; g < MG; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m0 + g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], *reinterpret_cast<const f16x8*>(&ch_[nt]), acc[m0 + g][nt], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < MG; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m0 + g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], *reinterpret_cast<const f16x8*>(&cl_[nt]), acc[m0 + g][nt], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < MG; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m0 + g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], *reinterpret_cast<const f16x8*>(&ch_[nt]), acc[m0 + g][nt], 0, 0, 0);
            }
        };
        __builtin_amdgcn_s_setprio(1);
        int s = 0;
        if constexpr (RD == 3) {
            for (; s + 3 <= nsteps; s += 3) {
                k_step(s, bh0, bl0, bh2, bl2); k_step(s + 1, bh1, bl1, bh0, bl0); k_step(s + 2, bh2, bl2, bh1, bl1);
            }
            if (nsteps - s >= 1) k_step(s, bh0, bl0, bh2, bl2);
            if (nsteps - s == 2) k_step(s + 1, bh1, bl1, bh0, bl0);
        } else {
            for (; s + 2 <= nsteps; s += 2) { k_step(s, bh0, bl0, bh1, bl1); k_step(s + 1, bh1, bl1, bh0, bl0); }
            if (nsteps - s == 1) k_step(s, bh0, bl0, bh1, bl1);
        }
        __builtin_amdgcn_s_setprio(0);
        const bool tile_done = (Cn.v != Cc.v) || (Cn.si == 0 && Cn.ch == 0);      // Cn is one chunk ahead: it has wrapped
        // the first two B fragments of the next chunk go in flight before the epilogue / the barrier
        if (Cn.v < VT) {
            if (tile_done) set_nclamp(Cn, nclamp_n);
            const int ns1 = p.seg[Cn.si].taps * KS;
            load_b(Cn.si, Cn.ch, 0, tile_done ? nclamp_n : nclamp, bh0, bl0);
            if constexpr (RD == 3) load_b(Cn.si, Cn.ch, min(1, ns1 - 1), tile_done ? nclamp_n : nclamp, bh1, bl1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tile_done) {
            epilogue(Cc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) nclamp[nt] = nclamp_n[nt];
        }
        Cc = Cn;
        if (Cn.v < VT) advance(Cn);
    };

    for (int s = 0; s <= N; ++s) {
        if (s >= 1) compute(s_patch + ((s - 1) & 1) * (PP * ROW));      // chunk s-1 lives in buffer (s-1)&1
        __syncthreads();
    }
    flush_stats();
}

// --------------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int KC>
static hipError_t launch_ws_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int ROW = KC + 4, TH = 2 * MT * WM, PP = (TH + 2) * 18, BN = WN * NT * 32;
    const size_t lds = ((size_t)2 * PP * ROW + 4 * 32 * 36 + 2 * WM * BN * 2) * 4;
    static unsigned long long attr_set = 0ull;
    auto kern = conv_ws_kernel<MT, NT, WM, WN, KC>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    const int PT = p.B * ((p.H + TH - 1) / TH) * ((p.W + 15) / 16), NB = (p.Cout + BN - 1) / BN;
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    const long tiles = (long)PT * NB;
    const int grid = (int)(tiles < cus ? tiles : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p, PT, NB);
    return hipGetLastError();
}

// tiles of a (TH x 16 pixels) x BN channels decomposition, and how evenly they fill `cus` persistent workgroups
static double ws_fill(const ConvParams& p, int TH, int BN, int cus, long* tiles_out) {
    const long tiles = (long)p.B * ((p.H + TH - 1) / TH) * ((p.W + 15) / 16) * ((p.Cout + BN - 1) / BN);
    *tiles_out = tiles;
    const long rounds = (tiles + cus - 1) / cus;
    return (double)tiles / (double)(rounds * cus);
}

bool conv_ws_supported(const ConvParams& p, int stride, int up) {
    // opt-in (PNPFLOW_HIP_WS=1): measured slower than conv_mfma16_kernel on every layer class of the BASELINE nets in round 2
    // (profiles/r02_ws_vs_lockstep_layers.md) - one consumer wave per SIMD has nothing to cover its epilogue / L2 round trips
    static const bool enabled = getenv("PNPFLOW_HIP_WS") && atoi(getenv("PNPFLOW_HIP_WS")) != 0;
    if (!enabled || stride != 1 || up != 0 || p.nseg < 1 || p.seg[0].taps != 9 || p.gnb_x != nullptr) return false;
    if (p.gn_C > 0 && (p.coef == nullptr || p.gn_C % 4 != 0)) return false;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if (s.w_mode != 0 || s.w16 == nullptr || (s.taps != 9 && s.taps != 1) || s.C % 16 != 0 || s.C < 16 || s.cstride % 4 != 0 || s.coff % 4 != 0) return false;
        if (s.xform != 0 && s.gn_off % 4 != 0) return false;
    }
    if (p.Cout % 4 != 0 || p.out_cstride % 4 != 0 || (p.residual != nullptr && p.res_cstride % 4 != 0)) return false;
    if (p.H < 4 || p.W < 4) return false;
    return true;
}

hipError_t launch_conv_ws(const ConvParams& p, hipStream_t stream) {
    static const int force = getenv("PNPFLOW_HIP_WS_CFG") ? atoi(getenv("PNPFLOW_HIP_WS_CFG")) : 0;     // profiling: 1..5 selects a tile shape
    const int cus = 256;
    bool all32 = true;
    for (int i = 0; i < p.nseg; ++i) all32 &= p.seg[i].C % 32 == 0;
    long t;
    if (p.Cout <= 32 || force == 1) return launch_ws_cfg<4, 1, 4, 1, 16>(p, stream);                     // 32x16 px x 32
    if ((p.Cout <= 64 && force == 0) || force == 2) {
        if (ws_fill(p, 32, 64, cus, &t) >= 0.8 || t >= 8 * cus || force == 2) return launch_ws_cfg<4, 2, 4, 1, 16>(p, stream);   // 32x16 px x 64
        return all32 ? launch_ws_cfg<2, 1, 2, 2, 32>(p, stream) : launch_ws_cfg<4, 1, 4, 1, 16>(p, stream);
    }
    if (!all32) return launch_ws_cfg<4, 2, 4, 1, 16>(p, stream);
    // >= 128 output channels: the largest tile that still fills the persistent workgroups evenly
    if (force == 3 || (force == 0 && (ws_fill(p, 16, 128, cus, &t) >= 0.85 || t >= 8 * cus))) return launch_ws_cfg<4, 2, 2, 2, 32>(p, stream);   // 16x16 px x 128
    if (force == 4 || (force == 0 && (ws_fill(p, 8, 128, cus, &t) >= 0.85 || t >= 8 * cus))) return launch_ws_cfg<2, 2, 2, 2, 32>(p, stream);    // 8x16 px x 128
    return launch_ws_cfg<2, 1, 2, 2, 32>(p, stream);                                                     // 8x16 px x 64
}

}  // namespace pf
