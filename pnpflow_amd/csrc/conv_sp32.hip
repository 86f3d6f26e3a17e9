// Persistent software-pipelined implicit-GEMM 3x3 convolution of the full-resolution 32-channel level (Cout = 32: level 0), ONE wave per
// SIMD (256-thread workgroups, one per CU), split-fp16 MFMA, fp32-equivalent - the structure of conv_sp.hip where the floor is HBM and
// the vector ALU, not the matrix pipe.  Reference: the convolutions of ResidualBlock at level width 32 (pnpflow/models.py:58-113).
//   * at this level a 16-channel chunk of a 16 x 16-pixel tile is 54 MFMAs per wave (1.7 k matrix-pipe cycles) against ~300 staging
//     instructions (GroupNorm + SiLU + operand scale + fp16 hi / lo split of six float4 per lane): the two-team kernels (conv_pp.hip) run
//     the staging of one team beside the MFMAs of the other ON THE SAME SIMDs, where each slows the other (matrix pipe 0.29-0.41 busy,
//     waves parked 0.47 of the time: profiles/r04_pmc_conv_counters.md).  Here one wave issues both in one stream: MFMAs of chunk v,
//     staging of chunk v + 1 into the other patch buffer, requests of chunk v + 2;
//   * the weights of the whole launch (18 KiB per 16-channel 3x3 chunk: [tap][hi | lo][k-half][column][8 halfs]) are copied to LDS once;
//   * a chunk is walked in three tap ROWS: the fragments of a row (three taps: 12 A + 6 B reads) are requested under the 18 MFMAs of the
//     row before; ONE workgroup barrier per chunk, in front of the MFMAs of its last row (the next chunk's patch is complete, this
//     chunk's is free);
//   * the epilogue of a tile (bias, residual, lane transpose, stores, GroupNorm statistics) runs in the MFMA shadows of the NEXT tile:
//     at a tile's end the 32 accumulator registers are copied aside (one wave tile is 64 pixels x 32 channels), and the two halves of
//     the closed tile are finished under the last tap rows of the next tile's first two chunks.  With a serial epilogue every CU of
//     the chip writes its tile at the same moment (conv_sp.hip at the 128-channel level: ~10 k cycles per tile of store back-pressure).
// Launches: GroupNorm + SiLU 3x3 chunks only (an even number of 16-channel chunks per K-segment), with or without the identity residual.
#include <cstdlib>
#include "pp_common.h"

namespace pf {

constexpr int S32_PITCH = 20, S32_PW = 18, S32_TH = 16;
constexpr int S32_NPIX = (S32_TH + 2) * S32_PW;                  // 324
constexpr int S32_PATCH = (S32_TH + 2) * S32_PITCH * 64;         // 23 040 B
constexpr int S32_A9 = (S32_NPIX * 4 + 255) / 256;               // 6
constexpr int S32_CHUNK = 9 * 2048;                              // weight image of a 16-channel 3x3 chunk: 18 432 B
constexpr int s32_lds(int nch) { return nch * S32_CHUNK + 2 * S32_PATCH; }

#ifdef PP_PROBE_BUILD
__device__ unsigned long long* g_sp32_dbg = nullptr;
#define S32_STAMP(k) do { if (stamp_buf != nullptr && blockIdx.x == 0 && tid == 0 && stamp_n < 64) stamp_buf[stamp_n * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define S32_STAMP(k) do { } while (0)
#endif

template <bool RES>
__global__ __launch_bounds__(256, 1) void conv_sp32_kernel(const PPParams p) {
    constexpr int MT = 2, TH = S32_TH, NPIX = S32_NPIX, A9 = S32_A9, PATCH = S32_PATCH;
    static_assert(A9 == 6, "two float4 of the next chunk are staged per tap row");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const pp_float_cptr scale_c = (pp_float_cptr)(uintptr_t)p.scale;
    const int nch = p.n9;
    const unsigned patch0 = (unsigned)(nch * S32_CHUNK);

    // ---- weights -> LDS, once per workgroup --------------------------------------------------------------------------------------------
    for (int c = 0; c < nch; ++c) {
        const uint4* src = reinterpret_cast<const uint4*>(p.ch[c].wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem + c * S32_CHUNK);
        for (int i = tid; i < S32_CHUNK / 16; i += 256) dst[i] = src[i];
    }

    // ---- per-lane constants of the staging (conv_sp.hip) -----------------------------------------------------------------------------------
    const int qi = tid & 3, p0 = tid >> 2;
    unsigned pk[A9], ldsw[A9];
#pragma unroll
    for (int i = 0; i < A9; ++i) {
        const int pp = min(p0 + 64 * i, NPIX - 1);
        const int py = pp / S32_PW, px = pp - py * S32_PW;
        pk[i] = (unsigned)(py * p.W + px) | ((py == 0 ? 1u : 0u) << 20) | ((py == TH + 1 ? 1u : 0u) << 21) | ((px == 0 ? 1u : 0u) << 22) |
                ((px == S32_PW - 1 ? 1u : 0u) << 23);
        ldsw[i] = patch0 + (unsigned)((py * S32_PITCH + px) * 64) + (unsigned)((((qi >> 1) ^ ((px >> 2) & 3)) << 4) + (qi & 1) * 8);
    }
    const int pix_safe = p.W + 1;
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned a_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned base = patch0 + (unsigned)(((wq * 2 * MT + prow) * S32_PITCH + pcol + kx) * 64);
        const unsigned s = (unsigned)(((pcol + kx) >> 2) & 3);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a_addr[kx][tm] = base + ((((unsigned)(hi + 2 * tm)) ^ s) << 4);
    }
    const unsigned b_lane = (unsigned)lane * 16u;
    const bool bit3 = (lane & 8) != 0;
    const int em = lane & 7;
    const int ch_of_col = 4 * (l31 & 7) + (l31 >> 3);
    const unsigned e_lane = (unsigned)((4 * hi + ((lane >> 3) & 3)) * 128 + em * 16);

    // ---- this workgroup's tiles: XCD-contiguous ranges, a rotated start ------------------------------------------------------------------
    const int G = gridDim.x;
    const int rg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int T = p.B << (p.lx + p.ly);
    const int t_begin = (int)((long)rg * T / G), t_end = (int)((long)(rg + 1) * T / G);
    const int ntl = t_end - t_begin;
    const int rot = ntl > 0 ? (int)(((long)rg * p.rot) % ntl) : 0;
    auto tile_of = [&](int it) __attribute__((always_inline)) -> PPTile {
        const int idx = min(it, ntl - 1);
        const int wrapped = idx + rot >= ntl ? idx + rot - ntl : idx + rot;
        const int tl = t_begin + wrapped;
        PPTile r;
        const int tx = tl & ((1 << p.lx) - 1), ty = (tl >> p.lx) & ((1 << p.ly) - 1);
        r.b = tl >> (p.lx + p.ly); r.oy0 = ty * TH; r.ox0 = tx * 16;
        r.edge = (ty == 0 ? 1 : 0) | (ty == (1 << p.ly) - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == (1 << p.lx) - 1 ? 8 : 0);
        return r;
    };

    // ---- staging state (conv_sp.hip) -----------------------------------------------------------------------------------------------------
    float4 ra[A9];
    struct Coef { float4 csc, csh; float ascale; unsigned inval; };
    Coef cf0, cf1;
    struct Src { const char* base; unsigned cs4; };
    const unsigned q16 = (unsigned)qi * 16u;
    struct Desc { const char* base; const char* cb; unsigned cs4; int edge; float ascale; };
    auto describe = [&](const PPTile& tl, int c) __attribute__((always_inline)) -> Desc {
        const int pix = (tl.b * p.H + tl.oy0) * p.W + tl.ox0 - p.W - 1;      // (32-bit: conv_sp32_supported bounds B H W cstride below 2^31)
        const int cstride = p.ch[c].cstride;
        Desc d;
        d.ascale = p.scale != nullptr ? scale_c[8 * tl.b + p.ch[c].seg] : 1.0f;
        d.edge = tl.edge;
        d.cb = reinterpret_cast<const char*>(p.coef + (tl.b * 2 * p.coef_stride + p.ch[c].gn_c0));
        d.base = reinterpret_cast<const char*>(p.ch[c].src + (pix * cstride + p.ch[c].coff));
        d.cs4 = (unsigned)cstride * 4u;
        return d;
    };
    auto prep = [&](Coef& N, const Desc& d) __attribute__((always_inline)) -> Src {
        N.ascale = d.ascale;
        unsigned inval = 0;
#pragma unroll
        for (int i = 0; i < A9; ++i) inval |= (((pk[i] >> 20) & (unsigned)d.edge) != 0u ? 1u : 0u) << i;
        N.inval = inval;
        N.csc = *reinterpret_cast<const float4*>(d.cb + q16); N.csh = *reinterpret_cast<const float4*>(d.cb + (unsigned)(p.coef_stride * 4) + q16);
        Src r; r.base = d.base; r.cs4 = d.cs4;
        return r;
    };
    auto issue_one = [&](const Coef& N, const Src& sr, int i) __attribute__((always_inline)) {
        const unsigned px = ((N.inval >> i) & 1u) ? (unsigned)pix_safe : (pk[i] & 0xffffu);
        ra[i] = *reinterpret_cast<const float4*>(sr.base + (__umul24(px, sr.cs4) + q16));
    };
    // GroupNorm + SiLU + operand scale + fp16 hi / lo split of float4 i0 .. i0 + 2 (one tap row's share of the next chunk) -> patch at byte
    // offset pofs, written STAGE BY STAGE over the twelve elements: a wave alone on its SIMD issues in order, and element by element the
    // chain fma -> mul -> exp -> add -> rcp -> mul -> mul -> cvt -> cvt -> fma -> cvt (two elements wide as hipcc paired it) cost
    // ~1.3 k cycles per row for ~200 instructions.  Same arithmetic per element as conv_mfma16's staging (silu_pp; hi = RNE16(x), lo =
    // RNE16(x - hi)); every thread stores (the threads past the end of the patch hold the clamped last pixel and write ITS values).
    auto transform_row = [&](const Coef& S, int i0, unsigned pofs) __attribute__((always_inline)) {
        float x[12], e[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float4 r = ra[i0 + k / 4];
            const float rk = (k & 3) == 0 ? r.x : (k & 3) == 1 ? r.y : (k & 3) == 2 ? r.z : r.w;
            const float sc = (k & 3) == 0 ? S.csc.x : (k & 3) == 1 ? S.csc.y : (k & 3) == 2 ? S.csc.z : S.csc.w;
            const float sh = (k & 3) == 0 ? S.csh.x : (k & 3) == 1 ? S.csh.y : (k & 3) == 2 ? S.csh.z : S.csh.w;
            x[k] = rk * sc + sh;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = __expf(-x[k]);
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = 1.0f + e[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) e[k] = __builtin_amdgcn_rcpf(e[k]);
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] = x[k] * e[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] *= ((S.inval >> (i0 + k / 4)) & 1u) ? 0.0f : S.ascale;
        _Float16 h[12], l[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) h[k] = (_Float16)x[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) l[k] = (_Float16)(x[k] - (float)h[k]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned addr = ldsw[i0 + j] + pofs;
            *reinterpret_cast<f16x4*>(smem + addr) = f16x4{h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]};
            *reinterpret_cast<f16x4*>(smem + (addr ^ 32u)) = f16x4{l[4 * j], l[4 * j + 1], l[4 * j + 2], l[4 * j + 3]};
        }
    };
    auto transform_one = [&](const Coef& S, int i, unsigned pofs) __attribute__((always_inline)) {      // (prologue only)
        float4 v = ra[i];
        v.x = silu_pp(v.x * S.csc.x + S.csh.x); v.y = silu_pp(v.y * S.csc.y + S.csh.y);
        v.z = silu_pp(v.z * S.csc.z + S.csh.z); v.w = silu_pp(v.w * S.csc.w + S.csh.w);
        const float f = ((S.inval >> i) & 1u) ? 0.0f : S.ascale;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        f16x4 l = {(_Float16)(v.x - (float)h[0]), (_Float16)(v.y - (float)h[1]), (_Float16)(v.z - (float)h[2]), (_Float16)(v.w - (float)h[3])};
        const unsigned addr = ldsw[i] + pofs;
        *reinterpret_cast<f16x4*>(smem + addr) = h;
        *reinterpret_cast<f16x4*>(smem + (addr ^ 32u)) = l;
    };

    // accumulators of the tile being multiplied, and of the tile closed before it (its epilogue rides in this tile's MFMA shadows)
    f32x16 acc[MT], old[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mt][r] = 0.f; old[mt][r] = 0.f; }
    float run1[4], run2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { run1[j] = 0.f; run2[j] = 0.f; }
    int run_b = -1, run_n = 0;

    auto flush_stats = [&]() __attribute__((always_inline)) {
        if (p.stats_out == nullptr || run_b < 0) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = (double)run1[j], q = (double)run2[j];
            a += __shfl_xor(a, 8); q += __shfl_xor(q, 8);
            a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            if (lane < 8) {
                double* dst = p.stats_out + ((size_t)run_b * 32 + em * 4 + j) * 2;
                unsafeAtomicAdd(dst, a); unsafeAtomicAdd(dst + 1, q);
            }
            run1[j] = 0.f; run2[j] = 0.f;
        }
        run_n = 0;
    };

    // inputs of a tile's epilogue, fetched when the tile is opened and carried to where it is closed
    struct TileIn { float addv, oscale, rscale; size_t pix0; };
    auto tile_inputs = [&](const PPTile& tl) __attribute__((always_inline)) -> TileIn {
        TileIn t;
        t.addv = p.addvec != nullptr ? p.addvec[(size_t)tl.b * p.addvec_bs + ch_of_col] : 0.f;
        const float inv_last = p.scale != nullptr ? scale_c[8 * tl.b + 4 + p.ch[nch - 1].seg] : 1.0f;
        t.oscale = p.out_scale * (1.0f / 256.0f) * inv_last;
        t.rscale = p.res_scale;
        t.pix0 = ((size_t)tl.b * p.H + tl.oy0 + wq * 2 * MT) * p.W + tl.ox0;
        return t;
    };
    auto tile_offs = [&](int mt, int g) __attribute__((always_inline)) -> size_t {
        return (size_t)((mt * 2 + (g >> 1)) * p.W) * 128 + (size_t)((g & 1) * 1024) + e_lane;
    };
    // one M-tile (32 pixels x 32 channels) of a closed tile: bias, residual, lane transpose, streamed stores, statistics
    float4 rv[4];
    auto issue_rv = [&](const TileIn& t, int mt) __attribute__((always_inline)) {
        if constexpr (RES) {
            const char* rbase = reinterpret_cast<const char*>(p.residual + t.pix0 * 32);
#pragma unroll
            for (int g = 0; g < 4; ++g) rv[g] = nt_load4(rbase + tile_offs(mt, g));
        }
    };
    auto epi_piece = [&](const f32x16& a, const TileIn& t, int mt) __attribute__((always_inline)) {
        char* obase = reinterpret_cast<char*>(p.out + t.pix0 * 32);
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = a[r] * t.oscale + t.addv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            oct_transpose(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3], bit3);
            float4 v = make_float4(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
            if constexpr (RES) {
                const float rsc = t.rscale; const float4 r4 = rv[g];
                v.x = fmaf(r4.x, rsc, v.x); v.y = fmaf(r4.y, rsc, v.y); v.z = fmaf(r4.z, rsc, v.z); v.w = fmaf(r4.w, rsc, v.w);
            }
            nt_store4(obase + tile_offs(mt, g), v);
            run1[0] += v.x; run1[1] += v.y; run1[2] += v.z; run1[3] += v.w;
            run2[0] += v.x * v.x; run2[1] += v.y * v.y; run2[2] += v.z * v.z; run2[3] += v.w * v.w;
        }
    };

    // ---- fragments of a tap ROW (three taps): two named sets ------------------------------------------------------------------------------
    f16x8 fa[2][3][MT][2], fb[2][3][2];
    auto load_row = [&](int set, int ky, unsigned pofs, unsigned wofs) __attribute__((always_inline)) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][kx][mt][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][1] + (unsigned)((mt * 2 + ky) * S32_PITCH * 64) + pofs);
            fb[set][kx][0] = *reinterpret_cast<const f16x8*>(smem + wofs + b_lane + (unsigned)((ky * 3 + kx) * 2048));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][kx][mt][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + (unsigned)((mt * 2 + ky) * S32_PITCH * 64) + pofs);
            fb[set][kx][1] = *reinterpret_cast<const f16x8*>(smem + wofs + b_lane + (unsigned)((ky * 3 + kx) * 2048 + 1024));
        }
    };
    auto mma_row = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kx][mt][1], fb[set][kx][0], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kx][mt][0], fb[set][kx][1], acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kx][mt][0], fb[set][kx][0], acc[mt], 0, 0, 0);
        }
        (void)0;
    };
    // the scheduler places a row's side work into the shadows of its 18 MFMAs: per MFMA one LDS read, then plain VALU / transcendentals /
    // SALU; LDS writes and requests towards the end
    auto row_pattern = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
            if (k >= 10) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (k >= 14) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (k >= 12) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
    };

    int stamp_n = 0; (void)stamp_n;
#ifdef PP_PROBE_BUILD
    unsigned long long* const stamp_buf = g_sp32_dbg;
#endif
    Desc dn; Src sr;
    TileIn told{}, tcur{}, tnext{};      // inputs of the tile whose epilogue is pending (old[]), of the current tile, of the next one
    // ---- one chunk: MFMAs of chunk (it, c) from patch SI; staging of the next chunk into patch SI ^ 1 under rows 0 and 1; under row 2 the
    // descriptor / coefficients of the chunk three ahead and, when EPI >= 0, M-tile EPI of the closed tile.  On entry the fragments of
    // row 0 are in set SI (three rows per chunk: the set of row r is (r + SI) & 1), ra[] / C belong to the next chunk, N / sr to the
    // chunk two ahead.
    auto chunk = [&](int it, int c, auto SI_, auto EPI_) __attribute__((always_inline)) {
        constexpr int SI = decltype(SI_)::value, EPI = decltype(EPI_)::value;
        constexpr unsigned PCUR = SI ? PATCH : 0, PNXT = SI ? 0 : PATCH;
        Coef& C = SI == 0 ? cf1 : cf0;
        Coef& N = SI == 0 ? cf0 : cf1;
        const unsigned wofs = (unsigned)(c * S32_CHUNK);
        const int c1 = c + 1 == nch ? 0 : c + 1;
        const unsigned wofs1 = (unsigned)(c1 * S32_CHUNK);
        S32_STAMP(0);
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            __builtin_amdgcn_sched_barrier(0);
            load_row((row + 1 + SI) & 1, row + 1, PCUR, wofs);
#if !defined(S32_ABL) || S32_ABL != 1
            transform_row(C, 3 * row, PNXT);
#endif
#pragma unroll
            for (int i = 3 * row; i < 3 * row + 3; ++i) issue_one(N, sr, i);
            if (EPI >= 0 && row == 1) issue_rv(told, EPI);
#if !defined(S32_ABL) || S32_ABL != 2
            mma_row((row + SI) & 1);
#endif
            row_pattern();
            __builtin_amdgcn_sched_barrier(0);
            S32_STAMP(1 + row);
        }
        // row 2: its fragments are in registers -> this chunk's patch is free; the next chunk's patch must be complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        S32_STAMP(3);
        load_row((3 + SI) & 1, 0, PNXT, wofs1);
        {
            int q = c + 3, wrap = 0;                                 // (two chunks per tile: three ahead may be two tiles ahead)
            if (q >= nch) { q -= nch; wrap = 1; }
            if (q >= nch) { q -= nch; wrap = 2; }
            dn = describe(tile_of(it + wrap), q);
            sr = prep(C, dn);      // (C: every float4 of the next chunk has been staged; it becomes N of the next chunk)
        }
        if constexpr (EPI >= 0) epi_piece(old[EPI], told, EPI);
        if constexpr (EPI == 1) tnext = tile_inputs(tile_of(it + 1));      // (the next tile's bias / output scale: under this row, not at the tile's end)
#if !defined(S32_ABL) || S32_ABL != 2
        mma_row((2 + SI) & 1);
#endif
        row_pattern();
        __builtin_amdgcn_sched_barrier(0);
        S32_STAMP(4);
        ++stamp_n;
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------------------------
    if (ntl <= 0) return;
    int seg_c1 = -1, seg_c2 = -1;
    for (int c = 1; c < nch; ++c)
        if (p.ch[c].seg != p.ch[c - 1].seg) { if (seg_c1 < 0) seg_c1 = c; else seg_c2 = c; }
    {
        const Src s0 = prep(cf0, describe(tile_of(0), 0));
#pragma unroll
        for (int i = 0; i < A9; ++i) issue_one(cf0, s0, i);
#pragma unroll
        for (int i = 0; i < A9; ++i) transform_one(cf0, i, 0);
        const int w1 = 1 >= nch ? 1 : 0;
        const Src s1 = prep(cf1, describe(tile_of(w1), 1 - (w1 ? nch : 0)));
#pragma unroll
        for (int i = 0; i < A9; ++i) issue_one(cf1, s1, i);
        const int w2 = 2 >= nch ? 1 : 0;
        sr = prep(cf0, describe(tile_of(w2), 2 - (w2 ? nch : 0)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_row(0, 0, 0, 0);
    }
    // the epilogue that rides in the first tile closes "tile 0" with zeros (scale, bias and residual scale 0: the stores are overwritten
    // by the tile's real epilogue, the statistics receive zeros) - no branch in the chunk bodies
    told = tile_inputs(tile_of(0)); told.addv = 0.f; told.oscale = 0.f; told.rscale = 0.f;
    run_b = tile_of(0).b;
    const int seg_b1 = seg_c1 > 0 ? seg_c1 : nch, seg_b2 = seg_c2 > 0 ? seg_c2 : nch;
    tcur = tile_inputs(tile_of(0));
#pragma unroll 1
    for (int it = 0; it < ntl; ++it) {
        const PPTile tl = tile_of(it);
        // the first chunk pair carries the epilogue of the tile closed before
        chunk(it, 0, ic<0>{}, ic<0>{});
        chunk(it, 1, ic<1>{}, ic<1>{});
#pragma unroll 1
        for (int sg = 0; sg < 3; ++sg) {
            const int lo = sg == 0 ? 2 : sg == 1 ? seg_b1 : seg_b2, hi_c = sg == 0 ? seg_b1 : sg == 1 ? seg_b2 : nch;
            if (lo >= hi_c) continue;
            if (sg > 0 && p.scale != nullptr) {
                const float ratio = scale_c[8 * tl.b + p.ch[lo].seg] * scale_c[8 * tl.b + 4 + p.ch[lo - 1].seg];
                if (ratio != 1.0f) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][r] *= ratio;
                }
            }
#pragma unroll 1
            for (int c = lo; c < hi_c; c += 2) {
                chunk(it, c, ic<0>{}, ic<-1>{});
                chunk(it, c + 1, ic<1>{}, ic<-1>{});
            }
        }
        // the tile is complete: its accumulators step aside, its epilogue rides in the next tile (or behind the loop)
        if (tl.b != run_b || run_n >= 32) { flush_stats(); run_b = tl.b; }
        ++run_n;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { old[mt][r] = acc[mt][r]; acc[mt][r] = 0.f; }
        told = tcur; tcur = tnext;
    }
    issue_rv(told, 0); epi_piece(old[0], told, 0);
    issue_rv(told, 1); epi_piece(old[1], told, 1);
    flush_stats();
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------

static int ilog2_exact_s32(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

bool conv_sp32_supported(const ConvParams& p, int stride, int up, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_SP32") ? atoi(getenv("PNPFLOW_HIP_SP32")) : 0;      // OFF by default: parity-green, but at this level it does not beat conv_pp yet; PNPFLOW_HIP_SP32=1 selects it
    if (mode == 0 || terms != 3 || stride != 1 || up != 0 || p.gnb_x != nullptr) return false;
    if (p.Cout != 32 || p.out_cstride != 32 || (p.residual != nullptr && p.res_cstride != 32)) return false;
    if (p.H % S32_TH || p.W % 16 || p.Hs != p.H || p.Ws != p.W || p.W > 2048) return false;
    if (ilog2_exact_s32(p.H / S32_TH) < 0 || ilog2_exact_s32(p.W / 16) < 0) return false;
    const int grid = persistent_grid();
    if (grid < 8 || (long)p.B * (p.H / S32_TH) * (p.W / 16) < 16L * grid) return false;
    int nch = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if ((double)p.B * p.H * p.W * s.cstride >= 2147483648.0) return false;      // 32-bit element offsets (describe)
        if (s.w_mode != 0 || s.w16 == nullptr || s.C % 32 || s.taps != 9 || s.xform != 2) return false;
        nch += s.C / 16;
    }
    if (nch < 2 || nch > 6) return false;
    if (p.stats_out == nullptr) return false;      // (the dummy epilogue of the first tile relies on the per-image bookkeeping)
    return p.gn_C > 0 && p.coef != nullptr && s32_lds(nch) <= 160 * 1024;
}

hipError_t launch_conv_sp32(const PPParams& p0, hipStream_t s) {
    if (p0.n9 < 2 || p0.n9 > 6 || (p0.n9 & 1) || p0.n1 != 0 || p0.cout != 32) return hipErrorInvalidValue;
    static unsigned long long attr_set[2] = {0ull, 0ull};
    const bool res = p0.residual != nullptr;
    const void* kern = res ? reinterpret_cast<const void*>(conv_sp32_kernel<true>) : reinterpret_cast<const void*>(conv_sp32_kernel<false>);
    { hipError_t e = set_max_dynamic_lds_once(kern, attr_set[res ? 1 : 0], 160 * 1024); if (e != hipSuccess) return e; }
    const int grid = persistent_grid();
    if (grid <= 0) return hipErrorInvalidConfiguration;
    PPParams p = p0;
    p.lx = ilog2_exact_s32(p.W / 16); p.ly = ilog2_exact_s32(p.H / S32_TH);
    p.rot = 5;
    if (res) hipLaunchKernelGGL(conv_sp32_kernel<true>, dim3(grid), dim3(256), s32_lds(p.n9), s, p);
    else hipLaunchKernelGGL(conv_sp32_kernel<false>, dim3(grid), dim3(256), s32_lds(p.n9), s, p);
    return hipGetLastError();
}

}  // namespace pf
