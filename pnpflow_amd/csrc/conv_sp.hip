// Persistent software-pipelined implicit-GEMM 3x3 convolution, ONE wave per SIMD (256-thread workgroups, one per CU, 512 registers per
// lane), split-fp16 MFMA, fp32-equivalent.  Reference: the convolutions of ResidualBlock (pnpflow/models.py:58-113).
//
// Why another structure.  The two-team kernels of these levels (conv_pp64.hip in round 4, conv_pp128.hip early in round 5: retired, see
// docs/HISTORY.md) put two teams of four waves on a CU and alternate a VALU phase (staging: GroupNorm + SiLU + fp16 hi / lo split of the
// next patch) of one team with the MFMA phase of the other.  The two waves of a SIMD share its VALU
// issue: every staging instruction of the partner costs the MFMA wave ~3 cycles (MI355X_MICROARCH.md "Two waves per SIMD"; round-5
// stamps of conv_pp128: 4.5-5.6 k cycles per MFMA phase of 3.46 k matrix-pipe cycles, whoever issues the LDS-DMA refills), and every
// phase boundary is a workgroup barrier.  A wave that is ALONE on its SIMD hides up to five single-issue instructions in the 32-cycle
// shadow of each of its own MFMAs for nothing.  So here a wave does everything itself, in one instruction stream per 16-channel chunk:
//   * MFMAs of chunk v from LDS fragments (A: an XOR-swizzled fp16 hi / lo patch; B: the chunk's weight image, lane-linear);
//   * the staging of chunk v + 1 (from raw fp32 registers requested a chunk earlier) into the OTHER patch buffer, and the requests of
//     chunk v + 2 into the registers it frees;
//   * the LDS-DMA refill of the weights, half a chunk ahead: taps 0..4 of a chunk live in slot X (40 KiB), taps 5..8 in slot Y (32 KiB);
//     while X is read Y is refilled and vice versa - two workgroup barriers per chunk, each placed BEFORE the MFMAs of the last tap of a
//     half (its fragments are in registers by then), so that the first fragments of the next half are requested under those MFMAs;
//   * with 512 registers a wave owns MT x NT accumulator tiles of 32 x 32 = 128 accumulator registers (64 pixels x 128 channels at
//     Cout = 128): 12 LDS fragment reads per 24 MFMAs.
// Workgroup tile: 8 MT rows x 16 pixels x 32 NT channels (wave w: rows 2 MT w .. 2 MT w + 2 MT - 1).  LDS: X | Y | patch 0 | patch 1.
// No scalar-memory loads inside the chunk loop: an s_load in flight turns every LDS wait hipcc inserts into lgkmcnt(0) (scalar loads return
// out of order), i.e. a wave waits for the fragment reads it has just issued for the NEXT tap before it may start the current one
// (round-5 stamps: +500 cycles under tap 6, +200 under taps 4 / 8 of 770).  The per-chunk descriptors (source, weight image, channel
// stride / offset, GroupNorm offset, K-segment) therefore sit in a 32-byte-per-chunk LDS table written once per workgroup, and the
// operand scale of a chunk is fetched with the chunk's GroupNorm coefficients as a vector load.
// vmcnt is counted by hand around the LDS-DMA pieces (memory operations return in order): pieces are issued right behind a barrier,
// BEFORE the three patch requests of that half-chunk, and waited for with vmcnt(3) in front of the next barrier.
#include <cstdlib>
#include "pp_common.h"

namespace pf {

constexpr int SP_PITCH = 20, SP_PW = 18;
constexpr int sp_rows(int MT) { return 8 * MT; }                                   // tile rows
constexpr int sp_npix(int MT) { return (sp_rows(MT) + 2) * SP_PW; }
constexpr int sp_patch(int MT) { return (sp_rows(MT) + 2) * SP_PITCH * 64; }      // MT 2: 23 040 B
constexpr int sp_a9(int MT) { return (sp_npix(MT) * 4 + 255) / 256; }             // float4 per lane and chunk (MT 2: 6)
// TERMS = 3: the fp32-equivalent split; TERMS = 1 (round 6): precision mode 2 - hi-only operands, one MFMA per (M-tile, N-tile) and tap, hi-only weight
// images (a tap is NT KiB; slot X is rounded up to whole 4 KiB refill rounds: at NT = 2 its five taps are 10 KiB and the refill fetches 12 - the two
// extra KiB are the head of tap 5 in the chunk image, never read from X)
constexpr int sp_tap(int NT, int TERMS = 3) { return (TERMS == 3 ? 2 : 1) * NT * 1024; }      // bytes of one tap: (hi | lo) x NT x 1 KiB
constexpr int sp_x(int NT, int TERMS = 3) { return (5 * sp_tap(NT, TERMS) + 4095) / 4096 * 4096; }      // slot X: taps 0..4
constexpr int sp_y(int NT, int TERMS = 3) { return 4 * sp_tap(NT, TERMS); }                  // slot Y: taps 5..8
constexpr int SP_TAB = PP_MAXCH * 32;                                               // chunk table: 32 B per chunk (see the kernel's prologue)
constexpr int sp_lds(int MT, int NT, int TERMS = 3) { return sp_x(NT, TERMS) + sp_y(NT, TERMS) + 2 * sp_patch(MT) + SP_TAB; }
// the A9 float4 of the next chunk are staged under the seven taps that carry no refill (0-3, 5-7), front-loaded: with A9 = 6 one under
// each of taps 0-3, 5, 6; with A9 = 10 two under taps 0-2 and one under taps 3, 5, 6, 7
constexpr int sp_stage_n(int A9, int tap) {            // float4 staged under `tap`
    const int slot = tap < 4 ? tap : tap < 8 ? tap - 1 : 7;      // taps 0-3 -> slots 0-3, taps 5-7 -> slots 4-6
    if (tap == 4 || tap == 8) return 0;
    return A9 / 7 + (slot < A9 % 7 ? 1 : 0);
}
constexpr int sp_stage_0(int A9, int tap) { int n = 0; for (int t = 0; t < tap; ++t) n += sp_stage_n(A9, t); return n; }      // first float4 staged under `tap`

__device__ __forceinline__ void glds16_sp(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

#ifdef PP_PROBE_BUILD
__device__ unsigned long long* g_sp_dbg = nullptr;
#define SP_STAMP(k) do { if (stamp_buf != nullptr && blockIdx.x == 0 && tid == 0 && stamp_n < 64) stamp_buf[stamp_n * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#endif

template <int MT, int NT, bool RES, int TERMS = 3>
__global__ __launch_bounds__(256, 1) void conv_sp_kernel(const PPParams p) {
    constexpr int TH = sp_rows(MT), NPIX = sp_npix(MT), A9 = sp_a9(MT), PATCH = sp_patch(MT);
    constexpr int TAPB = sp_tap(NT, TERMS), XB = sp_x(NT, TERMS), COUT = 32 * NT, OB = COUT * 4;        // OB: bytes of an output pixel
    constexpr int NFRAG = (TERMS == 3 ? 2 : 1) * (MT + NT), NMMA = TERMS * MT * NT;      // fragment reads / MFMAs of a tap
    static_assert(TERMS == 3 || TERMS == 1, "split terms");
    static_assert(sp_y(NT, TERMS) % 4096 == 0 && 9 * TAPB >= XB, "refill rounds of 4 KiB; slot X's over-fetch stays inside the chunk image");
    // hand-counted vmcnt: the patch requests issued behind a refill and in front of the barrier that publishes it
    constexpr int REQ1 = sp_stage_0(A9, 4), REQ2 = A9 - REQ1 + 3;      // taps 0-3; taps 5-7 + the three coefficient loads of tap 7 (scale, shift, operand scale)
    static_assert(sp_stage_0(A9, 9) == A9 && REQ1 + NT + 4 * MT * NT < 64, "staging distribution / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr unsigned PATCH0 = (unsigned)(XB + sp_y(NT, TERMS)), TAB = PATCH0 + 2u * (unsigned)PATCH;
    const pp_float_cptr scale_c = (pp_float_cptr)(uintptr_t)p.scale;
    const int nch = p.n9;

    // ---- per-lane constants of the staging ----------------------------------------------------------------------------------------------
    const int qi = tid & 3, p0 = tid >> 2;
    unsigned pk[A9], ldsw[A9];
#pragma unroll
    for (int i = 0; i < A9; ++i) {
        const int pp = min(p0 + 64 * i, NPIX - 1);
        const int py = pp / SP_PW, px = pp - py * SP_PW;
        pk[i] = (unsigned)(py * p.W + px) | ((py == 0 ? 1u : 0u) << 20) | ((py == TH + 1 ? 1u : 0u) << 21) | ((px == 0 ? 1u : 0u) << 22) |
                ((px == SP_PW - 1 ? 1u : 0u) << 23);
        ldsw[i] = PATCH0 + (unsigned)((py * SP_PITCH + px) * 64) + (unsigned)((((qi >> 1) ^ ((px >> 2) & 3)) << 4) + (qi & 1) * 8);
    }
    const int pix_safe = p.W + 1;

    // A-fragment addresses inside patch 0: lane = pixel (row prow of the M-tile's two rows, column pcol), k-half hi; [kx][term]
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned a_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned base = PATCH0 + (unsigned)(((wq * 2 * MT + prow) * SP_PITCH + pcol + kx) * 64);
        const unsigned s = (unsigned)(((pcol + kx) >> 2) & 3);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a_addr[kx][tm] = base + ((((unsigned)(hi + 2 * tm)) ^ s) << 4);
    }
    const unsigned b_lane = (unsigned)lane * 16u;
    const unsigned dma_lane = (unsigned)(wq * 1024 + lane * 16);

    // epilogue geometry (the lane transpose of pp_common.h)
    const bool bit3 = (lane & 8) != 0;
    const int em = lane & 7;
    const int ch_of_col = 4 * (l31 & 7) + (l31 >> 3);
    const unsigned e_lane = (unsigned)((4 * hi + ((lane >> 3) & 3)) * OB + em * 16);

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

    // ---- staging state --------------------------------------------------------------------------------------------------------------------
    float4 ra[A9];
    unsigned zero_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    struct Coef { float4 csc, csh; float ascale; unsigned inval; };
    Coef cf0, cf1;
    typedef const __attribute__((address_space(1))) char* gptr;      // (pointers rebuilt from table words must carry the global address space: a generic pointer makes every request a flat_load, which counts in lgkmcnt as well)
    struct Src { gptr base; unsigned cs4; };
    const unsigned q16 = (unsigned)qi * 16u;
    struct Desc { gptr base; const char* cb; unsigned cs4; int edge; unsigned sofs; };
    struct Ent { uint4 a, b; };      // table entry of a chunk: a = (source pointer, weight image pointer), b = (cstride, coff, gn_c0, seg)
    auto fetch = [&](int c) __attribute__((always_inline)) -> Ent {      // (a wave-uniform LDS read; the values reach scalar registers in describe)
        Ent e;
        e.a = *reinterpret_cast<const uint4*>(smem + TAB + (unsigned)c * 32u); e.b = *reinterpret_cast<const uint4*>(smem + TAB + (unsigned)c * 32u + 16u);
        return e;
    };
    auto sptr = [&](unsigned lo, unsigned hi) __attribute__((always_inline)) -> gptr {
        return (gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)lo));
    };
    auto describe = [&](const PPTile& tl, const Ent& e) __attribute__((always_inline)) -> Desc {
        // (32-bit index arithmetic: conv_sp_supported bounds B H W cstride below 2^31 - the 64-bit products of the first version were a
        // serial chain of ~60 scalar instructions per chunk)
        const int pix = (tl.b * p.H + tl.oy0) * p.W + tl.ox0 - p.W - 1;
        const int cstride = __builtin_amdgcn_readfirstlane((int)e.b.x), coff = __builtin_amdgcn_readfirstlane((int)e.b.y);
        const int gn_c0 = __builtin_amdgcn_readfirstlane((int)e.b.z), seg = __builtin_amdgcn_readfirstlane((int)e.b.w);
        Desc d;
        d.sofs = (unsigned)(8 * tl.b + seg) * 4u;
        d.edge = tl.edge;
        d.cb = reinterpret_cast<const char*>(p.coef + (tl.b * 2 * p.coef_stride + gn_c0));
        d.base = sptr(e.a.x, e.a.y) + (long)(pix * cstride + coff) * 4;
        d.cs4 = (unsigned)cstride * 4u;
        return d;
    };
    auto prep = [&](Coef& N, const Desc& d) __attribute__((always_inline)) -> Src {
        // (the operand scale as a VECTOR load - zero_v is a register hipcc cannot fold: a uniform address would become an s_load)
        N.ascale = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scale) + d.sofs + zero_v);
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
        const pp_f4v t_ = *reinterpret_cast<const __attribute__((address_space(1))) pp_f4v*>(sr.base + (__umul24(px, sr.cs4) + q16));
        ra[i] = make_float4(t_.x, t_.y, t_.z, t_.w);
    };
    // GroupNorm + SiLU + operand scale + fp16 hi / lo split of float4 number i -> patch at byte offset pofs.  Plain C++ (no asm blocks): the
    // ~60 instructions are spread by the scheduler over the MFMA shadows of a tap (sched_group_barrier pattern below); the values are
    // those of conv_mfma16's staging (silu_pp; hi = RNE16(x), lo = RNE16(x - hi)).  Every thread stores: the threads past the end of the
    // patch (A9 * 256 > 4 NPIX) hold the clamped last pixel and write ITS values to ITS address.  The launches this kernel takes are
    // GroupNorm + SiLU on every chunk (conv_sp_supported).
    auto transform_one = [&](const Coef& S, int i, unsigned pofs) __attribute__((always_inline)) {
        float4 v = ra[i];
        v.x = silu_pp(v.x * S.csc.x + S.csh.x); v.y = silu_pp(v.y * S.csc.y + S.csh.y);
        v.z = silu_pp(v.z * S.csc.z + S.csh.z); v.w = silu_pp(v.w * S.csc.w + S.csh.w);
        const float f = ((S.inval >> i) & 1u) ? 0.0f : S.ascale;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        if constexpr (TERMS == 3) {
            f16x4 l = {(_Float16)(v.x - (float)h[0]), (_Float16)(v.y - (float)h[1]), (_Float16)(v.z - (float)h[2]), (_Float16)(v.w - (float)h[3])};
            const unsigned addr = ldsw[i] + pofs;
            *reinterpret_cast<f16x4*>(smem + addr) = h;
            *reinterpret_cast<f16x4*>(smem + (addr ^ 32u)) = l;
        } else {      // one rounding to fp16; the pixel record keeps its 64-byte pitch, its lo half stays unwritten and unread
            *reinterpret_cast<f16x4*>(smem + ldsw[i] + pofs) = h;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    float run1[4 * NT], run2[4 * NT];
#pragma unroll
    for (int j = 0; j < 4 * NT; ++j) { run1[j] = 0.f; run2[j] = 0.f; }
    int run_b = -1, run_n = 0;

    auto flush_stats = [&]() __attribute__((always_inline)) {
        if (p.stats_out == nullptr || run_b < 0) return;
#pragma unroll
        for (int j = 0; j < 4 * NT; ++j) {
            double a = (double)run1[j], q = (double)run2[j];
            a += __shfl_xor(a, 8); q += __shfl_xor(q, 8);
            a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            if (lane < 8) {
                double* dst = p.stats_out + ((size_t)run_b * COUT + (j >> 2) * 32 + em * 4 + (j & 3)) * 2;
                unsafeAtomicAdd(dst, a); unsafeAtomicAdd(dst + 1, q);
            }
            run1[j] = 0.f; run2[j] = 0.f;
        }
        run_n = 0;
    };

    auto tile_offs = [&](int mt, int nt, int g) __attribute__((always_inline)) -> size_t {
        return (size_t)((mt * 2 + (g >> 1)) * p.W) * OB + (size_t)((g & 1) * 8 * OB + nt * 128) + e_lane;
    };
    // closes a tile: bias (+ time-embedding projection), residual, transpose, streamed stores, statistics.  The residual of accumulator
    // tile (mt, nt + 1) is requested before tile (mt, nt) is processed.
    // (the bias and the output scale of a tile are fetched when the tile is OPENED - tile_inputs - and ride along: at the end of a tile a
    // wave that is alone on its SIMD has nothing to cover a memory round trip with)
    // K-segment ids of the (up to three) segments' first chunks and of the last chunk: found once (kernel-argument reads)
    int seg_c1 = -1, seg_c2 = -1;
    for (int c = 1; c < nch; ++c)
        if (p.ch[c].seg != p.ch[c - 1].seg) { if (seg_c1 < 0) seg_c1 = c; else seg_c2 = c; }
    const int seg_id[4] = {p.ch[0].seg, p.ch[seg_c1 > 0 ? seg_c1 : 0].seg, p.ch[seg_c2 > 0 ? seg_c2 : 0].seg, p.ch[nch - 1].seg};
    float addv[NT]; float oscale = 0.f;
    auto tile_inputs = [&](const PPTile& tl) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) addv[nt] = p.addvec[(size_t)tl.b * p.addvec_bs + nt * 32 + ch_of_col];      // (conv_sp_supported: never null - the NT loads are counted in chunk 0's vmcnt)
        const float inv_last = scale_c[8 * tl.b + 4 + seg_id[3]];
        oscale = p.out_scale * (1.0f / 256.0f) * inv_last;
        asm volatile("" : "+v"(oscale));      // (consumed HERE: no scalar load may stay in flight into the chunk loop)
    };
    auto epilogue = [&](const PPTile& tl) __attribute__((always_inline)) {
        if (tl.b != run_b || run_n >= 32) { flush_stats(); run_b = tl.b; }
        ++run_n;
        const size_t pix0 = ((size_t)tl.b * p.H + tl.oy0 + wq * 2 * MT) * p.W + tl.ox0;
        char* obase = reinterpret_cast<char*>(p.out + pix0 * COUT);
        const char* rbase = RES ? reinterpret_cast<const char*>(p.residual + pix0 * COUT) : nullptr;
        float4 rv[3][4];      // residual of accumulator tiles k, k + 1, k + 2: requested TWO tiles ahead (one tile ahead part of the round trip was exposed: -3 % on the 64-channel level's residual launches, probe r05)
        if constexpr (RES) {
#pragma unroll
            for (int g = 0; g < 4; ++g) rv[0][g] = nt_load4(rbase + tile_offs(0, 0, g));
            if (MT * NT > 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) rv[1][g] = nt_load4(rbase + tile_offs(1 / NT, 1 % NT, g));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int k = mt * NT + nt;
                if constexpr (RES) {
                    if (k + 2 < MT * NT) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) rv[(k + 2) % 3][g] = nt_load4(rbase + tile_offs((k + 2) / NT, (k + 2) % NT, g));
                    }
                }
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = acc[mt][nt][r] * oscale + addv[nt];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    oct_transpose(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3], bit3);
                    float4 v = make_float4(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
                    if constexpr (RES) {
                        const float rsc = p.res_scale; const float4 r4 = rv[k % 3][g];
                        v.x = fmaf(r4.x, rsc, v.x); v.y = fmaf(r4.y, rsc, v.y); v.z = fmaf(r4.z, rsc, v.z); v.w = fmaf(r4.w, rsc, v.w);
                    }
                    *reinterpret_cast<float4*>(obase + tile_offs(mt, nt, g)) = v;
                    run1[nt * 4 + 0] += v.x; run1[nt * 4 + 1] += v.y; run1[nt * 4 + 2] += v.z; run1[nt * 4 + 3] += v.w;
                    run2[nt * 4 + 0] += v.x * v.x; run2[nt * 4 + 1] += v.y * v.y; run2[nt * 4 + 2] += v.z * v.z; run2[nt * 4 + 3] += v.w * v.w;
                }
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };

    // ---- fragments: two named sets (the fragments of tap + 1 are requested under the MFMAs of tap) -----------------------------------------
    f16x8 fa[2][MT][2], fb[2][NT][2];
    // tap of the chunk whose patch sits at byte offset pofs (0 / PATCH): A from the patch, B from slot X (taps 0..4) / Y (taps 5..8); in the
    // order the MFMA groups of mma_tap need them
    auto load_tap = [&](int set, int tap, unsigned pofs) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap % 3;
        const unsigned wb = (unsigned)(tap < 5 ? tap * TAPB : XB + (tap - 5) * TAPB) + b_lane;
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][mt][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][1] + (unsigned)((mt * 2 + ky) * SP_PITCH * 64) + pofs);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[set][nt][0] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(nt * 1024));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[set][mt][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + (unsigned)((mt * 2 + ky) * SP_PITCH * 64) + pofs);
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[set][nt][1] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(NT * 1024 + nt * 1024));
        }
    };
    // fragment number idx (0 .. 2 MT + 2 NT - 1, in load_tap's order) of a tap, alone: the refill taps request the next tap's fragments one per MFMA
    // shadow (as one burst in front of their individually fenced MFMAs the twelve reads cost ~150 cycles per refill tap)
    auto load_frag = [&](int set, int tap, unsigned pofs, int idx) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap % 3;
        const unsigned wb = (unsigned)(tap < 5 ? tap * TAPB : XB + (tap - 5) * TAPB) + b_lane;
        if constexpr (TERMS == 3) {
            if (idx < MT) fa[set][idx][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][1] + (unsigned)((idx * 2 + ky) * SP_PITCH * 64) + pofs);
            else if (idx < MT + NT) fb[set][idx - MT][0] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)((idx - MT) * 1024));
            else if (idx < 2 * MT + NT) fa[set][idx - MT - NT][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + (unsigned)(((idx - MT - NT) * 2 + ky) * SP_PITCH * 64) + pofs);
            else fb[set][idx - 2 * MT - NT][1] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(NT * 1024 + (idx - 2 * MT - NT) * 1024));
        } else {      // hi fragments only: NT B fragments, then MT A fragments
            if (idx < NT) fb[set][idx][0] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(idx * 1024));
            else fa[set][idx - NT][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + (unsigned)(((idx - NT) * 2 + ky) * SP_PITCH * 64) + pofs);
        }
    };
    // group g of a tap's 3 MT groups of NT MFMAs: term g / MT (a_lo w_hi, a_hi w_lo, a_hi w_hi), M-tile g % MT - an accumulator recurs every
    // MT NT MFMAs
    auto mma_group = [&](int set, int g) __attribute__((always_inline)) {
        const int term = TERMS == 3 ? g / MT : 2, mt = g % MT;      // (TERMS = 1: the a_hi w_hi product alone)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][mt][term == 0 ? 1 : 0], fb[set][nt][term == 1 ? 1 : 0], acc[mt][nt], 0, 0, 0);
    };
    auto mma_tap = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < TERMS * MT; ++g) mma_group(set, g);
    };
    // a tap whose side work is plain C++ (staging, requests, scalar work): the scheduler places it into the MFMA shadows by this pattern -
    // per MFMA at most one LDS read, two plain VALU, one transcendental and two SALU (a wave alone on its SIMD hides ~5 issue slots per
    // 32-cycle MFMA; the first version let 5-8 instructions into each of the first nine shadows and none into the rest: 920-1000 cycles
    // per tap of 768 matrix-pipe cycles); the LDS writes and the request in the last quarter
    auto tap_pattern = [&]() __attribute__((always_inline)) {
        if constexpr (TERMS == 3) {
#pragma unroll
        for (int k = 0; k < 3 * MT * NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k < 2 * (MT + NT)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k < 2 * (MT + NT)) __builtin_amdgcn_sched_group_barrier(0x002, A9 > 7 ? 2 : 1, 0);
            else __builtin_amdgcn_sched_group_barrier(0x002, A9 > 7 ? 5 : 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
            if (k >= 3 * MT * NT - 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (k == 3 * MT * NT - 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        } else {
            // a third of the shadows for the same staging: per MFMA one LDS read (six of eight), the vector work in equal shares (the tap is
            // bound by its vector issue as much as by its eight MFMAs), LDS writes and the request behind the last ones
#pragma unroll
        for (int k = 0; k < MT * NT; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (k < NFRAG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, A9 > 7 ? 10 : 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, A9 > 7 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
            if (k >= MT * NT - 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (k >= MT * NT - 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        }
    };
    // a tap that carries LDS-DMA refill pieces (asm statements: pinned by hand, ONE piece per MFMA shadow): `nrounds` x 4 KiB from tap t0 of
    // the chunk image `src` on.  Wave w fetches the w-th quarter of the range, contiguous: four pieces share one scalar base and one M0
    // (the instruction offset 0 / 1024 / 2048 / 3072 moves the global AND the LDS address), so a piece is one instruction plus a scalar
    // bump every fourth - the first version (M0 saved / set / restored and a 64-bit base add per piece) cost ~30 cycles per piece beyond
    // the MFMA it sat behind
    auto mma_tap_refill = [&](int set, const __attribute__((address_space(1))) char* src, int t0, int nrounds, int nset, int ntap, unsigned npofs) __attribute__((always_inline)) {
        const unsigned dst0 = (unsigned)(t0 < 5 ? t0 * TAPB : XB + (t0 - 5) * TAPB) + (unsigned)(wq * nrounds * 1024);
        const __attribute__((address_space(1))) char* sb = src + t0 * TAPB + wq * nrounds * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
        for (int g = 0; g < TERMS * MT; ++g) {
            const int term = TERMS == 3 ? g / MT : 2, mt = g % MT;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][mt][term == 0 ? 1 : 0], fb[set][nt][term == 1 ? 1 : 0], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int j = g * NT + nt;
                if (j < nrounds) {
                    if ((j & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst0 + (unsigned)(j * 1024)) : "memory");
                    if ((j & 3) == 0) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(b_lane), "s"(sb + j * 1024) : "memory");
                    if ((j & 3) == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(b_lane), "s"(sb + (j - 1) * 1024) : "memory");
                    if ((j & 3) == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(b_lane), "s"(sb + (j - 2) * 1024) : "memory");
                    if ((j & 3) == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(b_lane), "s"(sb + (j - 3) * 1024) : "memory");
                }
                if (j < NFRAG) load_frag(nset, ntap, npofs, j);      // fragment j of the tap behind this one
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
    };

    int stamp_n = 0; (void)stamp_n;
#ifdef PP_PROBE_BUILD
    unsigned long long* const stamp_buf = g_sp_dbg;
#endif
    // (seg_c1 / seg_c2: first chunks of K-segments 1 and 2 - the accumulator changes units there)
    gptr wimg_next = (gptr)(uintptr_t)p.ch[1].wimg;      // weight image of the chunk behind the current one
    Desc dn;            // descriptor of the chunk whose coefficients are fetched next (three ahead of the one being multiplied)
    Src sr;             // source of the requests of the current chunk (the chunk two ahead), prepared during the previous chunk
    // ---- one chunk: MFMAs of chunk (it, c) from patch SI; staging of the next chunk into patch SI ^ 1 (taps 0-3, 5, 6); requests of the
    // chunk after that; refills of slot X under tap 4 and of slot Y under tap 8; descriptor / coefficients of the chunk three ahead under
    // taps 6 / 7.  On entry: the fragments of tap 0 are in set SI, slots X and Y hold this chunk's weights (Y possibly still landing:
    // waited for at the first barrier), ra[] holds (or is receiving) the raw patch of the next chunk, C = its coefficients, N / sr = the
    // coefficients / source of the chunk two ahead.
    auto chunk = [&](int it, int c, auto SI_, bool last_chunk) __attribute__((always_inline)) {
        constexpr int SI = decltype(SI_)::value;
        constexpr unsigned PCUR = SI ? PATCH : 0, PNXT = SI ? 0 : PATCH;
        Coef& C = SI == 0 ? cf1 : cf0;      // coefficients of the chunk being staged (the next one)
        Coef& N = SI == 0 ? cf0 : cf1;      // coefficients of the chunk being requested (two ahead); C of the next chunk
        // (behind the last chunk of the launch the refills / fragment requests of a "next" chunk still run - into free slots, from valid
        // addresses, never multiplied - so that the chunk body has no branches; the kernel drains vmcnt before it ends)
        const gptr wnext = wimg_next;
        (void)last_chunk;
        SP_STAMP(0);
        // (nine taps per chunk: the fragment set of tap t is (t + SI) & 1)
        // ---- taps 0..3 (+ their share of the staging) ----------------------------------------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            __builtin_amdgcn_sched_barrier(0);
            load_tap((tap + 1 + SI) & 1, tap + 1, PCUR);
#pragma unroll
            for (int i = sp_stage_0(A9, tap); i < sp_stage_0(A9, tap) + sp_stage_n(A9, tap); ++i) { transform_one(C, i, PNXT); issue_one(N, sr, i); }
            mma_tap((tap + SI) & 1);
            tap_pattern();
            __builtin_amdgcn_sched_barrier(0);
            SP_STAMP(1 + tap);
        }
        __builtin_amdgcn_sched_barrier(0);
        // tap 4: its fragments are in registers -> slot X is free; slot Y (taps 5..8, requested under tap 8 of the previous chunk) must have
        // landed: behind its pieces only the requests of taps 0..3 were issued
        if (c == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(REQ1 + NT + 4 * MT * NT) : "memory");      // (+ the stores of the tile closed in front and the bias loads of this one)
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(REQ1) : "memory");
        // hipcc does not see the LDS-DMA pieces: its own counted wait in front of the first use of a raw patch register would be short by the
        // pieces issued in between, i.e. it would wait for pieces that have just been issued (stamps: ~400 cycles at the loop's back edge).
        // The raw registers staged before the next hand wait are complete HERE (they are older than the REQ1 youngest requests): touching
        // them here makes hipcc place its wait where it costs nothing
#pragma unroll
        for (int i = REQ1; i < A9; ++i) asm volatile("" :: "v"(ra[i].x));
        SP_STAMP(13);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SP_STAMP(5);
        mma_tap_refill((4 + SI) & 1, wnext, 0, XB / 4096, (5 + SI) & 1, 5, PCUR);
        SP_STAMP(6);
        // ---- taps 5, 6: staging of float4 4, 5; tap 6 also fetches the descriptor of the chunk three ahead; tap 7 its coefficients -----------
#pragma unroll
        for (int tap = 5; tap < 8; ++tap) {
            __builtin_amdgcn_sched_barrier(0);
            int q = c + 3, wrap = 0;                             // (two chunks per tile: three ahead may be two tiles ahead)
            if (q >= nch) { q -= nch; wrap = 1; }
            if (q >= nch) { q -= nch; wrap = 2; }
            const int c2 = c + 2 >= nch ? c + 2 - nch : c + 2;
            Ent eq, e2;
            if (tap == 6) { eq = fetch(q); e2 = fetch(c2); }      // in FRONT of the fragment reads: their values are waited for with a counted lgkmcnt
            load_tap((tap + 1 + SI) & 1, tap + 1, PCUR);
#pragma unroll
            for (int i = sp_stage_0(A9, tap); i < sp_stage_0(A9, tap) + sp_stage_n(A9, tap); ++i) { transform_one(C, i, PNXT); issue_one(N, sr, i); }
            if (tap == 6) {
                dn = describe(tile_of(it + wrap), eq);
                wimg_next = sptr(e2.a.z, e2.a.w);      // the next chunk's `wnext`
            }
            if (tap == 7) sr = prep(C, dn);      // (C: every float4 of the next chunk has been staged; it becomes N of the next chunk)
            mma_tap((tap + SI) & 1);
            tap_pattern();
            __builtin_amdgcn_sched_barrier(0);
            SP_STAMP(2 + tap);
        }
        __builtin_amdgcn_sched_barrier(0);
        // tap 8: its fragments are in registers -> slot Y and patch SI are free; slot X (next chunk's taps 0..4) must have landed - behind
        // its pieces: the requests of taps 5-7 and the two coefficient loads of tap 7 - and the next chunk's patch must be complete
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(REQ2) : "memory");
#pragma unroll
        for (int i = 0; i < REQ1; ++i) asm volatile("" :: "v"(ra[i].x));      // (as above: the requests of taps 0-3 are older than the REQ2 youngest loads)
        SP_STAMP(14);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        SP_STAMP(10);
        mma_tap_refill((8 + SI) & 1, wnext, 5, 4 * TAPB / 4096, (9 + SI) & 1, 0, PNXT);
        SP_STAMP(11);
        SP_STAMP(12);
        ++stamp_n;
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------------------------------
    if (ntl <= 0) return;
    // chunk table -> LDS (uniform scalar loads of the kernel argument, once per workgroup)
    for (int c = 0; c < nch; ++c) {
        if (tid == 0) {
            const unsigned long long sp_ = (unsigned long long)(uintptr_t)p.ch[c].src, wp_ = (unsigned long long)(uintptr_t)p.ch[c].wimg;
            *reinterpret_cast<uint4*>(smem + TAB + (unsigned)c * 32u) = make_uint4((unsigned)sp_, (unsigned)(sp_ >> 32), (unsigned)wp_, (unsigned)(wp_ >> 32));
            *reinterpret_cast<uint4*>(smem + TAB + (unsigned)c * 32u + 16u) = make_uint4((unsigned)p.ch[c].cstride, (unsigned)p.ch[c].coff, (unsigned)p.ch[c].gn_c0, (unsigned)p.ch[c].seg);
        }
    }
    __syncthreads();
    {
        // chunk 0: requested, staged into patch 0; chunk 1: requested; chunk 2: coefficients fetched, source prepared
        const Src s0 = prep(cf0, describe(tile_of(0), fetch(0)));
#pragma unroll
        for (int i = 0; i < A9; ++i) issue_one(cf0, s0, i);
        {
            const char* w0 = reinterpret_cast<const char*>(p.ch[0].wimg);
#pragma unroll
            for (int j = 0; j < (TERMS == 3 ? 9 * TAPB : XB) / 4096; ++j) glds16_sp(dma_lane, w0 + j * 4096, (unsigned)(j * 4096) + (unsigned)(wq * 1024));      // TERMS = 3: X | Y are contiguous, as the chunk image is; TERMS = 1: slot X (taps 0..4 + the round-up)
            if constexpr (TERMS != 3) {
#pragma unroll
                for (int j = 0; j < 4 * TAPB / 4096; ++j) glds16_sp(dma_lane, w0 + 5 * TAPB + j * 4096, (unsigned)(XB + j * 4096) + (unsigned)(wq * 1024));      // slot Y: taps 5..8
            }
        }
#pragma unroll
        for (int i = 0; i < A9; ++i) transform_one(cf0, i, 0);
        const Src s1 = prep(cf1, describe(tile_of(0), fetch(1)));      // (nch >= 2)
#pragma unroll
        for (int i = 0; i < A9; ++i) issue_one(cf1, s1, i);
        sr = prep(cf0, describe(tile_of(2 >= nch ? 1 : 0), fetch(2 >= nch ? 2 - nch : 2)));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_tap(0, 0, 0);
    }
    // (every K-segment has an even number of chunks: a chunk's parity - its patch buffer, coefficient set and fragment sets - is compile-time.
    // The chunk loop runs per K-segment so that the change of the accumulator's units at a segment switch sits between two loops: a branch
    // inside the chunk pair would end its basic block, and hipcc sinks the descriptor / coefficient arithmetic of taps 6 / 7 to its first
    // use behind that branch - the head of the next chunk, where nothing covers it: ~1 k cycles per chunk)
    const int seg_lo[4] = {0, seg_c1 > 0 ? seg_c1 : nch, seg_c2 > 0 ? seg_c2 : nch, nch};
#pragma unroll 1
    for (int it = 0; it < ntl; ++it) {
        tile_inputs(tile_of(it));
#pragma unroll 1
        for (int sg = 0; sg < 3; ++sg) {
            const int lo = seg_lo[sg], hi = seg_lo[sg + 1];
            if (lo >= hi) continue;
            if (sg > 0) {
                // the accumulator changes units: from the previous K-segment's operand scale to this one's (powers of two: exact)
                const PPTile tl = tile_of(it);
                const float ratio = scale_c[8 * tl.b + seg_id[sg]] * scale_c[8 * tl.b + 4 + seg_id[sg - 1]];
                if (ratio != 1.0f) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= ratio;
                }
            }
#pragma unroll 1
            for (int c = lo; c < hi; c += 2) {
                chunk(it, c, ic<0>{}, false);
                chunk(it, c + 1, ic<1>{}, false);
            }
        }
        // (behind the chunk loops, not under a condition inside the chunk body: hipcc hoisted the 128 accumulator reads of a conditional
        // epilogue above the branch - ~1 k cycles in EVERY chunk)
        epilogue(tile_of(it));
    }
    flush_stats();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the refills issued behind the last chunk
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------

static int ilog2_exact_sp(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

bool conv_sp_supported(const ConvParams& p, int stride, int up, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_SP") ? atoi(getenv("PNPFLOW_HIP_SP")) : 1;      // test-only A/B switch: 0 off, 1 on, 2 the 128-channel level only, 3 on in the default mode only
    if (mode == 0 || (terms != 3 && !(terms == 1 && mode != 3)) || stride != 1 || up != 0 || p.gnb_x != nullptr) return false;
    if (p.Cout != 128 && !(p.Cout == 64 && mode == 1)) return false;
    if (p.out_cstride != p.Cout || (p.residual != nullptr && p.res_cstride != p.Cout)) return false;
    // the hand-counted vmcnt in front of the first barrier of a tile's chunk 0 counts the NT bias loads of tile_inputs: a launch without a
    // bias vector would wait NT operations too loosely - such launches stay on conv_mfma16
    if (p.addvec == nullptr) return false;
    const int TH = p.Cout == 128 ? sp_rows(2) : sp_rows(4);      // 16 x 16-pixel workgroup tiles at 128 channels, 32 x 16 at 64
    if (p.H % TH || p.W % 16 || p.Hs != p.H || p.Ws != p.W) return false;
    // the staging packs the patch-pixel offset py * W + px (py <= TH + 1, px <= 17) into 16 bits (pk[], issue_one): ADVICE r5 - the bound
    // depends on the tile height, W <= 1024 at Cout 64 / W <= 2048 at Cout 128
    if ((TH + 1) * p.W + SP_PW - 1 > 65535) return false;
    if (ilog2_exact_sp(p.H / TH) < 0 || ilog2_exact_sp(p.W / 16) < 0) return false;
    // the persistent grid pays its prologue and pipeline fill over >= 4 tiles per workgroup
    const int grid = persistent_grid();
    if (grid < 8 || (long)p.B * (p.H / TH) * (p.W / 16) < 4L * grid) return false;
    int nch = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if ((double)p.B * p.H * p.W * s.cstride >= 2147483648.0) return false;      // 32-bit element offsets (describe)
        if (s.w_mode != 0 || s.w16 == nullptr || s.C % 32 || s.taps != 9 || s.xform != 2) return false;      // GroupNorm + SiLU 3x3 segments of an even number of 16-channel chunks
        nch += s.C / 16;
    }
    if (nch < 2 || nch > PP_MAXCH || (nch & 1)) return false;
    return p.gn_C > 0 && p.coef != nullptr && p.scale != nullptr;      // (with_coef: every GroupNorm-ed split-fp16 launch carries operand scales)
}

template <int MT, int NT, int TERMS>
static hipError_t launch_sp_t(const PPParams& p0, hipStream_t s) {
    static unsigned long long attr_set[2] = {0ull, 0ull};
    const bool res = p0.residual != nullptr;
    const void* kern = res ? reinterpret_cast<const void*>(conv_sp_kernel<MT, NT, true, TERMS>) : reinterpret_cast<const void*>(conv_sp_kernel<MT, NT, false, TERMS>);
    { hipError_t e = set_max_dynamic_lds_once(kern, attr_set[res ? 1 : 0], 160 * 1024); if (e != hipSuccess) return e; }
    const int grid = persistent_grid();
    if (grid <= 0) return hipErrorInvalidConfiguration;
    PPParams p = p0;
    p.lx = ilog2_exact_sp(p.W / 16); p.ly = ilog2_exact_sp(p.H / sp_rows(MT));
    p.rot = 5;
    if (res) hipLaunchKernelGGL((conv_sp_kernel<MT, NT, true, TERMS>), dim3(grid), dim3(256), sp_lds(MT, NT, TERMS), s, p);
    else hipLaunchKernelGGL((conv_sp_kernel<MT, NT, false, TERMS>), dim3(grid), dim3(256), sp_lds(MT, NT, TERMS), s, p);
    return hipGetLastError();
}

hipError_t launch_conv_sp(const PPParams& p0, hipStream_t s, int terms) {
    if (p0.n9 < 2 || (p0.n9 & 1) || p0.n1 != 0 || (terms != 3 && terms != 1)) return hipErrorInvalidValue;
    if (p0.cout == 128) return terms == 3 ? launch_sp_t<2, 4, 3>(p0, s) : launch_sp_t<2, 4, 1>(p0, s);
    if (p0.cout == 64) return terms == 3 ? launch_sp_t<4, 2, 3>(p0, s) : launch_sp_t<4, 2, 1>(p0, s);
    return hipErrorInvalidValue;
}

}  // namespace pf
