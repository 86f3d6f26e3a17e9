// Persistent two-team implicit-GEMM 3x3 convolution for the 128-channel level of the U-Net (Cout = 128: level 2), split-fp16 MFMA,
// fp32-equivalent - the structure of conv_pp64.hip (one 512-thread workgroup per CU for the whole launch, two teams of four waves one
// barrier apart, VALU phase / MFMA phase, in-register transposing epilogue, weights streamed from L2 into LDS by LDS-DMA) with what the
// wider level needs.  Reference: the convolutions of ResidualBlock at level width 128 (pnpflow/models.py:58-113).
//   * a team's tile is 8 rows x 16 pixels x all 128 output channels, its four waves a 2 x 2 grid of 64-pixel x 64-channel wave tiles (two
//     32-pixel M-tiles x two 32-column N-tiles: the 8 LDS fragment reads per 12 MFMAs and the 64 accumulator registers of conv_pp64) -
//     so a staged activation feeds TWICE the MFMAs it feeds at 64 output channels: per step a wave stages 3 float4 (conv_pp64: 6) for
//     the same 108 MFMAs, and the VALU phase of one team (which runs at less than half its stand-alone rate beside the other team's
//     MFMAs, profiles/r04_level1_probes.md) is no longer the longer phase;
//   * the weights of one 16-channel chunk are 72 KiB ([tap][hi | lo][N-tile 0..3][k-half][column][8 halfs]): two chunk slots do not fit
//     beside the patches, so the ring is kept per TAP - 16 slots of 8 KiB; tap k of step v (the same step sequence for both teams) lives
//     in slot (9 v + k) % 16.  Team 0 reads step v in phase 2 v, team 1 in phase 2 v + 1.  The refills are issued by the team that is in
//     its VALU phase - it has ~2.5 k cycles of slack per phase, while every LDS-DMA piece issued inside an MFMA phase costs that phase
//     100-185 cycles (MI355X_MICROARCH.md; round-5 stamps: 5.2 k cycles per 3.5 k of MFMAs with 8-14 pieces per wave in the phase):
//       - in its VALU phase of step v + 1 (phase 2 v + 1) team 0 fetches taps 0..6 of step v + 1 into the seven slots that hold taps
//         2..8 of step v - 1 (both teams finished with them in phase 2 v - 1) and waits for them (counted vmcnt: the DMA pieces are
//         issued BEFORE the patch prefetches of the phase, so the prefetches stay in flight) before the barrier that ends the phase;
//       - taps 7 and 8 of step v + 1 go where taps 0 and 1 of step v are - read by team 1 until phase 2 v + 1 ends - so team 1 fetches
//         them in ITS VALU phase of step v + 1 (phase 2 v + 2), waits for them and raises an LDS counter; team 0, whose MFMA phase of
//         step v + 1 runs beside, checks the counter before it requests the fragments of tap 7 (six taps = > 2 k cycles after the
//         pieces were issued: the check does not spin in practice, but the hand-over does not rest on timing); team 1 itself reads the
//         two taps after the next barrier;
//   * identity residual (ResidualBlock conv2 of the down path): the 64 registers of the residual tile live only during the VALU phase
//     that closes a tile, when the 64 fragment registers of the MFMA phase are free.
// Launches with folded 1x1 shortcut chunks stay on conv_mfma16 (a one-tap step stages as much as a nine-tap step for a ninth of its MFMAs).
// LDS: ring 16 x 8 KiB + two patches of [10 rows][pitch 20 pixels][4 pieces of 16 B] (conv_pp64's record and swizzle) = 156 672 B.
#include <cstdlib>
#include "pp_common.h"

namespace pf {

constexpr int P128_PITCH = 20, P128_PW = 18, P128_TH = 8;
constexpr int P128_NPIX = (P128_TH + 2) * P128_PW;                    // 180 patch pixels
constexpr int P128_PATCH = (P128_TH + 2) * P128_PITCH * 64;           // 12 800 B per team
constexpr int P128_A9 = (P128_NPIX * 4 + 255) / 256;                  // float4 per lane and chunk: 3
constexpr int P128_TAP = 8192;                                        // one tap of a 16-channel chunk: (hi | lo) x 4 N-tiles x 1 KiB
constexpr int P128_SLOTS = 16;
constexpr int P128_RING = P128_SLOTS * P128_TAP;                      // 131 072 B
constexpr int P128_CHUNK = 9 * P128_TAP;                              // weight image of a chunk in global memory: 73 728 B
constexpr int P128_FLAG = P128_RING + 2 * P128_PATCH;                 // LDS counter: late taps landed (4 per step, raised by the waves of team 1)
constexpr int P128_LDS = P128_FLAG + 16;                              // 156 688 B

// one LDS-DMA instruction: every lane fetches the 16 bytes at sbase + voff into LDS byte address lds_dst + 16 * lane (lds_dst is
// wave-uniform; M0 carries it and is restored: the compiler neither preserves M0 around an asm statement nor expects it changed)
__device__ __forceinline__ void glds16_p128(unsigned voff, const char* sbase, unsigned lds_dst) {      // (callers fold the piece offset into sbase: ONE lane-offset register)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

#ifdef PP_PROBE_BUILD
// tools/ubench/conv_pp128_probe.hip only: s_memtime stamps of workgroup 0, [team][step][8]
__device__ unsigned long long* g_pp128_dbg = nullptr;
#define P128_STAMP(k) do { if (stamp_buf != nullptr && blockIdx.x == 0 && t == 0 && stamp_n < 64) stamp_buf[(team * 64 + stamp_n) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define P128_STAMP(k) do { } while (0)
#endif

template <bool RES>
__global__ __launch_bounds__(512, 2) void conv_pp128_kernel(const PPParams p) {
    constexpr int TH = P128_TH, NPIX = P128_NPIX, A9 = P128_A9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int team = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int t = tid & 255, lane = t & 63, wq = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wq >> 1, wn = wq & 1;                                  // the wave's M-tile pair (rows 4 wm .. 4 wm + 3) and N-tile pair
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned patch0 = (unsigned)P128_RING + (unsigned)team * P128_PATCH;
    const pp_float_cptr scale_c = (pp_float_cptr)(uintptr_t)p.scale;      // scalar-cache reads (see conv_pp.hip)
    const int nch = p.n9;

    // ---- the nine taps of chunk 0 -> ring slots 0..8 ------------------------------------------------------------------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.ch[0].wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < P128_CHUNK / 16; i += 512) dst[i] = src[i];
        if (tid == 0) asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)P128_FLAG), "v"(0u) : "memory");      // (dynamic LDS starts at byte 0)
    }

    // ---- per-lane constants of the staging (conv_pp64.hip) -------------------------------------------------------------------------------
    // float4 number i of a chunk = patch pixel pp = (t >> 2) + 64 i, channel quad qi = t & 3.  pk: bits 0-15 pixel offset py * W + px inside
    // the patch, bits 20-23 the edges the pixel lies on (top, bottom, left, right); ldsw: byte address of the quad's hi halfs
    const int qi = t & 3, p0 = t >> 2;
    unsigned pk[A9], ldsw[A9];
#pragma unroll
    for (int i = 0; i < A9; ++i) {
        const int pp = min(p0 + 64 * i, NPIX - 1);
        const int py = pp / P128_PW, px = pp - py * P128_PW;
        pk[i] = (unsigned)(py * p.W + px) | ((py == 0 ? 1u : 0u) << 20) | ((py == TH + 1 ? 1u : 0u) << 21) | ((px == 0 ? 1u : 0u) << 22) |
                ((px == P128_PW - 1 ? 1u : 0u) << 23);
        ldsw[i] = patch0 + (unsigned)((py * P128_PITCH + px) * 64) + (unsigned)((((qi >> 1) ^ ((px >> 2) & 3)) << 4) + (qi & 1) * 8);
    }
    constexpr int LASTN = NPIX * 4 - (A9 - 1) * 256;            // threads that own a float4 number A9 - 1
    const int pix_safe = p.W + 1;                               // patch pixel (1, 1) = tile pixel (0, 0): inside the image for every tile

    // A-fragment addresses: lane = pixel (row prow of the M-tile's two rows, column pcol), k-half hi; [kx][term]
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned a_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned base = patch0 + (unsigned)(((wm * 4 + prow) * P128_PITCH + pcol + kx) * 64);
        const unsigned s = (unsigned)(((pcol + kx) >> 2) & 3);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a_addr[kx][tm] = base + ((((unsigned)(hi + 2 * tm)) ^ s) << 4);
    }
    // B fragments of the wave inside a tap slot: (term, N-tile 2 wn + nt) at term * 4096 + (2 wn + nt) * 1024 + 16 lane
    const unsigned b_lane = (unsigned)(wn * 2048 + lane * 16);
    const unsigned dma_lane = (unsigned)(lane * 16);

    // epilogue geometry: after the transpose the lane holds pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile, channels 32 N-tile + 4 em .. + 3
    const bool bit3 = (lane & 8) != 0;
    const int em = lane & 7;
    const int ch_of_col = 4 * (l31 & 7) + (l31 >> 3);
    const unsigned e_lane = (unsigned)((4 * hi + ((lane >> 3) & 3)) * 512 + wn * 256 + em * 16);      // inside a tile row of 16 pixels x 512 B

    // ---- this workgroup's tiles (as conv_pp.hip: XCD-contiguous ranges, the two teams interleaved, a rotated start) ---------------------
    const int G = gridDim.x;
    const int rg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int T = p.B << (p.lx + p.ly);
    const int t_begin = (int)((long)rg * T / G), t_end = (int)((long)(rg + 1) * T / G);
    const int ntl = t_end - t_begin;
    const int niter = (ntl + 1) / 2;
    const int rot = ntl > 0 ? (int)(((long)rg * p.rot) % ntl) : 0;
    auto tile_of = [&](int it) __attribute__((always_inline)) -> PPTile {
        const int idx = min(it * 2 + team, ntl - 1);
        const int wrapped = idx + rot >= ntl ? idx + rot - ntl : idx + rot;
        const int tl = t_begin + wrapped;
        PPTile r;
        const int tx = tl & ((1 << p.lx) - 1), ty = (tl >> p.lx) & ((1 << p.ly) - 1);
        r.b = tl >> (p.lx + p.ly); r.oy0 = ty * TH; r.ox0 = tx * 16;
        r.edge = (ty == 0 ? 1 : 0) | (ty == (1 << p.ly) - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == (1 << p.lx) - 1 ? 8 : 0);
        return r;
    };
    auto live_of = [&](int it) __attribute__((always_inline)) -> bool { return it >= 0 && (t_begin + it * 2 + team) < t_end; };

    // ---- registers of the staging (conv_pp64.hip: one raw patch in flight, every register re-requested the moment it has been staged) -----
    float4 ra[A9];
    struct Coef { float4 csc, csh; float ascale; unsigned inval; bool silu; };      // per-step staging coefficients: current / next
    Coef cf0, cf1;
    struct Src { const char* base; unsigned cs4; };                                 // uniform: first patch pixel of the step's chunk, bytes per pixel
    const unsigned q16 = (unsigned)qi * 16u;

    struct Desc { const char* base; const char* cb; unsigned cs4; int edge; float ascale; bool silu; };
    auto describe = [&](const PPTile& tl, int c) __attribute__((always_inline)) -> Desc {
        const long bpix = ((long)tl.b * p.H + tl.oy0) * p.W + tl.ox0;
        const int cstride = p.ch[c].cstride;
        Desc d;
        d.ascale = p.scale != nullptr ? scale_c[8 * tl.b + p.ch[c].seg] : 1.0f;
        d.silu = p.ch[c].xform == 2;
        d.edge = tl.edge;
        d.cb = reinterpret_cast<const char*>(p.coef + (size_t)tl.b * 2 * p.coef_stride + p.ch[c].gn_c0);
        d.base = reinterpret_cast<const char*>(p.ch[c].src + (bpix - p.W - 1) * cstride + p.ch[c].coff);
        d.cs4 = (unsigned)cstride * 4u;
        return d;
    };
    auto prep = [&](Coef& N, const Desc& d) __attribute__((always_inline)) -> Src {
        N.ascale = d.ascale; N.silu = d.silu;
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
    auto split_store = [&](float4 v, unsigned addr) __attribute__((always_inline)) {
        uint2 h, l;
        split4_pp(v, h.x, h.y, l.x, l.y);
        *reinterpret_cast<uint2*>(smem + addr) = h;
        *reinterpret_cast<uint2*>(smem + (addr ^ 32u)) = l;                       // the lo piece q + 2 sits at slot (q ^ s) ^ 2
    };
    auto transform_one = [&](const Coef& S, int i) __attribute__((always_inline)) {
        float4 v = ra[i];
        v.x = v.x * S.csc.x + S.csh.x; v.y = v.y * S.csc.y + S.csh.y; v.z = v.z * S.csc.z + S.csh.z; v.w = v.w * S.csc.w + S.csh.w;
        if (S.silu) silu4_pp(v);
        const float f = ((S.inval >> i) & 1u) ? 0.0f : S.ascale;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        if (i < A9 - 1 || t < LASTN) split_store(v, ldsw[i]);
    };

    // one accumulator per (M-tile, N-tile): four independent chains for the matrix pipe
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    float run1[8], run2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { run1[j] = 0.f; run2[j] = 0.f; }
    int run_b = -1, run_n = 0;

    auto flush_stats = [&]() __attribute__((always_inline)) {
        if (p.stats_out == nullptr || run_b < 0) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double a = (double)run1[j], q = (double)run2[j];
            a += __shfl_xor(a, 8); q += __shfl_xor(q, 8);
            a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            if (lane < 8) {
                double* dst = p.stats_out + ((size_t)run_b * 128 + (2 * wn + (j >> 2)) * 32 + em * 4 + (j & 3)) * 2;
                unsafeAtomicAdd(dst, a); unsafeAtomicAdd(dst + 1, q);
            }
            run1[j] = 0.f; run2[j] = 0.f;
        }
        run_n = 0;
    };

    // closes a tile: bias (+ time-embedding projection), residual, transpose, streamed stores, statistics.  The bias and the residual of the
    // wave's FIRST M-tile (32 registers) are requested at the start of the VALU phase that ends with the epilogue and land while the patch
    // is staged; the residual of the second M-tile is requested into the same registers once the first M-tile has been stored (a whole
    // residual tile beside the 64 accumulators, the raw patch and the staging coefficients spilled 160 registers)
    struct Res { float4 rv[2][4]; float addv[2]; };
    Res res;
    auto tile_offs = [&](int mt, int nt, int g) __attribute__((always_inline)) -> size_t {
        // pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile: row (g >> 1) of its two rows, column 8 (g & 1) + 4 hi + ((lane >> 3) & 3)
        return (size_t)((mt * 2 + (g >> 1)) * p.W) * 512 + (size_t)((g & 1) * 4096 + nt * 128) + e_lane;
    };
    auto issue_rv = [&](Res& R, const PPTile& tl, int mt) __attribute__((always_inline)) {
        if constexpr (RES) {
            const char* rbase = reinterpret_cast<const char*>(p.residual + (((size_t)tl.b * p.H + tl.oy0 + wm * 4) * p.W + tl.ox0) * 128);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) R.rv[nt][g] = nt_load4(rbase + tile_offs(mt, nt, g));
        }
    };
    auto issue_res = [&](Res& R, const PPTile& tl) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) R.addv[nt] = p.addvec != nullptr ? p.addvec[(size_t)tl.b * p.addvec_bs + (2 * wn + nt) * 32 + ch_of_col] : 0.f;
        issue_rv(R, tl, 0);
    };
    auto epilogue = [&](Res& R, const PPTile& tl) __attribute__((always_inline)) {
        if (tl.b != run_b || run_n >= 32) { flush_stats(); run_b = tl.b; }
        ++run_n;
        const float inv_last = p.scale != nullptr ? scale_c[8 * tl.b + 4 + p.ch[nch - 1].seg] : 1.0f;
        const float oscale = p.out_scale * (1.0f / 256.0f) * inv_last;
        char* obase = reinterpret_cast<char*>(p.out + (((size_t)tl.b * p.H + tl.oy0 + wm * 4) * p.W + tl.ox0) * 128);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = acc[mt][nt][r] * oscale + R.addv[nt];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    oct_transpose(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3], bit3);
                    float4 v = make_float4(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
                    if constexpr (RES) {
                        const float rsc = p.res_scale; const float4 r4 = R.rv[nt][g];
                        v.x = fmaf(r4.x, rsc, v.x); v.y = fmaf(r4.y, rsc, v.y); v.z = fmaf(r4.z, rsc, v.z); v.w = fmaf(r4.w, rsc, v.w);
                    }
                    nt_store4(obase + tile_offs(mt, nt, g), v);
                    run1[nt * 4 + 0] += v.x; run1[nt * 4 + 1] += v.y; run1[nt * 4 + 2] += v.z; run1[nt * 4 + 3] += v.w;
                    run2[nt * 4 + 0] += v.x * v.x; run2[nt * 4 + 1] += v.y * v.y; run2[nt * 4 + 2] += v.z * v.z; run2[nt * 4 + 3] += v.w * v.w;
                }
            }
            if (mt == 0) issue_rv(R, tl, 1);
        }
    };

    // MFMA phase of chunk c, whose taps sit in ring slots (rb + tap) % 16: per tap one A fragment pair (hi, lo) per M-tile and the B fragment
    // pairs of the wave's two N-tiles.  late_count != 0 (team 0, every step but the first): taps 7 and 8 of THIS step are being fetched
    // by team 1 during this phase; their fragments are requested once the LDS counter has reached late_count.
    auto mma_chunk = [&](const PPTile& tl, int c, int rb, unsigned late_count) __attribute__((always_inline)) {
        if (c == 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        } else if (p.scale != nullptr && p.ch[c - 1].seg != p.ch[c].seg) {
            // the accumulator changes units: from the previous K-segment's operand scale to this one's (powers of two: exact)
            const float ratio = scale_c[8 * tl.b + p.ch[c].seg] * scale_c[8 * tl.b + 4 + p.ch[c - 1].seg];
            if (ratio != 1.0f) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= ratio;
            }
        }
        __builtin_amdgcn_s_setprio(1);
        // fragments of tap + 1 are requested before the MFMAs of tap are issued (two named sets, conv_pp64.hip)
        f16x8 fa[2][2][2], fb[2][2][2];
        auto load_tap = [&](int set, int tap) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned off = (unsigned)((mt * 2 + ky) * P128_PITCH * 64);
                fa[set][mt][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + off);
                fa[set][mt][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][1] + off);
            }
            const unsigned wb = (unsigned)(((rb + tap) & 15) * P128_TAP) + b_lane;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                fb[set][nt][0] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(nt * 1024));
                fb[set][nt][1] = *reinterpret_cast<const f16x8*>(smem + wb + (unsigned)(4096 + nt * 1024));
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap == 6 && late_count != 0u) {
                unsigned landed;
                do {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(landed) : "v"((unsigned)P128_FLAG) : "memory");
                } while (__builtin_amdgcn_readfirstlane(landed) < late_count);
            }
            if (tap + 1 < 9) load_tap(cur ^ 1, tap + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- one step = VALU phase + MFMA phase of chunk c of the team's it-th tile; SI = c & 1 (nch is even: the step's global parity, i.e.
    // its coefficient set); rb = ring slot of the step's tap 0 ----------------------------------------------------------------------------
    int stamp_n = 0; (void)stamp_n;
#ifdef PP_PROBE_BUILD
    unsigned long long* const stamp_buf = g_pp128_dbg;      // null: no stamps (the probe's plain timing runs)
#endif
    Desc dn;            // descriptor of the step AFTER the current one
    int rb = 0;
    unsigned nstep = 0;              // steps completed by this team
    // the ring refills of a VALU phase (issued before the phase's patch prefetches): team 0 - taps 0..6 of the step it is staging (14 pieces
    // per wave); team 1 - taps 7, 8 of the same step, which team 0 is multiplying beside (4 pieces per wave).  Not on the first step: the
    // prologue placed all nine taps of chunk 0
    auto refill = [&](int c) __attribute__((always_inline)) {
        const char* src = reinterpret_cast<const char*>(p.ch[c].wimg);
        if (team == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const unsigned slot = (unsigned)(((rb + k) & 15) * P128_TAP);
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16_p128(dma_lane, src + (k * P128_TAP + (wq + 4 * j) * 1024), slot + (unsigned)((wq + 4 * j) * 1024));
            }
        } else {
#pragma unroll
            for (int k = 7; k < 9; ++k) {
                const unsigned slot = (unsigned)(((rb + k) & 15) * P128_TAP);
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16_p128(dma_lane, src + (k * P128_TAP + (wq + 4 * j) * 1024), slot + (unsigned)((wq + 4 * j) * 1024));
            }
        }
    };
    auto step = [&](int it, int c, auto SI_, auto FIRST_, bool last_step) __attribute__((always_inline)) {
        constexpr int SI = decltype(SI_)::value;
        // FIRST: the step opens a tile (c == 0) - compile-time, so that the bias / residual requests of the tile it closes are unconditional
        // loads (a conditionally executed load costs every later wait its count: vmcnt(0)); when there is no tile to close (the team's
        // first tile, or a dead last tile in front) they fetch a live tile's values and nothing is stored
        constexpr bool FIRST = decltype(FIRST_)::value != 0;
        Coef& C = SI == 0 ? cf0 : cf1;
        Coef& N = SI == 0 ? cf1 : cf0;
        const PPTile tl = tile_of(it);
        // ---- VALU phase: ring refills, then stage this step's patch, re-requesting every register for step + 1 as it is staged; on the
        // first step of a tile close the previous one ---------------------------------------------------------------------------------------
        P128_STAMP(0);
        bool close = false;
        if constexpr (FIRST) { close = it > 0 && live_of(it - 1); issue_res(res, tile_of(close ? it - 1 : it)); }
        const Src srN = prep(N, dn);
        const bool fill = nstep != 0u;
        if (fill) refill(c);
        P128_STAMP(1);
#pragma unroll
        for (int i = 0; i < A9; ++i) { transform_one(C, i); issue_one(N, srN, i); }
        // every refill piece of this phase has landed (memory operations return in order; the A9 prefetches behind them stay in flight)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A9) : "memory");
        if (team == 1 && fill && lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)P128_FLAG), "v"(1u) : "memory");      // (dynamic LDS starts at byte 0)
        P128_STAMP(2);
        if constexpr (FIRST) { if (close) epilogue(res, tile_of(it - 1)); }
        // the descriptor of step + 2 (scalar loads with a run-time chunk index), requested here so that the round trips pass behind the barrier
        {
            const int wrap = c + 2 >= nch ? 1 : 0;
            dn = describe(tile_of(it + wrap), c + 2 - (wrap ? nch : 0));
        }
        P128_STAMP(3);
        __syncthreads();
        P128_STAMP(4);
        // ---- MFMA phase -----------------------------------------------------------------------------------------------------------------
        mma_chunk(tl, c, rb, (team == 0 && fill) ? 4u * nstep : 0u);
        P128_STAMP(5);
        P128_STAMP(6);
        if (!(team == 1 && last_step)) __syncthreads();
        P128_STAMP(7);
        ++stamp_n;
        rb = (rb + 9) & 15;
        ++nstep;
    };
    auto tile_steps = [&](int it) __attribute__((always_inline)) {
        const bool last_tile = it == niter - 1;
        step(it, 0, ic<0>{}, ic<1>{}, false);
        step(it, 1, ic<1>{}, ic<0>{}, last_tile && 2 == nch);
#pragma unroll 1
        for (int c = 2; c < nch; c += 2) {
            step(it, c, ic<0>{}, ic<0>{}, false);
            step(it, c + 1, ic<1>{}, ic<0>{}, last_tile && c + 2 == nch);
        }
    };

    // ---- the walk --------------------------------------------------------------------------------------------------------------------
    if (ntl <= 0) return;
    {
        const Src s0 = prep(cf0, describe(tile_of(0), 0));
#pragma unroll
        for (int i = 0; i < A9; ++i) issue_one(cf0, s0, i);
    }
    dn = describe(tile_of(0), 1);
    __syncthreads();                  // ring slots 0..8 visible
    if (team == 1) __syncthreads();
#pragma unroll 1
    for (int it = 0; it < niter; ++it) tile_steps(it);
    if (live_of(niter - 1)) { issue_res(res, tile_of(niter - 1)); epilogue(res, tile_of(niter - 1)); }
    flush_stats();
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------

static int ilog2_exact128(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

bool conv_pp128_supported(const ConvParams& p, int stride, int up, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_PP128") ? atoi(getenv("PNPFLOW_HIP_PP128")) : 1;      // test-only A/B switch
    if (mode == 0 || terms != 3 || stride != 1 || up != 0 || p.gnb_x != nullptr) return false;
    if (p.Cout != 128 || p.out_cstride != 128 || (p.residual != nullptr && p.res_cstride != 128)) return false;
    if (p.H % P128_TH || p.W % 16 || p.Hs != p.H || p.Ws != p.W || p.W > 2048) return false;
    if (ilog2_exact128(p.H / P128_TH) < 0 || ilog2_exact128(p.W / 16) < 0) return false;
    // the persistent grid pays its prologue and pipeline fill over >= 4 tiles per team (two teams per CU)
    const int grid = persistent_grid();
    if (grid < 8 || (long)p.B * (p.H / P128_TH) * (p.W / 16) < 4L * 2 * grid) return false;
    int nch = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if (s.w_mode != 0 || s.w16 == nullptr || s.C % 16 || s.taps != 9 || s.xform == 0) return false;      // GroupNorm-ed 3x3 segments only
        nch += s.C / 16;
    }
    if (nch < 2 || nch > PP_MAXCH || (nch & 1)) return false;
    return p.gn_C > 0 && p.coef != nullptr;
}

hipError_t launch_conv_pp128(const PPParams& p0, hipStream_t s) {
    if (p0.n9 < 2 || (p0.n9 & 1) || p0.n1 != 0 || p0.cout != 128) return hipErrorInvalidValue;
    static unsigned long long attr_set[2] = {0ull, 0ull};
    const bool res = p0.residual != nullptr;
    const void* kern = res ? reinterpret_cast<const void*>(conv_pp128_kernel<true>) : reinterpret_cast<const void*>(conv_pp128_kernel<false>);
    { hipError_t e = set_max_dynamic_lds_once(kern, attr_set[res ? 1 : 0], 160 * 1024); if (e != hipSuccess) return e; }
    const int grid = persistent_grid();
    if (grid <= 0) return hipErrorInvalidConfiguration;
    PPParams p = p0;
    p.lx = ilog2_exact128(p.W / 16); p.ly = ilog2_exact128(p.H / P128_TH);
    p.rot = 5;
    if (res) hipLaunchKernelGGL(conv_pp128_kernel<true>, dim3(grid), dim3(512), P128_LDS, s, p);
    else hipLaunchKernelGGL(conv_pp128_kernel<false>, dim3(grid), dim3(512), P128_LDS, s, p);
    return hipGetLastError();
}

}  // namespace pf
