// Persistent two-team implicit-GEMM 3x3 convolution for the 64-channel level of the U-Net (Cout = 64: level 1), split-fp16 MFMA,
// fp32-equivalent - the structure of conv_pp.hip (one 512-thread workgroup per CU for the whole launch, two teams of four waves one
// barrier apart, VALU phase / MFMA phase, in-register transposing epilogue) with the changes the wider level needs.
// Reference: the convolutions of ResidualBlock at level width 64 (pnpflow/models.py:58-113).
//   * K is walked in chunks of SIXTEEN input channels (one k16-step per tap): the weights of a 16-channel chunk for both 32-column N-tiles
//     are 36 KiB ([tap][hi | lo][N-tile][k-half][column][8 halfs]), a whole 64 -> 64 layer 144 KiB - it does not fit beside the patches, so
//     the weights live in a TWO-SLOT LDS ring that team 1 refills straight from L2 with LDS-DMA (global_load_lds_dwordx4: no registers,
//     no VALU).  Step u reads slot u % 2.  During ITS MFMA phase of step u team 1 requests the weights of step u + 1 into the other slot
//     (last read by team 1 one step ago, by team 0 a phase before that) - nine 4 KiB rounds spread over the first five taps, behind the
//     matrix pipe - and waits for them (s_waitcnt vmcnt(0)) before the barrier that ends the phase: team 0 reads the slot in the next one;
//   * a wave owns TWO 32-pixel M-tiles x both N-tiles (64 x 64, MT = 2: 8 LDS fragment reads per 12 MFMAs - with 32 x 64 wave tiles,
//     6 reads per 6 MFMAs, the MFMA phase is LDS-bound); the fragments of tap + 1 are requested before the MFMAs of tap are issued;
//   * 64 accumulator registers leave room for ONE raw patch in flight: the prefetch is one step deep, but every register is re-requested
//     the moment it has been staged, so the bytes of step s + 1 fly during the rest of the VALU phase and the whole MFMA phase of step s;
//     the residual-free launches only (a residual tile would need another 64 registers).
// The number of chunks is a run-time value (4 for 64 -> 64, 8 for cat[64, 64] -> 64): the walk is a run-time loop over PAIRS of chunks so
// that a step's parity (its coefficient set and its ring slot) is compile-time; the chunk descriptor of a step (scalar loads with a
// run-time index) is requested at the end of the VALU phase two steps ahead (behind the barrier, not in front of the fragment waits).
// What bounds it (r4 stamps, tools/ubench/conv_pp64_probe.hip): per step a wave issues ~330 staging instructions and 108 MFMAs; the
// staging of one team runs beside the MFMA phase of the other ON THE SAME SIMDs and the two do not overlap freely - the VALU phase takes
// 2.5 k cycles alone and 4-6 k beside the other team's MFMAs (3.5 k matrix-pipe cycles per phase, 4.7-5.2 k measured).
// LDS patch of a team: [8 MT + 2 rows][pitch 20 pixels][4 pieces of 16 B] = (hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15), piece q of the pixel
// in column c at slot q ^ ((c >> 2) & 3); rows are 1280 B = 5 x 256 B apart, so the 16 pixels of a ds_read_b128 lane group (columns
// {0-3, 12-15} of one row + {4-11} of the next, shifted by the tap) fall on 16 different 16-byte bank slots.
#include <cstdlib>
#include "pp_common.h"

namespace pf {

// MT = M-tiles (2 rows x 16 pixels) per wave: a team's tile is 8 MT rows x 16 columns, its patch (8 MT + 2) x 18 pixels, row pitch 20
constexpr int P64_PITCH = 20, P64_PW = 18;
constexpr int p64_npix(int MT) { return (8 * MT + 2) * P64_PW; }
constexpr int p64_patch(int MT) { return (8 * MT + 2) * P64_PITCH * 64; }      // MT 1: 12 800 B, MT 2: 23 040 B per team
constexpr int p64_a9(int MT) { return (p64_npix(MT) * 4 + 255) / 256; }          // float4 per lane and chunk: 3 / 6
constexpr int P64_WSLOT = 36864;                                                 // one 16-channel chunk: 9 taps x 4 KiB
constexpr int p64_lds(int MT) { return 2 * P64_WSLOT + 2 * p64_patch(MT); }      // 99 328 / 119 808 B

// one LDS-DMA instruction: every lane fetches the 16 bytes at sbase + voff into LDS byte address lds_dst + 16 * lane (lds_dst is
// wave-uniform; M0 carries it and is restored: the compiler neither preserves M0 around an asm statement nor expects it changed)
__device__ __forceinline__ void glds16_pp(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

#ifdef PP_PROBE_BUILD
// tools/ubench/conv_pp64_probe.hip only: s_memtime stamps of workgroup 0, [team][step][8]
__device__ unsigned long long* g_pp64_dbg = nullptr;
#define P64_STAMP(k) do { if (stamp_buf != nullptr && blockIdx.x == 0 && t == 0 && stamp_n < 64) stamp_buf[(team * 64 + stamp_n) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define P64_STAMP(k) do { } while (0)
#endif

template <int MT, bool RES>
__global__ __launch_bounds__(512, 2) void conv_pp64_kernel(const PPParams p) {
    constexpr int TH = 8 * MT, NPIX = p64_npix(MT), A9 = p64_a9(MT), P64_PATCH = p64_patch(MT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int team = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int t = tid & 255, lane = t & 63, wm = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned patch0 = (unsigned)(2 * P64_WSLOT) + (unsigned)team * P64_PATCH;
    const pp_float_cptr scale_c = (pp_float_cptr)(uintptr_t)p.scale;      // scalar-cache reads (see conv_pp.hip)
    const int nch = p.n9;

    // ---- weights of chunk 0 -> ring slot 0 (slot 1 is filled by team 1 during its first MFMA phase) -----------------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.ch[0].wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < P64_WSLOT / 16; i += 512) dst[i] = src[i];
    }

    // ---- per-lane constants of the staging -------------------------------------------------------------------------------------------
    // float4 number i of a chunk = patch pixel pp = (t >> 2) + 64 i, channel quad qi = t & 3.  pk: bits 0-15 pixel offset py * W + px inside
    // the patch, bits 20-23 the edges the pixel lies on (top, bottom, left, right); ldsw: byte address of the quad's hi halfs
    const int qi = t & 3, p0 = t >> 2;
    unsigned pk[A9], ldsw[A9];
#pragma unroll
    for (int i = 0; i < A9; ++i) {
        const int pp = min(p0 + 64 * i, NPIX - 1);
        const int py = pp / P64_PW, px = pp - py * P64_PW;
        pk[i] = (unsigned)(py * p.W + px) | ((py == 0 ? 1u : 0u) << 20) | ((py == TH + 1 ? 1u : 0u) << 21) | ((px == 0 ? 1u : 0u) << 22) |
                ((px == P64_PW - 1 ? 1u : 0u) << 23);
        ldsw[i] = patch0 + (unsigned)((py * P64_PITCH + px) * 64) + (unsigned)((((qi >> 1) ^ ((px >> 2) & 3)) << 4) + (qi & 1) * 8);
    }
    constexpr int LASTN = NPIX * 4 - (A9 - 1) * 256;            // threads that own a float4 number A9 - 1
    const int pix_safe = p.W + 1;                               // patch pixel (1, 1) = tile pixel (0, 0): inside the image for every tile

    // A-fragment addresses: lane = pixel (row prow of the M-tile's two rows, column pcol), k-half hi; [kx][term]
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned a_addr[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned base = patch0 + (unsigned)(((wm * 2 * MT + prow) * P64_PITCH + pcol + kx) * 64);
        const unsigned s = (unsigned)(((pcol + kx) >> 2) & 3);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a_addr[kx][tm] = base + ((((unsigned)(hi + 2 * tm)) ^ s) << 4);
    }
    const unsigned b_lane = (unsigned)lane * 16u;
    const unsigned dma_lane = (unsigned)(wm * 1024 + lane * 16);      // this lane's 16 bytes inside a 4 KiB round of the weight refill

    // epilogue geometry: after the transpose the lane holds pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile, channels 32 nt + 4 em .. + 3
    const bool bit3 = (lane & 8) != 0;
    const int em = lane & 7;
    const int ch_of_col = 4 * (l31 & 7) + (l31 >> 3);
    const unsigned e_lane = (unsigned)((4 * hi + ((lane >> 3) & 3)) * 256 + em * 16);      // inside a tile row of 16 pixels x 256 B

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

    // ---- registers of the staging -------------------------------------------------------------------------------------------------------
    // ONE raw patch in flight (prefetch one step ahead, issued float4 by float4 as the VALU phase consumes the current one: every register
    // is re-requested the moment it has been staged, so the bytes of step s + 1 fly during the rest of the VALU phase and the whole MFMA
    // phase of step s).  Two sets would need 2 x 24 registers beside the 64 accumulator registers of the 64 x 64 wave tile.
    float4 ra[A9];
    struct Coef { float4 csc, csh; float ascale; unsigned inval; bool silu; };      // per-step staging coefficients: current / next
    Coef cf0, cf1;
    struct Src { const char* base; unsigned cs4; };                                 // uniform: first patch pixel of the step's chunk, bytes per pixel
    struct Res { float4 rv[MT][2][4]; float addv[2]; };
    Res res;       // residual + bias (+ time-embedding projection) of the tile being closed: requested at the start of the VALU phase that ends with its epilogue
    const unsigned q16 = (unsigned)qi * 16u;

    // descriptor of a step (uniform: scalar loads from the kernel arguments with a run-time chunk index, two levels deep for the operand
    // scale): fetched during the MFMA phase one step ahead, consumed by the VALU phase that requests the step's patch
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

    // one accumulator per (M-tile, N-tile): 2 MT independent chains for the matrix pipe
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
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
                double* dst = p.stats_out + ((size_t)run_b * 64 + (j >> 2) * 32 + em * 4 + (j & 3)) * 2;
                unsafeAtomicAdd(dst, a); unsafeAtomicAdd(dst + 1, q);
            }
            run1[j] = 0.f; run2[j] = 0.f;
        }
        run_n = 0;
    };

    auto epilogue = [&](const Res& R, const PPTile& tl) __attribute__((always_inline)) {
        if (tl.b != run_b || run_n >= 32) { flush_stats(); run_b = tl.b; }
        ++run_n;
        const float inv_last = p.scale != nullptr ? scale_c[8 * tl.b + 4 + p.ch[nch - 1].seg] : 1.0f;
        const float oscale = p.out_scale * (1.0f / 256.0f) * inv_last;
        char* obase = reinterpret_cast<char*>(p.out + (((size_t)tl.b * p.H + tl.oy0 + wm * 2 * MT) * p.W + tl.ox0) * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
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
                        const float rsc = p.res_scale; const float4 r4 = R.rv[mt][nt][g];
                        v.x = fmaf(r4.x, rsc, v.x); v.y = fmaf(r4.y, rsc, v.y); v.z = fmaf(r4.z, rsc, v.z); v.w = fmaf(r4.w, rsc, v.w);
                    }
                    // pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile: row (g >> 1) of its two rows, column 8 (g & 1) + 4 hi + ((lane >> 3) & 3)
                    char* dst = obase + (size_t)((mt * 2 + (g >> 1)) * p.W) * 256 + (g & 1) * 2048 + nt * 128 + e_lane;
                    nt_store4(dst, v);          // streamed (r4: -2.6 % per launch)
                    run1[nt * 4 + 0] += v.x; run1[nt * 4 + 1] += v.y; run1[nt * 4 + 2] += v.z; run1[nt * 4 + 3] += v.w;
                    run2[nt * 4 + 0] += v.x * v.x; run2[nt * 4 + 1] += v.y * v.y; run2[nt * 4 + 2] += v.z * v.z; run2[nt * 4 + 3] += v.w * v.w;
                }
            }
    };

    auto issue_res = [&](Res& R, const PPTile& tl) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) R.addv[nt] = p.addvec != nullptr ? p.addvec[(size_t)tl.b * p.addvec_bs + nt * 32 + ch_of_col] : 0.f;
        if constexpr (RES) {
            const char* rbase = reinterpret_cast<const char*>(p.residual + (((size_t)tl.b * p.H + tl.oy0 + wm * 2 * MT) * p.W + tl.ox0) * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        R.rv[mt][nt][g] = *reinterpret_cast<const float4*>(rbase + (size_t)((mt * 2 + (g >> 1)) * p.W) * 256 + (g & 1) * 2048 + nt * 128 + e_lane);
        }
    };

    // MFMA phase of chunk c from ring slot SLOT: per tap one A fragment pair (hi, lo) and the B fragment pairs of both N-tiles
    auto mma_chunk = [&](const PPTile& tl, int c, auto SLOT_, const char* refill_src) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOT_)::value;
        if (c == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        } else if (p.scale != nullptr && p.ch[c - 1].seg != p.ch[c].seg) {
            // the accumulator changes units: from the previous K-segment's operand scale to this one's (powers of two: exact)
            const float ratio = scale_c[8 * tl.b + p.ch[c].seg] * scale_c[8 * tl.b + 4 + p.ch[c - 1].seg];
            if (ratio != 1.0f) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= ratio;
            }
        }
        const unsigned wbase = (unsigned)(SLOT * P64_WSLOT) + b_lane;
        __builtin_amdgcn_s_setprio(1);
        // fragments of tap + 1 are requested before the MFMAs of tap are issued (two named sets): while a team is in this phase its SIMD has
        // no other wave to cover an LDS round trip - requested and consumed tap by tap, the 9 round trips cost more than the 108 MFMAs
        // (r4: 8.5 k cycles per phase against 3.5 k of matrix-pipe time)
        f16x8 fa[2][MT][2], fb[2][2][2];
        auto load_tap = [&](int set, int tap) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned off = (unsigned)((mt * 2 + ky) * P64_PITCH * 64);
                fa[set][mt][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][0] + off);
                fa[set][mt][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][1] + off);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                fb[set][nt][0] = *reinterpret_cast<const f16x8*>(smem + wbase + (unsigned)(((tap * 2 + 0) * 2 + nt) * 1024));
                fb[set][nt][1] = *reinterpret_cast<const f16x8*>(smem + wbase + (unsigned)(((tap * 2 + 1) * 2 + nt) * 1024));
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) load_tap(cur ^ 1, tap + 1);
            // ring refill (team 1): the nine 4 KiB rounds of the NEXT step's weights go out during the first five taps, into the slot both
            // teams finished with a phase ago; the remaining taps (>= 1.5 k cycles) cover the last round's L2 round trip
            if (refill_src != nullptr && tap < 5) {
#pragma unroll
                for (int r = 2 * tap; r < 2 * tap + 2 && r < 9; ++r)
                    glds16_pp(dma_lane + (unsigned)(r * 4096), refill_src, (unsigned)((SLOT ^ 1) * P64_WSLOT + r * 4096) + (unsigned)(wm * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- one step = VALU phase + MFMA phase of chunk c of the team's it-th tile; SI = c & 1 (nch is even: the step's global parity, i.e.
    // its coefficient set AND its ring slot) ----------------------------------------------------------------------------------------------
    int stamp_n = 0; (void)stamp_n;
#ifdef PP_PROBE_BUILD
    unsigned long long* const stamp_buf = g_pp64_dbg;      // null: no stamps (the probe's plain timing runs)
#endif
    Desc dn;            // descriptor of the step AFTER the current one
    auto step = [&](int it, int c, auto SI_, bool last_step) __attribute__((always_inline)) {
        constexpr int SI = decltype(SI_)::value;
        Coef& C = SI == 0 ? cf0 : cf1;
        Coef& N = SI == 0 ? cf1 : cf0;
        const PPTile tl = tile_of(it);
        // ---- VALU phase: stage this step's patch, re-request every register for step + 1 as it is staged; on the first step of a tile close
        // the previous one (its residual is requested first and lands while the patch is staged) -------------------------------------------
        P64_STAMP(0);
        bool close = false;
        if constexpr (SI == 0) close = c == 0 && it > 0 && live_of(it - 1);
        if (close) issue_res(res, tile_of(it - 1));
        const Src srN = prep(N, dn);
        P64_STAMP(1);
#pragma unroll
        for (int i = 0; i < A9; ++i) { transform_one(C, i); issue_one(N, srN, i); }
        P64_STAMP(2);
        if (close) epilogue(res, tile_of(it - 1));
        // the descriptor of step + 2 (scalar loads with a run-time chunk index, two levels deep for the operand scale), requested here so that
        // the round trips pass behind the barrier: scalar loads share their counter with LDS reads, at the head of the MFMA phase they
        // would have to be drained before the first fragment wait
        {
            const int wrap = c + 2 >= nch ? 1 : 0;
            dn = describe(tile_of(it + wrap), c + 2 - (wrap ? nch : 0));
        }
        P64_STAMP(3);
        __syncthreads();
        P64_STAMP(4);
        // ---- MFMA phase -----------------------------------------------------------------------------------------------------------------
        // team 1 keeps the weight ring: during this phase it fetches the weights of step + 1 into the other slot (last read by this team one
        // step ago, by team 0 a phase before that); team 0 reads that slot in the NEXT phase, so the refill must have landed when team 1
        // reaches the barrier (everything else the wait covers was requested a whole phase earlier)
        const int c1 = c + 1 == nch ? 0 : c + 1;
        mma_chunk(tl, c, SI_, (team == 1 && !last_step) ? reinterpret_cast<const char*>(p.ch[c1].wimg) : nullptr);
        P64_STAMP(5);
        if (team == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P64_STAMP(6);
        if (!(team == 1 && last_step)) __syncthreads();
        P64_STAMP(7);
        ++stamp_n;
    };
    auto tile_steps = [&](int it) __attribute__((always_inline)) {
        const bool last_tile = it == niter - 1;
#pragma unroll 1
        for (int c = 0; c < nch; c += 2) {
            step(it, c, ic<0>{}, false);
            step(it, c + 1, ic<1>{}, last_tile && c + 2 == nch);
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
    __syncthreads();                  // ring slots visible
    if (team == 1) __syncthreads();
#pragma unroll 1
    for (int it = 0; it < niter; ++it) tile_steps(it);
    if (live_of(niter - 1)) { issue_res(res, tile_of(niter - 1)); epilogue(res, tile_of(niter - 1)); }
    flush_stats();
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------

static int ilog2_exact64(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

bool conv_pp64_supported(const ConvParams& p, int stride, int up, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_PP64") ? atoi(getenv("PNPFLOW_HIP_PP64")) : 1;
    if (mode == 0 || terms != 3 || stride != 1 || up != 0 || p.gnb_x != nullptr) return false;
    // launches with an identity residual stay on conv_mfma16: the 64 x 64 wave tile + the residual tile do not fit 256 registers, and with
    // 32 x 64 wave tiles (MT = 1: 6 LDS fragment reads per 6 MFMAs) this kernel is slower than conv_mfma16 (r4: 375 vs 358 us, 80 x 128^2)
    if (p.Cout != 64 || p.out_cstride != 64 || p.residual != nullptr) return false;
    if (p.H % 16 || p.W % 16 || p.Hs != p.H || p.Ws != p.W) return false;
    if (ilog2_exact64(p.H / 16) < 0 || ilog2_exact64(p.W / 16) < 0) return false;
    // the persistent grid pays its prologue and pipeline fill over >= 4 tiles of 16 x 16 pixels per team
    const int grid = persistent_grid();
    if (grid < 8 || (long)p.B * (p.H / 16) * (p.W / 16) < 4L * 2 * grid) return false;
    if (p.W > 2048) return false;      // patch pixel offsets py * W + px are packed into 16 bits (pk[])
    int nch = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if (s.w_mode != 0 || s.w16 == nullptr || s.C % 16 || s.taps != 9 || s.xform == 0) return false;      // GroupNorm-ed 3x3 segments only
        nch += s.C / 16;
    }
    if (nch < 2 || nch > PP_MAXCH || (nch & 1)) return false;
    return p.gn_C > 0 && p.coef != nullptr;
}

// (the kernel is written for MT = 1 / 2 and with / without residual; the library launches the one shape that wins: 16 x 16-pixel team tiles,
// no residual)
hipError_t launch_conv_pp64(const PPParams& p0, hipStream_t s) {
    if (p0.n9 < 2 || (p0.n9 & 1) || p0.n1 != 0 || p0.cout != 64 || p0.residual != nullptr) return hipErrorInvalidValue;
    static unsigned long long attr_set = 0ull;
    auto kern = conv_pp64_kernel<2, false>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    const int grid = persistent_grid();
    if (grid <= 0) return hipErrorInvalidConfiguration;
    PPParams p = p0;
    p.lx = ilog2_exact64(p.W / 16); p.ly = ilog2_exact64(p.H / 16);
    p.rot = 5;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), p64_lds(2), s, p);
    return hipGetLastError();
}

}  // namespace pf
