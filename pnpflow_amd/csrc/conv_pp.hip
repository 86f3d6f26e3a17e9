// Persistent two-team ("ping-pong") implicit-GEMM convolution for the full-resolution levels of the U-Net (Cout = 32: level 0),
// split-fp16 MFMA, fp32-equivalent - the same products as conv_mfma16.hip (summed in two interleaved fp32 chains instead of one: the
// outputs agree with it to fp32 rounding, 4e-7 relative on a whole forward).  Reference: the convolutions of ResidualBlock at level width 32 (pnpflow/models.py:58-113).
//
// Why a second structure (profiles/r03_pmc_level0_counters.md, r03_level0_bound_ab.md): at this level the per-tile floor is HBM
// (~100 KB per 256-pixel tile against 3.5 k matrix-pipe cycles), yet conv_mfma16_kernel spends 12.8 k cycles per tile - every workgroup
// is a serial chain (first loads -> transform -> barrier -> 54 MFMAs per wave fed by L2 weight fragments -> transposing epilogue), each
// wave re-fetches the same 36 KB of weights through the texture path per tile, and the generic K-segment walk costs 1.8 k VALU + 1.5 k
// SALU instructions per wave per tile.  Here:
//   * ONE 512-thread workgroup per CU lives for the whole launch and walks a contiguous range of 16 x 16-pixel tiles;
//   * the weights of the launch ([chunk][k16-step][hi | lo][k-half][column][8 halfs], column 8 g + k = output channel 4 k + g: one lane-linear 1 KiB block per B fragment) are
//     copied to LDS ONCE per workgroup and every B fragment is a ds_read_b128 with an immediate offset;
//   * the 8 waves form two teams of 4.  A team alternates a VALU phase (epilogue of its previous tile, then GroupNorm + SiLU + operand
//     scale + fp16 hi / lo split of the raw fp32 halo patch that arrived in registers -> its LDS patch) with an MFMA phase (requests
//     the NEXT patch + this tile's residual, then walks the chunk's k16-steps from LDS).  The teams are one workgroup barrier apart, so
//     on every SIMD one wave is in its matrix phase while the other transforms / stores - complementary by construction, and the
//     next tile's bytes are in flight for a whole phase before they are needed;
//   * everything that does not depend on the tile (per-lane patch offsets, swizzled LDS addresses of every A fragment, border masks)
//     is computed once per launch: the k-loop has no VALU / SALU address work at all (immediate offsets only);
//   * the epilogue transposes each 32 x 32 accumulator tile across lanes (v_permlane16_swap + a DPP row rotation) instead of through LDS:
//     a lane gets four consecutive channels of one pixel, eight neighbouring lanes the pixel's whole 128-byte row - no scratch, no LDS traffic;
//   * GroupNorm statistics are summed per lane in fp64 across the tiles of an image and leave as one atomic per (wave, channel) when
//     the image changes (conv_mfma16: one per (workgroup, channel) per tile).
// LDS patch of a team: [18 x 18 pixels][8 pieces of 16 B] = (hi k0-7 | hi k8-15 | hi k16-23 | hi k24-31 | lo ...) with piece q of the
// pixel in column c stored at slot q ^ ((c >> 1) & 7): the 16 pixels a ds_read_b128 lane group touches ({0-3, 12-15} of one tile row +
// {4-11} of the next, shifted by the tap) land on 16 different 16-byte bank slots for every tap (the swizzle conv_dma.hip uses).
#include <cstdlib>
#include "pp_common.h"

namespace pf {

// MT = M-tiles (2 rows x 16 pixels) per wave: a team's tile is 8 MT rows x 16 columns, its halo patch (8 MT + 2) x 18 pixels x 128 B
constexpr int PP_PW = 18;
constexpr int pp_w9(int TERMS) { return TERMS == 3 ? 36864 : 18432; }      // LDS bytes of a 9-tap / 1-tap 32-channel chunk image ([k16-step][hi | lo] or [k16-step][hi])
constexpr int pp_w1(int TERMS) { return TERMS == 3 ? 4096 : 2048; }
constexpr int pp_npix(int MT) { return (8 * MT + 2) * PP_PW; }
constexpr int pp_patch_bytes(int MT) { return pp_npix(MT) * 128; }                         // MT 1: 23 040 B, MT 2: 41 472 B per team
constexpr int pp_a9(int MT) { return (pp_npix(MT) * 8 + 255) / 256; }                      // float4 per lane of a 9-tap chunk: 6 / 11

// The K-chunk structure is compile-time: N9 nine-tap chunks (GroupNorm(+SiLU) staging) followed by N1 one-tap chunks (raw: a folded 1x1
// shortcut), NCH = N9 + N1 steps per tile.  Register prefetch is TWO steps deep: the patch of step s + 2 is requested at the end of the
// VALU phase of step s into the register set that phase has just staged (sets alternate with the step parity, so the tile loop is unrolled
// by two and every register has a static name), the residual of a tile two steps before the epilogue that adds it.  One step of distance
// was not enough: the burst had a single MFMA phase (~0.9 us) to land and every VALU phase began with ~1 us of exposed latency
// (r4: on-chip 78 us + loads 78 us + stores 97 us were ADDITIVE, 267 us against a 132 us memory skeleton).
#ifdef PP_PROBE_BUILD
// tools/ubench/conv_pp_probe.hip only: s_memtime stamps of workgroup 0, [team][step][8]
__device__ unsigned long long* g_pp_dbg = nullptr;
__device__ int g_pp_dbg_skip = 0;      // steps of workgroup 0 that pass before the 64 stamped ones (read once per kernel: a per-stamp read perturbs every phase)
#define PP_STAMP(k) do { if (stamp_buf != nullptr && blockIdx.x == 0 && t == 0 && stamp_n >= 0 && stamp_n < 64) stamp_buf[(team * 64 + stamp_n) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif

// TERMS = 3: the fp32-equivalent split (a_lo w_hi + a_hi w_lo + a_hi w_hi); TERMS = 1 (round 6): precision mode 2 - operands rounded once to fp16 (hi only:
// the products of conv_mfma16's TERMS = 1 form), one MFMA per k16-step, hi-only weight images in LDS (half the footprint: every chunk structure fits two
// workgroups per CU) and hi-only patch records.  Under the board's power cap the two dropped MFMAs are the saving.
template <int MT, int N9, int N1, bool RES, int TEAMS = 2, int TERMS = 3>
__global__ __launch_bounds__(256 * TEAMS, 2) void conv_pp_kernel(const PPParams p) {
    constexpr int NCH = N9 + N1;
    constexpr int W9 = pp_w9(TERMS), W1 = pp_w1(TERMS), WSTEP = TERMS == 3 ? 2048 : 1024;      // LDS bytes of a 9-tap / 1-tap chunk image, of one k16-step
    static_assert(TERMS == 3 || TERMS == 1, "split terms");
    constexpr int PP_NPIX = pp_npix(MT), PP_PATCH_BYTES = pp_patch_bytes(MT), PP_A9 = pp_a9(MT), TH = 8 * MT;
    constexpr int LASTN = PP_NPIX * 8 - (PP_A9 - 1) * 256;          // threads that own a float4 number PP_A9 - 1
    constexpr int WBYTES = N9 * W9 + N1 * W1;                       // LDS weight images: 9-tap chunks first
    static_assert(NCH >= 1 && NCH <= PP_MAXCH && N9 >= 1, "chunk structure");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int team = TEAMS == 2 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;
    const int t = tid & 255, lane = t & 63, wm = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned patch0 = (unsigned)WBYTES + (unsigned)team * PP_PATCH_BYTES;      // byte offset of this team's patch in LDS
    // the per-image operand scales are read through the scalar cache (constant address space): as ordinary global loads hipcc makes them
    // VECTOR loads, and every wait for one drains the patch prefetch (one in-order vmcnt)
    const pp_float_cptr scale_c = (pp_float_cptr)(uintptr_t)p.scale;
    auto wlds = [](int c) constexpr { return c < N9 ? c * W9 : N9 * W9 + (c - N9) * W1; };

    // ---- weights -> LDS, once per workgroup --------------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint4* src = reinterpret_cast<const uint4*>(p.ch[c].wimg);
        uint4* dst = reinterpret_cast<uint4*>(smem + wlds(c));
        const int n16 = (c < N9 ? 9 : 1) * 256;             // taps * 2 k16-steps * 2 KiB / 16
        if constexpr (TERMS == 3) {
            for (int i = tid; i < n16; i += 256 * TEAMS) dst[i] = src[i];
        } else {
            // hi halves only: the global image is [k16-step][hi 1 KiB | lo 1 KiB]
            for (int i = tid; i < n16 / 2; i += 256 * TEAMS) dst[i] = src[(i >> 6) * 128 + (i & 63)];
        }
    }

    // ---- per-lane constants of the staging (independent of the tile) -----------------------------------------------------------------
    const int qi = t & 7, p0 = t >> 3;
    // float4 number i of a 9-tap chunk = patch pixel pp = p0 + 32 i, channel quad qi.  One packed word per i: bits 0-15 pixel offset
    // py * W + px inside the patch (pixels), bits 16-18 the LDS slot (piece ^ swizzle) of the quad's hi halfs, bits 20-23 the edges the
    // pixel lies on (top, bottom, left, right)
    unsigned pk9[PP_A9];
#pragma unroll
    for (int i = 0; i < PP_A9; ++i) {
        const int pp = min(p0 + 32 * i, PP_NPIX - 1);
        const int py = pp / PP_PW, px = pp - py * PP_PW;
        pk9[i] = (unsigned)(py * p.W + px) | ((unsigned)((qi >> 1) ^ ((px >> 1) & 7)) << 16) |
                 ((py == 0 ? 1u : 0u) << 20) | ((py == TH + 1 ? 1u : 0u) << 21) | ((px == 0 ? 1u : 0u) << 22) | ((px == PP_PW - 1 ? 1u : 0u) << 23);
    }
    const unsigned ldsw9 = patch0 + (unsigned)min(p0, PP_NPIX - 1) * 128u + (unsigned)((qi & 1) * 8);       // + 4096 i + (slot << 4)
    const int pix_safe = p.W + 1;                          // patch pixel (1, 1) = tile pixel (0, 0): inside the image for every tile
    // 1-tap chunks stage the TH x 16 interior only: float4 number t + 256 i = pixel (row (t >> 7) + 2 i, column (t >> 3) & 15), quad qi
    const int r1 = t >> 7, c1 = (t >> 3) & 15;
    const int pixoff1 = r1 * p.W + c1;
    const unsigned ldsw1 = patch0 + (unsigned)((r1 + 1) * PP_PW + c1 + 1) * 128u + (unsigned)((((qi >> 1) ^ (((c1 + 1) >> 1) & 7)) << 4) + (qi & 1) * 8);

    // A-fragment addresses: lane = pixel (row prow of the M-tile's two rows, column pcol), k-half hi; [kx][j][term]
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned a_addr[3][2][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const unsigned base = patch0 + (unsigned)((wm * 2 * MT + prow) * PP_PW + pcol + kx) * 128u;
        const unsigned s = (unsigned)(((pcol + kx) >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) a_addr[kx][j][tm] = base + ((((unsigned)(j * 2 + hi + 4 * tm)) ^ s) << 4);
    }
    const unsigned b_lane = (unsigned)lane * 16u;

    // epilogue geometry: after the transpose the lane holds pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile, channels 4 em .. 4 em + 3
    const bool bit3 = (lane & 8) != 0;
    const int em = lane & 7;
    const int ch_of_col = 4 * (l31 & 7) + (l31 >> 3);                                // output channel carried by MFMA column l31
    const unsigned e_lane = (unsigned)((4 * hi + ((lane >> 3) & 3)) * 128 + em * 16);  // byte offset inside a tile row of 16 pixels x 128 B

    // ---- this workgroup's tiles ----------------------------------------------------------------------------------------------------
    const int G = gridDim.x;
    const int rg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);                      // XCD-contiguous ranges (G % 8 == 0)
    const int T = p.B << (p.lx + p.ly);
    const int t_begin = (int)((long)rg * T / G), t_end = (int)((long)(rg + 1) * T / G);
    const int ntl = t_end - t_begin;
    const int niter = (ntl + TEAMS - 1) / TEAMS;
    const int rot = ntl > 0 ? (int)(((long)rg * p.rot) % ntl) : 0;
    auto tile_of = [&](int it) __attribute__((always_inline)) -> PPTile {
        // past the range: the last tile again (its stores are masked).  The walk starts `rot` tiles into the range and wraps: neighbouring
        // CUs, whose ranges begin a whole number of tile rows apart, would otherwise sit on the same column tile - the same address bits
        // below the row pitch - at every moment of the launch
        const int idx = min(it * TEAMS + team, ntl - 1);
        const int wrapped = idx + rot >= ntl ? idx + rot - ntl : idx + rot;
        const int tl = t_begin + wrapped;
        PPTile r;
        const int tx = tl & ((1 << p.lx) - 1), ty = (tl >> p.lx) & ((1 << p.ly) - 1);
        r.b = tl >> (p.lx + p.ly); r.oy0 = ty * TH; r.ox0 = tx * 16;
        r.edge = (ty == 0 ? 1 : 0) | (ty == (1 << p.ly) - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == (1 << p.lx) - 1 ? 8 : 0);
        return r;
    };
    auto live_of = [&](int it) __attribute__((always_inline)) -> bool { return it >= 0 && (t_begin + it * TEAMS + team) < t_end; };

    // ---- register sets ---------------------------------------------------------------------------------------------------------------
    struct Pre {
        float4 ra[PP_A9];             // raw fp32 patch of a chunk, in flight
        float4 csc, csh;              // GroupNorm coefficients of this lane's channel quad for that chunk
        float ascale;                 // operand scale of the chunk's K-segment
        unsigned inval;               // bit i: float4 i lies outside the image (loaded from a safe pixel, staged as zero)
    };
    Pre pre0, pre1;
    struct Res { float4 rv[MT][4]; float addv; };      // residual + bias (+ time-embedding projection) of one tile
    // The library instantiates (and the tests cover) 8 x 16-pixel team tiles only: MT stays a parameter of the tile geometry, the prefetch
    // distances below are the MT = 1 ones (16 x 16 tiles spilled 57-156 registers with this two-step prefetch: +30 ... +70 % per launch, r4)
    static_assert(MT == 1, "conv_pp_kernel: the residual / accumulator scheme below is the one of 8 x 16-pixel team tiles");
    // residual of a tile: requested RD = 2 steps before the epilogue that adds it (as the patches: a step is short); with one chunk per tile
    // two residuals are then in flight (two sets, by tile parity); with more chunks the tile before has been closed when the request is issued
    constexpr int RD = 2;
    constexpr int NRES = NCH == 1 ? 2 : 1;
    Res res0, res1;      // (named objects, not an array: hipcc's counted vmcnt waits degrade to vmcnt(4) / vmcnt(0) when the sets are array elements)

    // (the prologue's requests: the first two steps' patches)
    auto issue_patch = [&](Pre& S, const PPTile& tl, auto C_) __attribute__((always_inline)) {
        constexpr int C = decltype(C_)::value;
        const long bpix = ((long)tl.b * p.H + tl.oy0) * p.W + tl.ox0;
        const int cstride = p.ch[C].cstride;
        S.ascale = p.scale != nullptr ? scale_c[8 * tl.b + p.ch[C].seg] : 1.0f;
        // addresses = one uniform 64-bit base per step (scalar unit) + a 32-bit per-lane byte offset: pixel offset x (cstride * 4) as a
        // 24-bit multiply-add (v_mad_u32_u24, full rate; the generic 64-bit form cost a v_mul_lo_u32 + v_lshl_add_u64 per load)
        const unsigned cs4 = (unsigned)cstride * 4u, q16 = (unsigned)qi * 16u;
        if constexpr (C < N9) {
            const char* base = reinterpret_cast<const char*>(p.ch[C].src + (bpix - p.W - 1) * cstride + p.ch[C].coff);
            unsigned inval = 0;
#pragma unroll
            for (int i = 0; i < PP_A9; ++i) inval |= (((pk9[i] >> 20) & (unsigned)tl.edge) != 0u ? 1u : 0u) << i;
            S.inval = inval;
            const char* cb = reinterpret_cast<const char*>(p.coef + (size_t)tl.b * 2 * p.coef_stride + p.ch[C].gn_c0);
            S.csc = *reinterpret_cast<const float4*>(cb + q16); S.csh = *reinterpret_cast<const float4*>(cb + (unsigned)(p.coef_stride * 4) + q16);
#pragma unroll
            for (int i = 0; i < PP_A9; ++i) {
                const unsigned px = ((inval >> i) & 1u) ? (unsigned)pix_safe : (pk9[i] & 0xffffu);
                S.ra[i] = *reinterpret_cast<const float4*>(base + (__umul24(px, cs4) + q16));      // (not streamed: neighbouring tiles re-read the halo from L2; nt here measured +1 %)
            }
        } else {
            const char* base = reinterpret_cast<const char*>(p.ch[C].src + bpix * cstride + p.ch[C].coff);
            S.inval = 0;
#pragma unroll
            for (int i = 0; i < TH / 2; ++i) S.ra[i] = *reinterpret_cast<const float4*>(base + (__umul24((unsigned)(pixoff1 + 2 * i * p.W), cs4) + q16));
        }
    };

    auto split_store = [&](float4 v, unsigned addr) __attribute__((always_inline)) {
        if constexpr (TERMS == 3) {
            uint2 h, l;
            split4_pp(v, h.x, h.y, l.x, l.y);
            *reinterpret_cast<uint2*>(smem + addr) = h;
            *reinterpret_cast<uint2*>(smem + (addr ^ 64u)) = l;                       // the lo piece q + 4 sits at slot (q ^ s) ^ 4
        } else {
            uint2 h;      // one rounding to fp16 (RNE), as conv_mfma16's TERMS = 1 staging; the record keeps its 128-byte pitch, the lo half stays unwritten and unread
            asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(h.x), "=&v"(h.y) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            *reinterpret_cast<uint2*>(smem + addr) = h;
        }
    };

    // staging of step (it, C) from set S interleaved with the requests of step + 2 into the same set: every register is re-requested the
    // moment it has been staged, so the VALU work of the staging runs while the vector-memory queue accepts the requests (issued as one
    // burst behind the staging they stalled for 1.5-2.7 k cycles, r4 stamps; interleaved: -0.6 ... -3 % per launch, bit-identical); the
    // set's scalars and coefficients are replaced last
    auto stage_and_request = [&](Pre& S, auto C_, const PPTile& t2, auto C2_) __attribute__((always_inline)) {
        constexpr int C = decltype(C_)::value, C2 = decltype(C2_)::value;
        constexpr int NT_ = C < N9 ? PP_A9 : TH / 2, NL_ = C2 < N9 ? PP_A9 : TH / 2;
        const long bpix = ((long)t2.b * p.H + t2.oy0) * p.W + t2.ox0;
        const int cstride = p.ch[C2].cstride;
        const unsigned cs4 = (unsigned)cstride * 4u, q16 = (unsigned)qi * 16u;
        const char* base = reinterpret_cast<const char*>(p.ch[C2].src + (bpix - (C2 < N9 ? p.W + 1 : 0)) * cstride + p.ch[C2].coff);
        unsigned inval2 = 0;
        if constexpr (C2 < N9) {
#pragma unroll
            for (int i = 0; i < PP_A9; ++i) inval2 |= (((pk9[i] >> 20) & (unsigned)t2.edge) != 0u ? 1u : 0u) << i;
        }
        const bool silu = C < N9 ? p.ch[C].xform == 2 : false;
#pragma unroll
        for (int i = 0; i < (NT_ > NL_ ? NT_ : NL_); ++i) {
            if (i < NT_) {
                float4 v = S.ra[i];
                if constexpr (C < N9) {
                    v.x = v.x * S.csc.x + S.csh.x; v.y = v.y * S.csc.y + S.csh.y; v.z = v.z * S.csc.z + S.csh.z; v.w = v.w * S.csc.w + S.csh.w;
                    if (silu) silu4_pp(v);
                    const float f = ((S.inval >> i) & 1u) ? 0.0f : S.ascale;
                    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                    if (i < PP_A9 - 1 || t < LASTN) split_store(v, ldsw9 + (unsigned)(i * 4096) + (((pk9[i] >> 16) & 7u) << 4));
                } else {
                    v.x *= S.ascale; v.y *= S.ascale; v.z *= S.ascale; v.w *= S.ascale;
                    split_store(v, ldsw1 + (unsigned)(2 * i * PP_PW * 128));
                }
            }
            if (i < NL_) {
                if constexpr (C2 < N9) {
                    const unsigned px = ((inval2 >> i) & 1u) ? (unsigned)pix_safe : (pk9[i] & 0xffffu);
                    S.ra[i] = *reinterpret_cast<const float4*>(base + (__umul24(px, cs4) + q16));
                } else {
                    S.ra[i] = *reinterpret_cast<const float4*>(base + (__umul24((unsigned)(pixoff1 + 2 * i * p.W), cs4) + q16));
                }
            }
        }
        S.inval = inval2;
        S.ascale = p.scale != nullptr ? scale_c[8 * t2.b + p.ch[C2].seg] : 1.0f;
        if constexpr (C2 < N9) {
            const char* cb = reinterpret_cast<const char*>(p.coef + (size_t)t2.b * 2 * p.coef_stride + p.ch[C2].gn_c0);
            S.csc = *reinterpret_cast<const float4*>(cb + q16); S.csh = *reinterpret_cast<const float4*>(cb + (unsigned)(p.coef_stride * 4) + q16);
        }
    };

    // two accumulators per M-tile (even / odd k16-steps, summed in the epilogue): with one, the 54 MFMAs of a chunk are a single dependent
    // chain and the wave stalls on every issue (r4 counters: SQ_WAIT_INST_ANY 19 % of the wave's cycles)
    constexpr int NACC = 2;
    f32x16 acc[NACC][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { for (int a = 0; a < NACC; ++a) acc[a][mt][r] = 0.f; }
    // running (sum, sum of squares) of this lane's four output channels over the tiles of image run_b: fp32 per lane (at most 32 tiles x 4
    // pixels between flushes - conv_mfma16 sums 256 pixels per channel in fp32 before its fp64 atomic), reduced across the wave's 8 lanes per
    // channel quad and added to the fp64 statistics when the image changes.  (Reducing and accumulating per TILE - shuffles + an LDS fp64
    // read-modify-write - was 2.5 k cycles of every VALU phase together with the transposes, r4 stamps.)
    float run1[4] = {0.f, 0.f, 0.f, 0.f}, run2[4] = {0.f, 0.f, 0.f, 0.f};
    int run_b = -1, run_n = 0;

    // one k16-step: A fragments (hi, lo) of the wave's M-tiles + B fragments (hi, lo) from LDS, three MFMAs per M-tile into the accumulator
    // chain of the step's parity.  (Fragments requested one / two steps ahead in named register sets shorten the MFMA phase from 3.5 k to
    // 2.6 k cycles - r4 stamps - and change nothing end to end: the VALU phase and the memory queue are the critical path; not kept, the
    // registers are needed by the two prefetch sets.)
    auto mma_step = [&](unsigned wbase, int s, int ky, int kx, int j) __attribute__((always_inline)) {
        const int ai = (s & 1) % NACC;
        if constexpr (TERMS == 3) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(smem + wbase + (unsigned)((s * 2 + 0) * 1024));
            const f16x8 bl = *reinterpret_cast<const f16x8*>(smem + wbase + (unsigned)((s * 2 + 1) * 1024));
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned off = (unsigned)((mt * 2 + ky) * PP_PW * 128);
                ah[mt] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][j][0] + off);
                al[mt] = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][j][1] + off);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ai][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[ai][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ai][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[ai][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ai][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[ai][mt], 0, 0, 0);
        } else {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(smem + wbase + (unsigned)(s * WSTEP));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + a_addr[kx][j][0] + (unsigned)((mt * 2 + ky) * PP_PW * 128));
                acc[ai][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ai][mt], 0, 0, 0);
            }
        }
    };

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

    auto epilogue = [&](const Res& R, const PPTile& tl, bool live) __attribute__((always_inline)) {
        if (!live) return;                                   // (the first VALU phase has no finished tile; a repeated last tile is not stored twice)
        if (tl.b != run_b || run_n >= 32) { flush_stats(); run_b = tl.b; }
        ++run_n;
        const float inv_last = p.scale != nullptr ? scale_c[8 * tl.b + 4 + p.ch[NCH - 1].seg] : 1.0f;
        const float oscale = p.out_scale * (1.0f / 256.0f) * inv_last;
        float* obase = p.out + (((size_t)tl.b * p.H + tl.oy0 + wm * 2 * MT) * p.W + tl.ox0) * 32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = (NACC == 2 ? acc[0][mt][r] + acc[NACC - 1][mt][r] : acc[0][mt][r]) * oscale + R.addv;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                oct_transpose(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3], bit3);
                float4 v = make_float4(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
                if constexpr (RES) {
                    const float rsc = p.res_scale; const float4 r4 = R.rv[mt][g];
                    v.x = fmaf(r4.x, rsc, v.x); v.y = fmaf(r4.y, rsc, v.y); v.z = fmaf(r4.z, rsc, v.z); v.w = fmaf(r4.w, rsc, v.w);
                }
                // pixel 8 g + 4 hi + ((lane >> 3) & 3) of the M-tile: row (g >> 1) of its two rows, column 8 (g & 1) + 4 hi + ((lane >> 3) & 3)
                char* dst = reinterpret_cast<char*>(obase) + (size_t)((mt * 2 + (g >> 1)) * p.W) * 128 + (g & 1) * 1024 + e_lane;
                nt_store4(dst, v);         // streamed: the next conv reads the tensor after this launch has written all of it (r4: -2.4 % per launch)
                run1[0] += v.x; run1[1] += v.y; run1[2] += v.z; run1[3] += v.w;
                run2[0] += v.x * v.x; run2[1] += v.y * v.y; run2[2] += v.z * v.z; run2[3] += v.w * v.w;
            }
        }
    };

    auto issue_res = [&](Res& R, const PPTile& tl) __attribute__((always_inline)) {
        R.addv = p.addvec != nullptr ? p.addvec[(size_t)tl.b * p.addvec_bs + ch_of_col] : 0.f;
        if constexpr (RES) {
            const char* rbase = reinterpret_cast<const char*>(p.residual + (((size_t)tl.b * p.H + tl.oy0 + wm * 2 * MT) * p.W + tl.ox0) * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    R.rv[mt][g] = nt_load4(rbase + (size_t)((mt * 2 + (g >> 1)) * p.W) * 128 + (g & 1) * 1024 + e_lane);      // read once
        }
    };

    auto mma_chunk = [&](const PPTile& tl, auto C_) __attribute__((always_inline)) {
        constexpr int C = decltype(C_)::value;
        if constexpr (C == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { for (int a = 0; a < NACC; ++a) acc[a][mt][r] = 0.f; }
        } else {
            if (p.scale != nullptr && p.ch[C - 1].seg != p.ch[C].seg) {
                // the accumulator changes units: from the previous K-segment's operand scale to this one's (powers of two: exact)
                const float ratio = scale_c[8 * tl.b + p.ch[C].seg] * scale_c[8 * tl.b + 4 + p.ch[C - 1].seg];
                if (ratio != 1.0f) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { for (int a = 0; a < NACC; ++a) acc[a][mt][r] *= ratio; }
                }
            }
        }
        const unsigned wbase = (unsigned)wlds(C) + b_lane;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (C < N9) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_step(wbase, tap * 2 + j, tap / 3, tap % 3, j);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_step(wbase, j, 1, 1, j);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    int stamp_n = 0; (void)stamp_n;
#ifdef PP_PROBE_BUILD
    unsigned long long* const stamp_buf = g_pp_dbg;      // null: no stamps (the probe's plain timing runs)
    stamp_n = -g_pp_dbg_skip;
    // shader-clock ticks of the walk of every 32nd workgroup (slots behind the step stamps): ticks / launch time = the clock the launch ran at
    if (stamp_buf != nullptr && (blockIdx.x & 31) == 0 && tid == 0) stamp_buf[2 * 64 * 8 + 2 * (blockIdx.x >> 5)] = __builtin_readcyclecounter();
#endif
    // ---- one step = VALU phase + MFMA phase of chunk C of the team's it-th tile (PAR = it & 1) -------------------------------------------
    // register set of step (it, C): the step's global index it * NCH + C, modulo 2
    auto step = [&](int it, auto PAR_, auto C_, bool last_step) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, C = decltype(C_)::value;
        constexpr int SI = (NCH % 2 == 0) ? (C & 1) : ((PAR + C) & 1);
        Pre& S = SI == 0 ? pre0 : pre1;
        const PPTile tl = tile_of(it);
        // ---- VALU phase ---------------------------------------------------------------------------------------------------------------
        PP_STAMP(0);
        PP_STAMP(1);
        if constexpr (C == 0) epilogue((NRES == 2 && (PAR ^ 1) == 1) ? res1 : res0, tile_of(it - 1), live_of(it - 1) && it > 0);      // the previous tile (parity PAR ^ 1)
        PP_STAMP(2);
        // staging + requests, two steps ahead: the patch of step (it, C) + 2 into the set being staged ...
        constexpr int C2 = (C + 2) % NCH, DT = (C + 2) / NCH;
        stage_and_request(S, C_, tile_of(it + DT), ic<C2>{});
        PP_STAMP(3);
        // ... and, when that step opens a tile, the residual of the tile that closes in front of it (tile it + DT - 1, parity PAR + DT - 1)
        constexpr int CR = (C + RD) % NCH, DTR = (C + RD) / NCH;
        if constexpr (CR == 0) issue_res((NRES == 2 && ((PAR + DTR - 1) & 1) == 1) ? res1 : res0, tile_of(it + DTR - 1));
        PP_STAMP(4);
        __syncthreads();
        // ---- MFMA phase (LDS + matrix pipe only) ----------------------------------------------------------------------------------------
        PP_STAMP(5);
        mma_chunk(tl, C_);
        PP_STAMP(6);
        if (!(TEAMS == 2 && team == 1 && last_step)) __syncthreads();
        PP_STAMP(7);
        ++stamp_n;
    };
    auto tile_steps = [&](int it, auto PAR_) __attribute__((always_inline)) {
        const bool last_tile = it == niter - 1;
        if constexpr (NCH >= 1) step(it, PAR_, ic<0>{}, last_tile && NCH == 1);
        if constexpr (NCH >= 2) step(it, PAR_, ic<1>{}, last_tile && NCH == 2);
        if constexpr (NCH >= 3) step(it, PAR_, ic<2>{}, last_tile && NCH == 3);
        if constexpr (NCH >= 4) step(it, PAR_, ic<3>{}, last_tile && NCH == 4);
        if constexpr (NCH >= 5) step(it, PAR_, ic<4>{}, last_tile && NCH == 5);
        if constexpr (NCH >= 6) step(it, PAR_, ic<5>{}, last_tile && NCH == 6);
    };

    // ---- the walk --------------------------------------------------------------------------------------------------------------------
    if (ntl <= 0) return;             // (never with the launcher's grid: T >= 8 G)
    issue_patch(pre0, tile_of(0), ic<0>{});                                        // step 0
    if constexpr (NCH == 1) { issue_patch(pre1, tile_of(1), ic<0>{}); if constexpr (RD == 2) issue_res(res0, tile_of(0)); }   // step 1 = tile 1; tile 0 closes at step 1
    else issue_patch(pre1, tile_of(0), ic<1>{});
    __syncthreads();                  // weights visible
    if (TEAMS == 2 && team == 1) __syncthreads();
    // (both halves are unconditional inside the loop: with `if (it + 1 < niter)` around the odd half hipcc's waitcnt pass also has to cover
    // the path that skips it, on which the even half's registers were requested just 4 vector-memory operations ago - it then waits vmcnt(4)
    // in every even step, i.e. for the patch of the step before: one step of prefetch distance again)
    int it = 0;
    for (; it + 1 < niter; it += 2) {
        tile_steps(it, ic<0>{});
        tile_steps(it + 1, ic<1>{});
    }
    if (it < niter) tile_steps(it, ic<0>{});
    if (NRES == 2 && ((niter - 1) & 1)) epilogue(res1, tile_of(niter - 1), live_of(niter - 1));
    else epilogue(res0, tile_of(niter - 1), live_of(niter - 1));
    flush_stats();
#ifdef PP_PROBE_BUILD
    if (stamp_buf != nullptr && (blockIdx.x & 31) == 0 && tid == 0) stamp_buf[2 * 64 * 8 + 2 * (blockIdx.x >> 5) + 1] = __builtin_readcyclecounter();
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------

static int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

constexpr int PP_MT = 1;      // 8 x 16-pixel team tiles: at 2 waves per SIMD two register sets of raw patch + the residual + accumulators fit 256 registers.
                              // (16 x 16 tiles - MT = 2, 11 float4 per prefetch set - spill 57-156 registers with the two-step prefetch and measured
                              // +30 ... +70 % per launch, r4; they would need conv_pp64's one-step re-request scheme)

// chunk structures instantiated: ResidualBlock conv1 / conv2 of the down path (one 3x3 chunk, with / without the identity residual), conv1 of
// the up path over cat[h, skip] (two / three 3x3 chunks), conv2 of the up path with the folded 1x1 shortcut (one 3x3 + two / three 1x1 chunks)
static bool pp_structure(int n9, int n1, bool res) {
    if (res) return n9 == 1 && n1 == 0;
    return (n1 == 0 && n9 >= 1 && n9 <= 3) || (n9 == 1 && (n1 == 2 || n1 == 3));
}

bool conv_pp_supported(const ConvParams& p, int stride, int up, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_PP") ? atoi(getenv("PNPFLOW_HIP_PP")) : 1;      // test-only A/B switch: 0 off, 1 on, 3 on in the default mode only (mode 2 on conv_mfma16)
    constexpr int TH = 8 * PP_MT;
    if (mode == 0 || (terms != 3 && !(terms == 1 && mode != 3)) || stride != 1 || up != 0 || p.gnb_x != nullptr) return false;
    if (p.Cout != 32 || p.out_cstride != 32 || (p.residual != nullptr && p.res_cstride != 32)) return false;
    if (p.H % TH || p.W % 16 || p.Hs != p.H || p.Ws != p.W) return false;
    if (ilog2_exact(p.H / TH) < 0 || ilog2_exact(p.W / 16) < 0) return false;
    // a persistent grid pays its prologue (weights -> LDS per workgroup, two-step pipeline fill) over >= 32 tiles per team: measured at 256^2,
    // B = 8 / 16 / 32 / 80 -> +11 % / +1.5 % / -1.4 % / -3 % on the whole forward; smaller launches stay on conv_mfma16
    // (teams of the launch = 2 per CU; 32 x 512 tiles on the 256 CUs of an MI355X)
    const int grid = persistent_grid();
    if (grid < 8 || (long)p.B * (p.H / TH) * (p.W / 16) < 32L * 2 * grid) return false;
    if (p.W > 2048) return false;      // patch pixel offsets py * W + px are packed into 16 bits (pk9[])
    int n9 = 0, n1 = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if (s.w_mode != 0 || s.w16 == nullptr || s.C % 32) return false;
        if (s.taps == 9) { if (n1 > 0 || s.xform == 0) return false; n9 += s.C / 32; }     // 9-tap chunks are GroupNorm-ed (their coefficient loads are unconditional)
        else if (s.taps == 1) { if (s.xform != 0) return false; n1 += s.C / 32; }
        else return false;
    }
    if (!pp_structure(n9, n1, p.residual != nullptr)) return false;
    if (p.gn_C > 0 && p.coef == nullptr) return false;
    return (size_t)n9 * pp_w9(terms) + (size_t)n1 * pp_w1(terms) + 2 * pp_patch_bytes(PP_MT) <= 160 * 1024;
}

// TEAMS = 1: two independent 4-wave workgroups per CU (each with its own copy of the weights) where LDS allows - the teams of one
// 8-wave workgroup wait for each other at every phase boundary, independent workgroups only for their own data (r4: 192 vs 214 us on the
// one-chunk conv); TEAMS = 2 (one workgroup per CU, weights shared) for the structures whose weights do not fit twice
template <int N9, int N1, bool RES, int TEAMS, int TERMS>
static hipError_t launch_pp_tt(const PPParams& p0, hipStream_t s) {
    static unsigned long long attr_set = 0ull;
    auto kern = conv_pp_kernel<PP_MT, N9, N1, RES, TEAMS, TERMS>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    const int grid = persistent_grid() * (TEAMS == 1 ? 2 : 1);
    if (grid <= 0) return hipErrorInvalidConfiguration;
    PPParams p = p0;
    p.lx = ilog2_exact(p.W / 16); p.ly = ilog2_exact(p.H / (8 * PP_MT));
    p.rot = 5;
    const size_t lds = (size_t)N9 * pp_w9(TERMS) + (size_t)N1 * pp_w1(TERMS) + TEAMS * pp_patch_bytes(PP_MT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * TEAMS), lds, s, p);
    return hipGetLastError();
}

template <int N9, int N1, bool RES, int TERMS>
static hipError_t launch_pp_t(const PPParams& p, hipStream_t s) {
    constexpr size_t one = (size_t)N9 * pp_w9(TERMS) + (size_t)N1 * pp_w1(TERMS) + pp_patch_bytes(PP_MT);
    if constexpr (2 * one <= 160 * 1024) return launch_pp_tt<N9, N1, RES, 1, TERMS>(p, s);
    else return launch_pp_tt<N9, N1, RES, 2, TERMS>(p, s);
}

template <int TERMS>
static hipError_t launch_conv_pp_terms(const PPParams& p, hipStream_t s) {
    const bool res = p.residual != nullptr;
    if (p.n9 == 1 && p.n1 == 0) return res ? launch_pp_t<1, 0, true, TERMS>(p, s) : launch_pp_t<1, 0, false, TERMS>(p, s);
    if (res) return hipErrorInvalidValue;
    if (p.n9 == 2 && p.n1 == 0) return launch_pp_t<2, 0, false, TERMS>(p, s);
    if (p.n9 == 3 && p.n1 == 0) return launch_pp_t<3, 0, false, TERMS>(p, s);
    if (p.n9 == 1 && p.n1 == 2) return launch_pp_t<1, 2, false, TERMS>(p, s);
    if (p.n9 == 1 && p.n1 == 3) return launch_pp_t<1, 3, false, TERMS>(p, s);
    return hipErrorInvalidValue;
}

hipError_t launch_conv_pp(const PPParams& p, hipStream_t s, int terms) {
    if (terms == 1) return launch_conv_pp_terms<1>(p, s);
    if (terms == 3) return launch_conv_pp_terms<3>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace pf
