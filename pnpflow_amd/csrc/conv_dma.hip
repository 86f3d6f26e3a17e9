// Split-fp16 implicit-GEMM convolution whose A operand arrives by LDS-DMA (round 3).
//
// conv_mfma16.hip stages its activation patch through registers: global fp32 loads -> GroupNorm + SiLU + power-of-two scale + hi/lo split
// on the VALU -> ds_write, between two barriers per K-chunk, once per N-block and per halo overlap (1.4-2.8x redundant on the 128 / 256
// channel levels).  Here that transform is done ONCE per consumed tensor by `prep_split_kernel`, which writes the MFMA-ready operand
//     a16[b][chunk][Hs+2][Ws+2][128 B]          one 128-byte RECORD per (pixel, K-chunk), zero border rows / columns
//         TERMS = 3:  32 channels per chunk, record = [32 hi halfs | 32 lo halfs]        (fp32-equivalent mode)
//         TERMS = 1:  64 channels per chunk, record = [64 hi halfs]                      (precision mode 2: no low halves anywhere)
// and `conv_dma_kernel` fetches the halo patch of a chunk straight into LDS with `global_load_lds_dwordx4` (no registers, no VALU, no
// bounds checks: the zero border is part of the tensor), double-buffered, ONE barrier per chunk.  The 16-byte pieces of a record are
// XOR-swizzled by the patch column on the SOURCE side (LDS-DMA writes lane-linear), so that the A-fragment ds_read_b128 of the 16
// pixels of a lane group hit 16 different 16-byte bank slots for every tap shift.
//
// Weight fragments still go straight from L2 into a register ring two k16-steps ahead (one wave = 32 output channels: nothing to share
// through LDS), but as inline-asm loads with hand-counted `s_waitcnt vmcnt(N)`: next to an LDS-DMA in flight hipcc waits `vmcnt(0)` at
// the first use of any ordinary load (cdna_hip_programming.md, "Three .s-level traps" (b)), which would drain the next chunk's patch
// at every k-step.  The ring runs across chunk boundaries (the first two fragments of chunk c+1 are requested during the last two
// steps of chunk c).  Memory operations return in order, so the queue at every wait is known statically:
//     chunk entry            [.. DMA(c) ..][B(c,0)][B(c,1)]                       wait vmcnt(2 NB)          -> DMA(c) has landed
//     step 0 / 1             [B(c,s)][..][DMA(c+1) x IPW][B(c,s+2)]               wait vmcnt(2 NB + IPW)
//     step s >= 2            [B(c,s)][B(c,s+1)][B(c,s+2)]                         wait vmcnt(2 NB)
// (NB = loads per fragment step, IPW = DMA instructions per wave and chunk; the last chunk issues no DMA and waits vmcnt(2 NB)).
//
// Epilogue (bias / time embedding / residual / statistics / fused GroupNorm-backward first stage) is conv_mfma16.hip's.
//
// UP = 2 (round 6): the nearest-x2 upsampling conv (models.py:70-91: F.interpolate(scale 2, nearest) then a 3x3 conv) in its PHASE form.  Output
// pixel (2i + dy, 2j + dx) reads upsampled rows 2i + dy - 1 .. 2i + dy + 1 = source rows {i-1, i, i} (dy = 0) / {i, i, i+1} (dy = 1), so each of the
// four output phases is a 2 x 2 conv of the SOURCE image whose taps are sums of the 3 x 3 weights that land on the same source pixel
// (engine.hip: packed phase image, N = 4 Cout rows, 4 taps): 16 instead of 36 multiply-adds per source pixel and (c_in, c_out) pair.  The grid tiles
// the source image; a wave's 32 weight rows belong to one phase (dy, dx) = ((n / Cout) >> 1, (n / Cout) & 1), its taps (ty, tx) read patch pixel
// (row + ty + dy, col + tx + dx) and it writes pixel (2 row + dy, 2 col + dx).  The image border needs no case: the taps that fall outside the
// upsampled image are exactly the ones whose source pixel is outside the source image (zero border records).
// Reference: the convolutions of pnpflow/models.py:94-113 (ResidualBlock), :145-162 (SelfAttention 1x1), :70-91 (Upsample).
#include <cstdlib>
#include <type_traits>
#include "pf_common.h"


namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define PF_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PF_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float silu_fast_d(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------------------------
// prep: GroupNorm-apply + SiLU + operand scale + fp16 hi/lo split, once per (tensor, consuming launch)
// ---------------------------------------------------------------------------------------------------------------------------------
template <int TERMS>
__global__ __launch_bounds__(256) void prep_split_kernel(const PrepParams p) {
    const int si = blockIdx.z, b = blockIdx.y;
    const int C = p.C[si], G8 = C >> 3, Hp = p.Hs + 2, Wp = p.Ws + 2;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= Hp * Wp * G8) return;
    const int pix = item / G8, g8 = item - pix * G8;
    const int py = pix / Wp, px = pix - py * Wp;
    constexpr int GPC = TERMS == 3 ? 4 : 8;                 // 8-channel groups per chunk record
    const int NCH = G8 / GPC, ch = g8 / GPC, sub = g8 - ch * GPC;
    char* rec = reinterpret_cast<char*>(p.dst[si]) + ((size_t)((size_t)b * NCH + ch) * (Hp * Wp) + pix) * 128;
    const bool interior = py >= 1 && py <= p.Hs && px >= 1 && px <= p.Ws;
    uint4 oh = make_uint4(0u, 0u, 0u, 0u), ol = oh;
    if (interior) {
        const float* s = p.src[si] + ((size_t)((size_t)b * p.Hs + (py - 1)) * p.Ws + (px - 1)) * p.cstride[si] + p.coff[si] + g8 * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(s), a1 = *reinterpret_cast<const float4*>(s + 4);
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const int xf = p.xform[si];
        if (xf != 0) {
            const float* cb = p.coef + (size_t)b * 2 * p.coef_stride + p.gn_off[si] + g8 * 8;
            const float4 c0 = *reinterpret_cast<const float4*>(cb), c1 = *reinterpret_cast<const float4*>(cb + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(cb + p.coef_stride), h1 = *reinterpret_cast<const float4*>(cb + p.coef_stride + 4);
            const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = v[i] * sc[i] + sh[i]; if (xf == 2) v[i] = silu_fast_d(v[i]); }
        }
        const float a_scale = p.scale != nullptr ? p.scale[8 * b + si] : 1.0f;
        f16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float w = v[i] * a_scale;
            // opaque: the hi that is stored and the hi that is subtracted must be the same rounding of the same fp32 value (see the
            // note in conv_mfma16.hip's staging; -ffp-contract=fast may otherwise fuse the multiply into the subtraction)
            asm volatile("" : "+v"(w));
            h[i] = (_Float16)w;
            l[i] = (_Float16)(w - (float)h[i]);
        }
        oh = *reinterpret_cast<const uint4*>(&h); ol = *reinterpret_cast<const uint4*>(&l);
    }
    *reinterpret_cast<uint4*>(rec + sub * 16) = oh;
    if constexpr (TERMS == 3) *reinterpret_cast<uint4*>(rec + 64 + sub * 16) = ol;
}

hipError_t launch_prep_split(const PrepParams& p, hipStream_t s) {
    if (p.nseg < 1 || p.nseg > 3) return hipErrorInvalidValue;
    int maxG8 = 0;
    for (int i = 0; i < p.nseg; ++i) {
        if (p.C[i] % (p.terms == 3 ? 32 : 64) != 0 || p.dst[i] == nullptr || p.src[i] == nullptr) return hipErrorInvalidValue;
        if (p.xform[i] != 0 && p.coef == nullptr) return hipErrorInvalidValue;
        maxG8 = p.C[i] / 8 > maxG8 ? p.C[i] / 8 : maxG8;
    }
    const long items = (long)(p.Hs + 2) * (p.Ws + 2) * maxG8;
    dim3 grid((unsigned)((items + 255) / 256), (unsigned)p.B, (unsigned)p.nseg);
    if (p.terms == 3) hipLaunchKernelGGL(prep_split_kernel<3>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(prep_split_kernel<1>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv
// ---------------------------------------------------------------------------------------------------------------------------------
#define PF_WAITV(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")

// one fragment step of the weight ring: hi (and lo) quads of this lane's output channel, requested by inline asm (hidden from hipcc's
// waitcnt bookkeeping on purpose - see the header).  `s_nop 4`: the scalar base may come fresh out of a v_readfirstlane
// (VALU-written SGPR read by a VMEM instruction: 5 wait states, which hipcc does not pad inside an asm string).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));       // (a native vector: HIP's uint4 is a struct, i.e. a memory operand to an asm)
struct BFrag { u32x4 h, l; };

template <int TERMS>
__device__ __forceinline__ void bload(BFrag& f, unsigned voff, unsigned voff_lo, const char* sbase) {
    u32x4 h, l;      // (plain locals: an asm operand that is a member reached through a reference is an indirect operand)
    if constexpr (TERMS == 3) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4" : "=&v"(h), "=&v"(l) : "v"(voff), "v"(voff_lo), "s"(sbase) : "memory");
        f.h = h; f.l = l;
    } else {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(h) : "v"(voff), "s"(sbase) : "memory");
        f.h = h;
    }
}
// wait until at most N younger vector-memory operations are outstanding; naming the fragment "+v" pins every consumer below the wait
template <int TERMS, int N>
__device__ __forceinline__ void bwait(BFrag& f) {
    u32x4 h = f.h, l = f.l;
    if constexpr (TERMS == 3) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(h), "+v"(l) : "n"(N) : "memory"); f.h = h; f.l = l; }
    else { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(h) : "n"(N) : "memory"); f.h = h; }
}

// one LDS-DMA instruction: every lane fetches the 16 bytes at sbase + voff into LDS byte address lds_dst + 16 * lane (lds_dst is
// wave-uniform; M0 carries it and is restored: the compiler does not preserve M0 around an asm statement nor expect it changed)
__device__ __forceinline__ void glds16(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MT, int NT, int WM, int WN, int UP, int TERMS, bool GNB>
__global__ __launch_bounds__(256, 3) void conv_dma_kernel(const ConvParams p) {
    static_assert(NT == 1, "one 32-channel N-tile per wave (the weight ring is sized for it)");
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int KC = TERMS == 3 ? 32 : 64;     // channels per chunk = one 128-byte record per pixel
    constexpr int KS = KC / 16;                  // k16-steps per tap and chunk
    constexpr int TH = 2 * MT * WM, TW = 16, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int NPIECE = PP * 8, NINST = (NPIECE + 63) / 64, IPW = (NINST + 3) / 4;   // 16-byte pieces / DMA wave-instructions per chunk / per wave
    constexpr int BUF = IPW * 4 * 1024;          // bytes of one patch buffer
    constexpr int BN = WN * NT * 32;
    constexpr int NB = TERMS == 3 ? 2 : 1;       // vector-memory operations per fragment step
    constexpr int EPI = (4 * 32 * 36 + WM * BN * 2) * 4;
    static_assert(EPI <= 2 * BUF, "epilogue scratch overlays the patch buffers");
    static_assert((MT * 2 + 2) * PW * 128 < 65536, "ds_read immediate offsets");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const s_buf = smem_raw;

    const int WC = UP == 2 ? 4 * p.Cout : p.Cout;       // rows of the weight image (UP = 2: four phases x Cout)
    const int tiles_x = (UP == 2 ? p.Ws : p.W) / TW, tiles_y = (UP == 2 ? p.Hs : p.H) / TH;
    int bid, nb;
    if (p.xcd_map) {
        const int NBk = WC / BN;
        const int total = gridDim.x;
        const int work = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
        bid = work / NBk; nb = work % NBk;
    } else {
        bid = blockIdx.x; nb = blockIdx.y;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nb * BN;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int Wp = p.Ws + 2;
    const size_t plane = (size_t)(p.Hs + 2) * Wp * 128;                  // bytes of one (image, chunk) plane of an a16 tensor

    float seg_scale[3], seg_inv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { seg_scale[i] = p.scale != nullptr ? p.scale[8 * b + i] : 1.0f; seg_inv[i] = p.scale != nullptr ? p.scale[8 * b + 4 + i] : 1.0f; }

    // source byte offset (inside an image-chunk plane) of the 16-byte piece this lane brings in DMA instruction i.  LDS slot
    // s = pixel * 8 + q holds piece  q ^ ((px >> 1) & 7)  of that pixel's record.
    unsigned a_off[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int slot = min((i * 4 + wave) * 64 + lane, NPIECE - 1);
        const int pp = slot >> 3, q = slot & 7;
        const int py = pp / PW, px = pp - py * PW;
        const int g = q ^ ((px >> 1) & 7);
        int sy, sx;
        if (UP == 1) { sy = (oy0 - 1 + py + 2) >> 1; sx = (ox0 - 1 + px + 2) >> 1; }      // nearest-x2 upsampled view of the padded source
        else { sy = oy0 + py; sx = ox0 + px; }
        a_off[i] = (unsigned)((sy * Wp + sx) * 128 + g * 16);
    }
    const unsigned lds0 = (unsigned)(size_t)PF_LPTR(s_buf) + (unsigned)wave * 1024u;      // LDS byte address of this wave's first DMA slot
    auto dma = [&](int dsi, int dch, int buf) __attribute__((always_inline)) {
        const ConvSeg& sg = p.seg[dsi];
        const char* base = reinterpret_cast<const char*>(sg.a16) + ((size_t)b * (sg.C / KC) + dch) * plane;
#pragma unroll
        for (int i = 0; i < IPW; ++i) glds16(a_off[i], base, lds0 + (unsigned)(buf * BUF + i * 4096));
    };

    // weight fragments: per-lane byte offset of this lane's output channel inside a [slice][tap] block, scalar base per step
    const int nclamp = min(n0 + wn * 32 + l31, WC - 1);
    // UP = 2: phase and first output channel of this wave's 32 weight rows (Cout is a multiple of 32: a wave never straddles two phases)
    const int phase = UP == 2 ? (n0 + wn * 32) / p.Cout : 0, ph_dy = phase >> 1, ph_dx = phase & 1;
    const int nc0 = UP == 2 ? (n0 + wn * 32) - phase * p.Cout : n0 + wn * 32;
    // block (k16-slice, tap) = [hi halves: Cout x 32 B][lo halves: Cout x 32 B] (TERMS = 1: the hi part only): a wave's fragment load is one
    // contiguous 1 KiB run
    const unsigned b_voff = (unsigned)(nclamp * 32 + hi * 16);
    const unsigned b_voff_lo = b_voff + (unsigned)WC * 32u;
    const size_t wblk = (size_t)WC * (TERMS == 3 ? 64 : 32);           // bytes of one [slice][tap] block
    // first block of chunk (si, ch): fragment step (tap, j) of the chunk is block j * taps + tap behind it
    auto wchunk = [&](int wsi, int wch) -> const char* {
        const ConvSeg& sg = p.seg[wsi];
        return reinterpret_cast<const char*>(TERMS == 3 ? sg.w16 : sg.w16h) + (size_t)wch * KS * sg.taps * wblk;
    };

    // A fragments: lane (l31, hi) reads the record of patch pixel (row0 + prow + ky, pcol + kx), piece (part * 4 | j * 2 | hi) ^ swizzle
    const int prow = l31 >> 4, pcol = l31 & 15;
    unsigned lb[3];                              // (pixel byte address | swizzled hi bit) per kx
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        // (UP = 2: entry tx of the 2 x 2 window = patch column pcol + tx + dx, patch row offset dy folded into the pixel address)
        const int col = pcol + kx + (UP == 2 ? ph_dx : 0), row = wm * MT * 2 + prow + (UP == 2 ? ph_dy : 0);
        lb[kx] = (unsigned)((row * PW + col) * 128) | (unsigned)((((col >> 1) & 7) ^ hi) << 4);
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // one k16-step: A fragments of every M-tile from the patch buffer, then the MFMAs of this wave's 32 output channels
    // (lbc = lb + byte offset of the current buffer: the XOR below touches bits 4-6 only, the buffer offset bits >= 10)
    auto load_a = [&](const unsigned (&lbc)[3], int ky, int kx, int j, f16x8 (&ah)[MT], f16x8 (&al)[MT]) __attribute__((always_inline)) {
        const unsigned base_h = lbc[kx] ^ (unsigned)((j * 2) << 4);
        const unsigned base_l = lbc[kx] ^ (unsigned)((4 + j * 2) << 4);
        // (in the order the MFMAs consume them: the low halves first)
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) al[mt] = *reinterpret_cast<const f16x8*>(s_buf + base_l + (mt * 2 + ky) * PW * 128);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const f16x8*>(s_buf + base_h + (mt * 2 + ky) * PW * 128);
    };
    auto mma = [&](const f16x8 (&ah)[MT], const f16x8 (&al)[MT], const BFrag& f) __attribute__((always_inline)) {
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], __builtin_bit_cast(f16x8, f.h), acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], __builtin_bit_cast(f16x8, f.l), acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], __builtin_bit_cast(f16x8, f.h), acc[mt], 0, 0, 0);
    };

    // ---- main loop -----------------------------------------------------------------------------------------------------------------
    // A requested weight fragment's loads are invisible to the compiler, so between its request and its counted wait the registers
    // must not be copied, spilled or renamed (tools/isa_audit_asm_loads.py replays the compiled ISA and checks exactly that).
    // Per chunk:   [request B(0), B(1)]  ->  wait for this chunk's patch  ->  barrier  ->  request the next chunk's patch  ->  steps
    // (step s requests B(s + 2)).  Queue (oldest first) and the wait of every step:
    //     [B0][B1][DMA x IPW] + [B2] at step 0:  B0 needs vmcnt(2 NB + IPW);  step 1 the same;  steps >= 2: vmcnt(2 NB) (B(s+1), B(s+2));
    //     without the carry the last two steps request nothing: vmcnt(NB), vmcnt(0).
    // CARRY (9-tap loop only): the last two steps request steps 0 / 1 of the NEXT chunk instead, into the registers the next
    // iteration expects them in (18 / 36 steps are a multiple of the ring length 3), so a chunk does not start with a cold L2 round trip.
    // The last chunk requests its own patch again into the free buffer (never read), which keeps the counts static.
    BFrag f0, f1, f2;
    f0.h = u32x4{0u, 0u, 0u, 0u}; f0.l = f0.h; f1 = f0; f2 = f0;
    auto chunk_body = [&](auto taps_c, auto carry_c, const char* wcur, const char* wnext, int ntaps, int cur, int nsi, int nch) __attribute__((always_inline)) {
        constexpr int TAPS = decltype(taps_c)::value;
        constexpr bool CARRY = decltype(carry_c)::value;
        constexpr int NS = TAPS * KS;
        static_assert(NS >= 2 && NS <= 36 && (!CARRY || NS % 3 == 0), "unrolled steps / ring phase");
        static_assert(TAPS == 9 || TAPS == 1 || (TAPS == 4 && UP == 2), "3x3, 1x1, or the 2x2 phase window of the upsampling conv");
        // weight block of fragment step s2 of this chunk ((tap, slice) = (s2 / KS, s2 % KS)), or of step s2 - NS in {0, 1} of the next
        auto wstep = [&](int s2) -> const char* {
            if (s2 < NS) return wcur + (size_t)((s2 % KS) * TAPS + s2 / KS) * wblk;
            return wnext + (size_t)((s2 - NS) * ntaps) * wblk;
        };
        if constexpr (!CARRY) {
            bload<TERMS>(f0, b_voff, b_voff_lo, wstep(0));
            bload<TERMS>(f1, b_voff, b_voff_lo, wstep(1));
        }
        PF_WAITV(2 * NB);                                 // in-order return: everything older than B0 / B1 - this chunk's patch - has landed
        __builtin_amdgcn_s_barrier();                     // raw barrier (no vmcnt(0) drain); every wave has finished reading the other buffer
        dma(nsi, nch, cur ^ 1);
        unsigned lbc[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) lbc[kx] = lb[kx] + (unsigned)(cur * BUF);
#define PF_KY(S) (TAPS == 9 ? ((S) / KS) / 3 : TAPS == 4 ? ((S) / KS) >> 1 : 1)
#define PF_KX(S) (TAPS == 9 ? ((S) / KS) % 3 : TAPS == 4 ? ((S) / KS) & 1 : 1)
#define PF_STEP(S, FC, FN)                                                                                                     \
        if constexpr ((S) < NS) {                                                                                              \
            constexpr bool REQ_ = CARRY || (S) + 2 < NS;                                                                       \
            if constexpr (REQ_) bload<TERMS>(FN, b_voff, b_voff_lo, wstep((S) + 2));                                                      \
            constexpr int W_ = REQ_ ? ((S) < 2 ? 2 * NB + IPW : 2 * NB) : ((S) + 1 < NS ? ((S) < 2 ? NB + IPW : NB) : ((S) < 2 ? IPW : 0)); \
            bwait<TERMS, W_>(FC);                                                                                              \
            f16x8 ah_[MT], al_[MT];                                                                                            \
            load_a(lbc, PF_KY(S), PF_KX(S), (S) % KS, ah_, al_);                                                               \
            mma(ah_, al_, FC);                                                                                                 \
        }
#define PF_STEP3(S) PF_STEP((S), f0, f2) PF_STEP((S) + 1, f1, f0) PF_STEP((S) + 2, f2, f1)
        PF_STEP3(0) PF_STEP3(3) PF_STEP3(6) PF_STEP3(9) PF_STEP3(12) PF_STEP3(15) PF_STEP3(18) PF_STEP3(21) PF_STEP3(24) PF_STEP3(27) PF_STEP3(30) PF_STEP3(33)
#undef PF_STEP3
#undef PF_STEP
#undef PF_KY
#undef PF_KX
    };
    auto rescale = [&](int rsi) __attribute__((always_inline)) {
        // the accumulator changes units: from segment rsi-1's operand scale to segment rsi's (both powers of two: exact)
        const float ratio = (rsi == 1 ? seg_scale[1] * seg_inv[0] : seg_scale[2] * seg_inv[1]);
        if (ratio != 1.0f) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] *= ratio;
        }
    };

    dma(0, 0, 0);
    int si = 0, ch = 0, cur = 0;
    bool more = true;
    constexpr bool CARRY9 = false;      // (the weight ring carried across chunk boundaries measured <= 0.5 %: profiles/r03_ab_dma_variants.md)
    if (CARRY9 && p.seg[0].taps == 9) {
        const char* w0 = wchunk(0, 0);
        bload<TERMS>(f0, b_voff, b_voff_lo, w0);
        bload<TERMS>(f1, b_voff, b_voff_lo, w0 + (size_t)9 * wblk);      // step 1 = (tap 0, slice 1)
    }
    if constexpr (UP == 2) {      // one raw segment of 2 x 2 phase windows
        while (more) {
            int nsi = si, nch = ch + 1;
            if (nch * KC >= p.seg[si].C) { nsi = si + 1; nch = 0; }
            more = nsi < p.nseg;
            chunk_body(std::integral_constant<int, 4>{}, std::false_type{}, wchunk(si, ch), nullptr, 4, cur, more ? nsi : si, more ? nch : ch);
            si = nsi; ch = nch; cur ^= 1;
        }
    }
    // 9-tap segments first (conv_dma_supported orders them so), then the 1-tap ones: two loops, one body each
    while (more && p.seg[si].taps == 9) {
        if (ch == 0 && si > 0) rescale(si);
        int nsi = si, nch = ch + 1;
        if (nch * KC >= p.seg[si].C) { nsi = si + 1; nch = 0; }
        more = nsi < p.nseg;
        const int dsi = more ? nsi : si, dch = more ? nch : ch;
        // the carried requests at the end of this chunk are for the next 9-tap chunk; past the last one they re-read this chunk's
        // first two blocks (never used: the 1-tap loop below starts its own ring)
        const bool next9 = more && p.seg[nsi].taps == 9;
        chunk_body(std::integral_constant<int, 9>{}, std::integral_constant<bool, CARRY9>{}, wchunk(si, ch), next9 ? wchunk(nsi, nch) : wchunk(si, ch), 9, cur, dsi, dch);
        si = nsi; ch = nch; cur ^= 1;
    }
    // the carried overrun requests of the last 9-tap chunk are still in flight in (f0, f1): nothing below may reuse those registers
    // before they have landed (the compiler considers them dead - the ISA audit caught it handing one to the rescale factor)
    if constexpr (CARRY9) PF_WAITV(0);
    while (more) {
        if (ch == 0 && si > 0) rescale(si);
        int nsi = si, nch = ch + 1;
        if (nch * KC >= p.seg[si].C) { nsi = si + 1; nch = 0; }
        more = nsi < p.nseg;
        chunk_body(std::integral_constant<int, 1>{}, std::false_type{}, wchunk(si, ch), nullptr, 1, cur, more ? nsi : si, more ? nch : ch);
        si = nsi; ch = nch; cur ^= 1;
    }
    PF_WAITV(0);                                          // the last chunk's overrun requests must land before the scratch below reuses LDS

    // ---- epilogue (conv_mfma16.hip's) ----------------------------------------------------------------------------------------------
    __syncthreads();                                   // every wave is done reading the patch
    constexpr int TP = 36;                             // scratch row pitch in floats (32 + 4 pad)
    float* s_tr = reinterpret_cast<float*>(s_buf) + wave * (32 * TP);
    float* s_red = reinterpret_cast<float*>(s_buf) + 4 * 32 * TP;   // [WM][BN][2] behind the 4 scratch tiles
    const float oscale = p.out_scale * (1.0f / 256.0f) * (p.nseg == 1 ? seg_inv[0] : (p.nseg == 2 ? seg_inv[1] : seg_inv[2]));
    const int cq = lane & 7;                           // this lane's channel quad inside a 32-channel tile
    {
        const int ncol = nc0;                          // first channel of this wave's N-tile
        const int n = ncol + l31;
        const float add = (p.addvec != nullptr && n < p.Cout) ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f;
        const int n4 = ncol + cq * 4;
        const bool nok4 = n4 < p.Cout;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        float4 g_mu, g_rs, g_ga, g_be;
        if constexpr (GNB) {
            const int gc = p.gnb_coff + min(n4, p.Cout - 4);
            g_mu = *reinterpret_cast<const float4*>(p.gnb_mu + (size_t)b * p.gnb_Ct + gc);
            g_rs = *reinterpret_cast<const float4*>(p.gnb_rs + (size_t)b * p.gnb_Ct + gc);
            g_ga = *reinterpret_cast<const float4*>(p.gnb_gamma + gc);
            g_be = *reinterpret_cast<const float4*>(p.gnb_beta + gc);
        }
        constexpr int RG = (MT >= 2 && !GNB) ? 2 : 1;      // (GNB: a second set of residual + GroupNorm-input registers spills at the 168-register bound)
        float4 rv[RG][4];
        float4 xv[GNB ? RG : 1][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (GNB) {
                if (mt % RG == 0) {
#pragma unroll
                    for (int g = 0; g < RG; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int px = (lane >> 3) + 8 * i;
                            const int oy = min(oy0 + (wm * MT + mt + g) * 2 + (px >> 4), p.H - 1), ox = min(ox0 + (px & 15), p.W - 1);
                            const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                            xv[g][i] = *reinterpret_cast<const float4*>(p.gnb_x + pix * p.gnb_xstride + min(n4, p.Cout - 4));
                        }
                }
            }
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
                s_tr[row * TP + l31] = acc[mt][r] * oscale + add;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = (lane >> 3) + 8 * i;    // pixel of the 32-pixel M-tile (2 rows x 16 cols)
                int oy = oy0 + (wm * MT + mt) * 2 + (px >> 4), ox = ox0 + (px & 15);
                if constexpr (UP == 2) { oy = 2 * oy + ph_dy; ox = 2 * ox + ph_dx; }      // (tile coordinates are source pixels)
                float4 v = *reinterpret_cast<const float4*>(s_tr + px * TP + cq * 4);
                if (nok4 && oy < p.H && ox < p.W) {
                    const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
                    if (p.residual != nullptr) {
                        const float rsc = p.res_scale;
                        v.x = fmaf(rv[mt % RG][i].x, rsc, v.x); v.y = fmaf(rv[mt % RG][i].y, rsc, v.y); v.z = fmaf(rv[mt % RG][i].z, rsc, v.z); v.w = fmaf(rv[mt % RG][i].w, rsc, v.w);
                    }
                    if constexpr (GNB) {
                        // dyhat = da * act'(u) * gamma,  u = gamma*yhat + beta,  yhat = (x - mu)*rstd   (same expressions as gn_bwd_pre_kernel)
                        const float4 xx = xv[mt % RG][i];
                        float d[4] = {v.x, v.y, v.z, v.w};
                        const float xs[4] = {xx.x, xx.y, xx.z, xx.w}, mm[4] = {g_mu.x, g_mu.y, g_mu.z, g_mu.w}, rr[4] = {g_rs.x, g_rs.y, g_rs.z, g_rs.w};
                        const float gg[4] = {g_ga.x, g_ga.y, g_ga.z, g_ga.w}, bb[4] = {g_be.x, g_be.y, g_be.z, g_be.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float yh = (xs[j] - mm[j]) * rr[j];
                            if (p.gnb_silu) {
                                const float u = yh * gg[j] + bb[j];
                                const float sgm = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                                d[j] *= sgm * (1.0f + u * (1.0f - sgm));
                            }
                            d[j] *= gg[j];
                            s1[j] += d[j]; s2[j] += d[j] * yh;
                        }
                        *reinterpret_cast<float4*>(p.out + pix * p.out_cstride + n4) = make_float4(d[0], d[1], d[2], d[3]);
                    } else {
                        *reinterpret_cast<float4*>(p.out + pix * p.out_cstride + n4) = (p.pack_from_p1 != 0 && n4 >= p.pack_from_p1 - 1) ? pack_hilo4(v) : v;
                        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                        s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();            // scratch is rewritten by the next tile
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (GNB || p.stats_out != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
            }
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = wn * 32 + cq * 4 + j;
                    s_red[(wm * BN + col) * 2] = s1[j]; s_red[(wm * BN + col) * 2 + 1] = s2[j];
                }
            }
        }
    }
    if (GNB || p.stats_out != nullptr) {
        __syncthreads();
        for (int t = tid; t < BN * 2; t += 256) {
            const int col = t >> 1, which = t & 1;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) tot += s_red[(w * BN + col) * 2 + which];
            const int n = UP == 2 ? (n0 + col) % p.Cout : n0 + col;      // (UP = 2: the four phases add into the same channel)
            if (n < p.Cout) {
                if constexpr (GNB) unsafeAtomicAdd(p.gnb_sum + ((size_t)b * p.gnb_Ct + p.gnb_coff + n) * 2 + which, (double)tot);
                else unsafeAtomicAdd(p.stats_out + ((size_t)b * p.Cout + n) * 2 + which, (double)tot);
            }
        }
    }
}

// tile of the only instantiated shape: 8 x 16 pixels x 128 output channels (4 M-tiles per weight fragment, one N-tile per wave)
static constexpr int DMA_TH = 8, DMA_BN = 128;

static int dma_mode() {      // PNPFLOW_HIP_DMA: 0 off, 1 (default) where the grid fills the chip, 2 wherever the shape is supported (tests)
    static const int m = getenv("PNPFLOW_HIP_DMA") ? atoi(getenv("PNPFLOW_HIP_DMA")) : 1;
    return m;
}

// the phase form of the nearest-x2 upsampling conv (UP = 2): shape test (what the engine asks before it builds the 4-phase 2 x 2 weight images) ...
bool conv_dma_phase_shape_ok(const ConvParams& p, int terms) {
    static const int mode = getenv("PNPFLOW_HIP_UPPHASE") ? atoi(getenv("PNPFLOW_HIP_UPPHASE")) : 1;      // test-only A/B switch: 0 = the 9-tap form on the upsampled view
    if (mode == 0 || dma_mode() == 0 || p.nseg != 1 || p.residual != nullptr || p.gnb_x != nullptr) return false;
    if (p.H != 2 * p.Hs || p.W != 2 * p.Ws || p.Hs % DMA_TH != 0 || p.Ws % 16 != 0 || p.Cout % 32 != 0 || (4 * p.Cout) % DMA_BN != 0) return false;
    if (p.seg[0].w_mode != 0 || p.seg[0].C % (terms == 3 ? 32 : 64) != 0) return false;
    if ((size_t)(p.Hs + 2) * (p.Ws + 2) * 128 >= (1ull << 31)) return false;
    if (dma_mode() >= 2) return true;
    return (long)p.B * (p.Hs / DMA_TH) * (p.Ws / 16) * (4 * p.Cout / DMA_BN) >= 512;
}
// ... and the launch test: one raw segment of taps = 4 that carries the packed phase image
static bool conv_dma_phase_supported(const ConvParams& p, int terms) {
    return conv_dma_phase_shape_ok(p, terms) && p.seg[0].taps == 4 && (terms == 3 ? p.seg[0].w16 : p.seg[0].w16h) != nullptr;
}

bool conv_dma_supported(const ConvParams& p, int stride, int up, int terms) {
    if (dma_mode() == 0 || stride != 1 || (up != 0 && up != 1 && up != 2)) return false;
    if (up == 2) return conv_dma_phase_supported(p, terms);
    if (p.Cout % DMA_BN != 0 || p.H % DMA_TH != 0 || p.W % 16 != 0) return false;
    if (up == 1 && (p.H != 2 * p.Hs || p.W != 2 * p.Ws)) return false;
    if (up == 0 && (p.H != p.Hs || p.W != p.Ws)) return false;
    const int kc = terms == 3 ? 32 : 64;
    for (int i = 1; i < p.nseg; ++i) if (p.seg[i].taps == 9 && p.seg[i - 1].taps == 1) return false;     // ring phase: 9-tap chunks run at phase 0
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& s = p.seg[i];
        if (s.w_mode != 0 || (terms == 3 ? s.w16 : s.w16h) == nullptr) return false;
        if (s.C % kc != 0 || (s.taps != 9 && s.taps != 1)) return false;
    }
    if ((size_t)(p.Hs + 2) * (p.Ws + 2) * 128 >= (1ull << 31)) return false;     // 32-bit piece offsets inside a plane
    if (dma_mode() >= 2) return true;
    const long wgs = (long)p.B * (p.H / DMA_TH) * (p.W / 16) * (p.Cout / DMA_BN);
    if (wgs < 512) return false;
    // The prep pass costs one read + one write of every operand element (measured: 30 us on average, 8 % of a forward when every
    // qualifying launch took this path - profiles/r03_ab_dma_variants.md), the conv it feeds is ~15 % faster than the register-staged
    // kernel.  It pays where a prepped element feeds enough multiply-adds: MACs per operand element = K_total * Cout / (channels
    // prepped per output pixel).  Same-box layer table: 2304 (16^2 x 256, one or two 3x3 segments) and the upsampling convs (operand
    // at quarter resolution) win, 1152 (32^2 x 128) and the launches that drag 1-tap shortcut segments along lose; the pure 1x1
    // launches win when Cout >= 3 C (q,k,v stacked), not at Cout = C.
    constexpr double min_ratio = 2000.0;      // (frozen in round 5: it was an environment knob while the layer table was measured)
    double macs = 0.0, elts = 0.0; bool all1 = true;
    for (int i = 0; i < p.nseg; ++i) { macs += (double)p.seg[i].taps * p.seg[i].C * p.Cout; elts += p.seg[i].C; all1 &= p.seg[i].taps == 1; }
    if (up) elts *= 0.25;
    if (all1) return p.Cout >= 3 * (int)elts;
    return macs / elts >= min_ratio;
}

template <int UP, int TERMS>
static hipError_t launch_dma_t(const ConvParams& p, hipStream_t stream) {
    constexpr int MT = 4, WM = 1, WN = 4;
    constexpr int PH = DMA_TH + 2, PW = 18, NINST = (PH * PW * 8 + 63) / 64, IPW = (NINST + 3) / 4, BUF = IPW * 4 * 1024;
    const size_t lds = 2 * BUF;
    const int tiles = UP == 2 ? p.B * (p.Hs / DMA_TH) * (p.Ws / 16) : p.B * (p.H / DMA_TH) * (p.W / 16);
    dim3 grid(tiles, (UP == 2 ? 4 * p.Cout : p.Cout) / DMA_BN);
    ConvParams pp = p; pp.xcd_map = 0;
    if (((long)grid.x * grid.y) % 8 == 0) { pp.xcd_map = 1; grid = dim3(grid.x * grid.y, 1); }
    if (p.gnb_x != nullptr) {
        if constexpr (UP == 0) {
            hipLaunchKernelGGL((conv_dma_kernel<MT, 1, WM, WN, UP, TERMS, true>), grid, dim3(256), lds, stream, pp);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    hipLaunchKernelGGL((conv_dma_kernel<MT, 1, WM, WN, UP, TERMS, false>), grid, dim3(256), lds, stream, pp);
    return hipGetLastError();
}

size_t conv_dma_a16_bytes(int B, int C, int Hs, int Ws, int terms) {
    return (size_t)B * C * (Hs + 2) * (Ws + 2) * (terms == 3 ? 4 : 2);
}

hipError_t launch_conv_dma(const ConvParams& p, int up, hipStream_t s, int terms) {
    for (int i = 0; i < p.nseg; ++i) if (p.seg[i].a16 == nullptr) return hipErrorInvalidValue;
    if (up == 2) return terms == 3 ? launch_dma_t<2, 3>(p, s) : launch_dma_t<2, 1>(p, s);
    if (terms == 3) return up ? launch_dma_t<1, 3>(p, s) : launch_dma_t<0, 3>(p, s);
    return up ? launch_dma_t<1, 1>(p, s) : launch_dma_t<0, 1>(p, s);
}

}  // namespace pf
