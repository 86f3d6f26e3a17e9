// Fused single-head self-attention core of SelfAttention.forward (pnpflow/models.py:148-160):
//     S = (q k^T) * C^-1/2 ;  P = softmax(S, dim=-1) ;  O = P v
// for q, k, v stacked as [B][T][3C] fp32 (the output of the fused q/k/v 1x1 convolution), O as [B][T][C].
// One launch replaces two per-image GEMM launches on the fp32 matrix path plus a softmax launch and the HBM round
// trips of S and P (T x T per image).  One workgroup = one image x 32 queries; the 4 waves split the keys (phase 1)
// and the output channels (phase 3).  Both contractions run on v_mfma_f32_32x32x16_f16 with hi+lo fp16 operand pairs
// (3 MFMAs per product, fp32 accumulate: fp32-equivalent, see conv_mfma16.hip):
//   phase 0  the 32 x C query block is split and parked in LDS in A-fragment order;
//   phase 1  k is streamed in 32-channel chunks: coalesced float4 loads (two chunks in flight), split once per
//            workgroup, staged in LDS in B-fragment order; each wave accumulates S for its T/4 keys;
//   phase 2  S (scaled) goes through LDS; 8 threads per query row do max / exp / sum / normalise and write P back as
//            hi+lo A fragments (over the query block, which is no longer needed);
//   phase 3  v is read straight from L2 (for a fixed key the 32 lanes of a fragment column are 128 contiguous bytes),
//            split in registers three steps ahead of its use; each wave accumulates O for its C/4 channels.
// Shapes: T = 128*TK, C = 128*TC with TK, TC in {1, 2}, plus T = 512 ... 4096 in steps of 256 through the key-blocked
// variant at the end of this file; everything else stays on the unfused path (engine.hip).
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// The value is made opaque first: with -ffp-contract=fast hipcc otherwise re-derives it from the multiply that produced
// it, rounds the STORED hi from the fp32 product (v_mul + v_cvt_pk) but subtracts a hi obtained by a fused
// v_fma_mixlo_f16 of the unrounded product; at a double-rounding tie the two differ by one f16 ulp and hi+lo is off by
// 2^-11 relative (found by the direct parity test of this kernel: 1 element in 180 000).
__device__ __forceinline__ void split4(float4 v, f16x4& h, f16x4& l) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
    l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
}

// Output of a workgroup's 32 x C tile: lane (l31, hi) holds rows (r & 3) + 8 (r >> 2) + 4 hi of channel wave * 32 TC + 32 nt + l31 (for a fixed r the 32 lanes
// of a half-wave write 128 contiguous bytes).  `row_scale`: the long variant's 1 / l per row (LDS), or null.  With the folded output projection (AttnParams)
// the tile leaves as  out = acc + bias[c] + residual[b][t][c].
// Packed keys / values (AttnParams::kv_packed: the stacked q,k,v conv stored hi | lo << 16 per element - the same split as split4, made once per element
// instead of once per query tile): a float4 of four packed words -> (4 hi halfs, 4 lo halfs) with two v_perm_b32 per pair
__device__ __forceinline__ void unpack4(float4 v, f16x4& h, f16x4& l) {
    const unsigned u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 hh = {__builtin_amdgcn_perm(u1, u0, 0x05040100u), __builtin_amdgcn_perm(u3, u2, 0x05040100u)};
    const u32x2 ll = {__builtin_amdgcn_perm(u1, u0, 0x07060302u), __builtin_amdgcn_perm(u3, u2, 0x07060302u)};
    h = __builtin_bit_cast(f16x4, hh); l = __builtin_bit_cast(f16x4, ll);
}
// eight values of one channel (fp32, or packed words): the B fragment pair of a P v step
template <bool packed>
__device__ __forceinline__ void v_frag(const float (&c)[8], f16x8& vh, f16x8& vl) {
    if constexpr (packed) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(c[i]);
        const u32x4 hh = {__builtin_amdgcn_perm(u[1], u[0], 0x05040100u), __builtin_amdgcn_perm(u[3], u[2], 0x05040100u),
                          __builtin_amdgcn_perm(u[5], u[4], 0x05040100u), __builtin_amdgcn_perm(u[7], u[6], 0x05040100u)};
        const u32x4 ll = {__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u),
                          __builtin_amdgcn_perm(u[5], u[4], 0x07060302u), __builtin_amdgcn_perm(u[7], u[6], 0x07060302u)};
        vh = __builtin_bit_cast(f16x8, hh); vl = __builtin_bit_cast(f16x8, ll);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { vh[i] = (_Float16)c[i]; vl[i] = (_Float16)(c[i] - (float)vh[i]); }
    }
}

template <int TC>
__device__ __forceinline__ void attn_store(const AttnParams& p, const f32x16 (&acc_o)[TC], const float* row_scale, int b, int T, int q0, int wave, int l31, int hi) {
    constexpr int C = 128 * TC;
    const int ch0 = wave * TC * 32 + l31;
    const size_t tile = ((size_t)b * T + q0) * C + ch0;
    float* obase = p.out + tile;
    if (p.residual == nullptr) {
#pragma unroll
        for (int nt = 0; nt < TC; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                obase[(size_t)row * C + nt * 32] = row_scale != nullptr ? acc_o[nt][r] * row_scale[row] : acc_o[nt][r];
            }
        return;
    }
    // every residual value is requested before the first store: loads and stores share one in-order counter, so a load issued behind stores waits
    // for their acknowledgements (interleaved load - add - store: +30 us on a 77 us kernel)
    const float* rbase = p.residual + tile;
    float rv[TC][16];
#pragma unroll
    for (int nt = 0; nt < TC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[nt][r] = rbase[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * C + nt * 32];
#pragma unroll
    for (int nt = 0; nt < TC; ++nt) {
        const float bias = p.bias != nullptr ? p.bias[ch0 + nt * 32] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float o = row_scale != nullptr ? acc_o[nt][r] * row_scale[row] : acc_o[nt][r];
            o = (o + bias) + rv[nt][r];
            obase[(size_t)row * C + nt * 32] = o;
            s1 += o; s2 += o * o;
        }
        // the tile's share of the result's GroupNorm statistics (32 rows per channel, fp32): a plain store per (tile, channel); launch_partial_stats adds the tiles
        if (p.stats_part != nullptr) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (hi == 0) *reinterpret_cast<float2*>(p.stats_part + (((size_t)b * (T / 32) + q0 / 32) * C + ch0 + nt * 32) * 2) = make_float2(s1, s2);
        }
    }
}

template <int TK, int TC, bool KVP>
__global__ __launch_bounds__(256) void attn_fused_kernel(const AttnParams p) {
    constexpr int T = 128 * TK, C = 128 * TC, BQ = 32;
    constexpr int QROW = C + 4;                 // dwords per query row: C/2 hi | C/2 lo | 4 pad (conflict-free ds_read_b128)
    constexpr int PROW = T + 4;                 // dwords per probability row: T/2 hi | T/2 lo | 4 pad
    constexpr int AROW = QROW > PROW ? QROW : PROW;
    constexpr int KROW = 36;                    // dwords per key row of a 32-channel chunk: 16 hi | 16 lo | 4 pad
    constexpr int SROW = T + 4;                 // floats per score row
    constexpr int KBUF = T * KROW, SBUF = BQ * SROW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* s_a = reinterpret_cast<uint32_t*>(smem_raw);           // [BQ][AROW]: q fragments, later P fragments
    uint32_t* s_k = s_a + BQ * AROW;                                  // [T][KROW]  : k chunk (phase 1)
    float* s_s = reinterpret_cast<float*>(s_k);                       // [BQ][SROW] : scores (phases 1 -> 2), aliases the k chunk
    static_assert(KBUF >= SBUF || true, "");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All T/32 query blocks of an image
    // re-read that image's k and v, so they are mapped to ONE XCD (image = 8*(slot / NQB) + xcd): with the natural order
    // the 8 blocks of an image sat on 8 different XCDs and every L2 fetched every k/v once more from the fabric.
    constexpr int NQB = T / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / NQB) * 8 + xcd, q0 = (slot % NQB) * BQ;
    if (b >= p.B) return;
    const float* base = p.qkv + (size_t)b * T * 3 * C;
    constexpr bool kvp = KVP;

    // k chunks 0 and 1 start their trip before anything else
    constexpr int K_PER = T * 8 / 256;          // float4 per thread per 32-channel chunk of k
    constexpr int NCH = C / 32;
    float4 rka[K_PER], rkb[K_PER];              // chunks c and c+1 in flight: a chunk is loaded two iterations before its use
    auto k_prefetch = [&](int chunk, float4 (&rk)[K_PER]) {
#pragma unroll
        for (int i = 0; i < K_PER; ++i) {
            const int idx = tid + i * 256, key = idx >> 3, c4 = idx & 7;
            rk[i] = *reinterpret_cast<const float4*>(base + (size_t)key * 3 * C + C + min(chunk, NCH - 1) * 32 + c4 * 4);
        }
    };
    k_prefetch(0, rka); k_prefetch(1, rkb);

    // ---- phase 0: query block -> LDS (hi | lo) ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < BQ * (C / 4) / 256; ++i) {
        const int idx = tid + i * 256, row = idx / (C / 4), c4 = idx % (C / 4);
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(q0 + row) * 3 * C + c4 * 4);
        f16x4 h, l; split4(v, h, l);
        *reinterpret_cast<f16x4*>(s_a + row * QROW + c4 * 2) = h;
        *reinterpret_cast<f16x4*>(s_a + row * QROW + C / 2 + c4 * 2) = l;
    }

    // ---- phase 1: S = q k^T ---------------------------------------------------------------------------------------
    f32x16 acc_s[TK];
#pragma unroll
    for (int nt = 0; nt < TK; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_s[nt][r] = 0.f;
    auto k_chunk = [&](int chunk, float4 (&rk)[K_PER]) {
        __syncthreads();                        // the previous chunk has been consumed (first pass: the q block is complete)
#pragma unroll
        for (int i = 0; i < K_PER; ++i) {
            const int idx = tid + i * 256, key = idx >> 3, c4 = idx & 7;
            f16x4 h, l;
            if constexpr (kvp) unpack4(rk[i], h, l); else split4(rk[i], h, l);
            *reinterpret_cast<f16x4*>(s_k + key * KROW + c4 * 2) = h;
            *reinterpret_cast<f16x4*>(s_k + key * KROW + 16 + c4 * 2) = l;
        }
        __syncthreads();
        k_prefetch(chunk + 2, rk);              // (clamped re-load of the last chunk at the end: unconditional loads keep the waitcnts static)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = chunk * 2 + jj;       // k16-step over the channels
            const f16x8 ah = *reinterpret_cast<const f16x8*>(s_a + l31 * QROW + j * 8 + hi * 4);
            const f16x8 al = *reinterpret_cast<const f16x8*>(s_a + l31 * QROW + C / 2 + j * 8 + hi * 4);
#pragma unroll
            for (int nt = 0; nt < TK; ++nt) {
                const int krow = (wave * TK + nt) * 32 + l31;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(s_k + krow * KROW + jj * 8 + hi * 4);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(s_k + krow * KROW + 16 + jj * 8 + hi * 4);
                acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_s[nt], 0, 0, 0);
                acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_s[nt], 0, 0, 0);
                acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_s[nt], 0, 0, 0);
            }
        }
    };
    for (int chunk = 0; chunk < NCH; chunk += 2) { k_chunk(chunk, rka); k_chunk(chunk + 1, rkb); }
    __syncthreads();                            // every wave is done with the last k chunk: its LDS becomes the score tile
#pragma unroll
    for (int nt = 0; nt < TK; ++nt) {
        const int col = (wave * TK + nt) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            s_s[row * SROW + col] = acc_s[nt][r] * p.scale;          // torch.bmm(q, k) * C**-0.5  (models.py:154)
        }
    }
    __syncthreads();

    // v fragments: keys j*16 + hi*8 + i of this lane's channel(s); a ring of 4 register sets, loaded three steps ahead
    // of their use - the first three leave before the softmax, which does not depend on them
    const float* vbase = base + 2 * C + wave * TC * 32 + l31;        // + key*3C + nt*32
    auto load_v = [&](int j, float (&dst)[TC][8]) {
#pragma unroll
        for (int nt = 0; nt < TC; ++nt)
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[nt][i] = vbase[(size_t)(j * 16 + hi * 8 + i) * 3 * C + nt * 32];
    };
    float v0[TC][8], v1[TC][8], v2[TC][8], v3[TC][8];
    load_v(0, v0); load_v(1, v1); load_v(2, v2);

    // ---- phase 2: softmax over the keys (models.py:155), P -> LDS as hi | lo A fragments ---------------------------
    {
        const int row = tid >> 3, part = tid & 7;
        float4 x[T / 32];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < T / 32; ++i) {
            x[i] = *reinterpret_cast<const float4*>(s_s + row * SROW + (i * 8 + part) * 4);
            m = fmaxf(fmaxf(m, fmaxf(x[i].x, x[i].y)), fmaxf(x[i].z, x[i].w));
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < T / 32; ++i) {
            x[i].x = expf(x[i].x - m); x[i].y = expf(x[i].y - m); x[i].z = expf(x[i].z - m); x[i].w = expf(x[i].w - m);
            s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < T / 32; ++i) {
            const float4 pv = make_float4(x[i].x * inv, x[i].y * inv, x[i].z * inv, x[i].w * inv);
            f16x4 h, l; split4(pv, h, l);
            const int k2 = (i * 8 + part) * 2;                       // dword index of these 4 keys
            *reinterpret_cast<f16x4*>(s_a + row * PROW + k2) = h;
            *reinterpret_cast<f16x4*>(s_a + row * PROW + T / 2 + k2) = l;
        }
    }
    __syncthreads();

    // ---- phase 3: O = P v -----------------------------------------------------------------------------------------
    f32x16 acc_o[TC];
#pragma unroll
    for (int nt = 0; nt < TC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[nt][r] = 0.f;
    auto v_step = [&](int j, float (&cur)[TC][8], float (&nxt)[TC][8]) {
        load_v(min(j + 3, T / 16 - 1), nxt);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 ph = *reinterpret_cast<const f16x8*>(s_a + l31 * PROW + j * 8 + hi * 4);
        const f16x8 pl = *reinterpret_cast<const f16x8*>(s_a + l31 * PROW + T / 2 + j * 8 + hi * 4);
#pragma unroll
        for (int nt = 0; nt < TC; ++nt) {
            f16x8 vh, vl;
            v_frag<kvp>(cur[nt], vh, vl);
            acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, acc_o[nt], 0, 0, 0);
            acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, acc_o[nt], 0, 0, 0);
            acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, acc_o[nt], 0, 0, 0);
        }
    };
    int j = 0;
    for (; j + 4 <= T / 16; j += 4) { v_step(j, v0, v3); v_step(j + 1, v1, v0); v_step(j + 2, v2, v1); v_step(j + 3, v3, v2); }
    static_assert((T / 16) % 4 == 0, "the ring of 4 realigns every 4 steps");

    attn_store<TC>(p, acc_o, nullptr, b, T, q0, wave, l31, hi);
}

// Long-sequence variant (T a multiple of 256 above 256: the 1024-token attention block of the 256x256 nets).  The keys are
// walked in blocks of 256 with an online softmax: per block S = q k^T (as above), the running row maximum m and row sum l
// are updated, P = exp(S - m) stays unnormalised, the accumulated output is rescaled by exp(m_old - m_new) and O += P v;
// the division by l happens once at the end.  The query block is re-staged per key block (P overwrites it in LDS).
template <int TC, bool KVP>
__global__ __launch_bounds__(256) void attn_fused_long_kernel(const AttnParams p) {
    constexpr int TB = 256, TK = 2, C = 128 * TC, BQ = 32;      // keys per block: 2 tiles of 32 per wave
    constexpr int QROW = C + 4, PROW = TB + 4, AROW = QROW > PROW ? QROW : PROW, KROW = 36, SROW = TB + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint32_t* s_a = reinterpret_cast<uint32_t*>(smem_raw);           // [BQ][AROW]: q fragments, then P fragments (per key block)
    uint32_t* s_k = s_a + BQ * AROW;                                  // [TB][KROW] : k chunk
    float* s_s = reinterpret_cast<float*>(s_k);                       // [BQ][SROW] : scores, aliases the k chunk
    float* s_scale = reinterpret_cast<float*>(s_k + TB * KROW);       // [BQ]       : per-row rescale / final 1/l
    const int T = p.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int nqb = T / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / nqb) * 8 + xcd, q0 = (slot % nqb) * BQ;       // all query blocks of an image on one XCD (see above)
    if (b >= p.B) return;
    const float* base = p.qkv + (size_t)b * T * 3 * C;
    constexpr bool kvp = KVP;

    constexpr int K_PER = TB * 8 / 256, NCH = C / 32;
    float4 rka[K_PER], rkb[K_PER];
    f32x16 acc_o[TC];
#pragma unroll
    for (int nt = 0; nt < TC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[nt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                             // of row tid>>3 (the 8 threads of a row hold copies)

    for (int kb = 0; kb < T / TB; ++kb) {
        const float* kbase = base + (size_t)kb * TB * 3 * C;          // keys / values of this block
        auto k_prefetch = [&](int chunk, float4 (&rk)[K_PER]) {
#pragma unroll
            for (int i = 0; i < K_PER; ++i) {
                const int idx = tid + i * 256, key = idx >> 3, c4 = idx & 7;
                rk[i] = *reinterpret_cast<const float4*>(kbase + (size_t)key * 3 * C + C + min(chunk, NCH - 1) * 32 + c4 * 4);
            }
        };
        k_prefetch(0, rka); k_prefetch(1, rkb);
        // query block -> LDS (every wave is past the previous block's P: barrier at the end of the loop body)
#pragma unroll
        for (int i = 0; i < BQ * (C / 4) / 256; ++i) {
            const int idx = tid + i * 256, row = idx / (C / 4), c4 = idx % (C / 4);
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(q0 + row) * 3 * C + c4 * 4);
            f16x4 h, l; split4(v, h, l);
            *reinterpret_cast<f16x4*>(s_a + row * QROW + c4 * 2) = h;
            *reinterpret_cast<f16x4*>(s_a + row * QROW + C / 2 + c4 * 2) = l;
        }
        f32x16 acc_s[TK];
#pragma unroll
        for (int nt = 0; nt < TK; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[nt][r] = 0.f;
        auto k_chunk = [&](int chunk, float4 (&rk)[K_PER]) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K_PER; ++i) {
                const int idx = tid + i * 256, key = idx >> 3, c4 = idx & 7;
                f16x4 h, l;
                if constexpr (kvp) unpack4(rk[i], h, l); else split4(rk[i], h, l);
                *reinterpret_cast<f16x4*>(s_k + key * KROW + c4 * 2) = h;
                *reinterpret_cast<f16x4*>(s_k + key * KROW + 16 + c4 * 2) = l;
            }
            __syncthreads();
            k_prefetch(chunk + 2, rk);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = chunk * 2 + jj;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(s_a + l31 * QROW + j * 8 + hi * 4);
                const f16x8 al = *reinterpret_cast<const f16x8*>(s_a + l31 * QROW + C / 2 + j * 8 + hi * 4);
#pragma unroll
                for (int nt = 0; nt < TK; ++nt) {
                    const int krow = (wave * TK + nt) * 32 + l31;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(s_k + krow * KROW + jj * 8 + hi * 4);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(s_k + krow * KROW + 16 + jj * 8 + hi * 4);
                    acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_s[nt], 0, 0, 0);
                    acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_s[nt], 0, 0, 0);
                    acc_s[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_s[nt], 0, 0, 0);
                }
            }
        };
        for (int chunk = 0; chunk < NCH; chunk += 2) { k_chunk(chunk, rka); k_chunk(chunk + 1, rkb); }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < TK; ++nt) {
            const int col = (wave * TK + nt) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_s[((r & 3) + 8 * (r >> 2) + 4 * hi) * SROW + col] = acc_s[nt][r] * p.scale;
        }
        // the first three v fragments of this key block leave before the softmax update
        const float* vbase = kbase + 2 * C + wave * TC * 32 + l31;
        auto load_v = [&](int j, float (&dst)[TC][8]) {
#pragma unroll
            for (int nt = 0; nt < TC; ++nt)
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[nt][i] = vbase[(size_t)(j * 16 + hi * 8 + i) * 3 * C + nt * 32];
        };
        float v0[TC][8], v1[TC][8], v2[TC][8], v3[TC][8];
        load_v(0, v0); load_v(1, v1); load_v(2, v2);
        __syncthreads();
        {   // online softmax update of row tid>>3
            const int row = tid >> 3, part = tid & 7;
            float4 x[TB / 32];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < TB / 32; ++i) {
                x[i] = *reinterpret_cast<const float4*>(s_s + row * SROW + (i * 8 + part) * 4);
                mx = fmaxf(fmaxf(mx, fmaxf(x[i].x, x[i].y)), fmaxf(x[i].z, x[i].w));
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float m_new = fmaxf(m_run, mx);
            const float sc = expf(m_run - m_new);                    // 0 on the first block (m_run = -inf)
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < TB / 32; ++i) {
                x[i].x = expf(x[i].x - m_new); x[i].y = expf(x[i].y - m_new); x[i].z = expf(x[i].z - m_new); x[i].w = expf(x[i].w - m_new);
                sm += (x[i].x + x[i].y) + (x[i].z + x[i].w);
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            l_run = l_run * sc + sm; m_run = m_new;
            if (part == 0) s_scale[row] = sc;
#pragma unroll
            for (int i = 0; i < TB / 32; ++i) {
                f16x4 h, l; split4(x[i], h, l);
                const int k2 = (i * 8 + part) * 2;
                *reinterpret_cast<f16x4*>(s_a + row * PROW + k2) = h;
                *reinterpret_cast<f16x4*>(s_a + row * PROW + TB / 2 + k2) = l;
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < TC; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[nt][r] *= s_scale[(r & 3) + 8 * (r >> 2) + 4 * hi];
        auto v_step = [&](int j, float (&cur)[TC][8], float (&nxt)[TC][8]) {
            load_v(min(j + 3, TB / 16 - 1), nxt);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 ph = *reinterpret_cast<const f16x8*>(s_a + l31 * PROW + j * 8 + hi * 4);
            const f16x8 pl = *reinterpret_cast<const f16x8*>(s_a + l31 * PROW + TB / 2 + j * 8 + hi * 4);
#pragma unroll
            for (int nt = 0; nt < TC; ++nt) {
                f16x8 vh, vl;
                v_frag<kvp>(cur[nt], vh, vl);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, acc_o[nt], 0, 0, 0);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, acc_o[nt], 0, 0, 0);
                acc_o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, acc_o[nt], 0, 0, 0);
            }
        };
        for (int j = 0; j + 4 <= TB / 16; j += 4) { v_step(j, v0, v3); v_step(j + 1, v1, v0); v_step(j + 2, v2, v1); v_step(j + 3, v3, v2); }
        __syncthreads();                                              // P and s_scale are free for the next key block
    }
    if ((tid & 7) == 0) s_scale[tid >> 3] = 1.0f / l_run;
    __syncthreads();
    attn_store<TC>(p, acc_o, s_scale, b, T, q0, wave, l31, hi);
}

template <int TC, bool KVP>
hipError_t launch_attn_long_cfg(const AttnParams& p, hipStream_t s) {
    constexpr int C = 128 * TC;
    constexpr int AROW = (C > 256 ? C : 256) + 4;
    const size_t lds = (size_t)(32 * AROW + 256 * 36 + 32) * 4;
    static unsigned long long attr_set = 0ull;
    auto kern = attn_fused_long_kernel<TC, KVP>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3((p.T / 32) * ((p.B + 7) / 8) * 8), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int TK, int TC, bool KVP>
hipError_t launch_attn_cfg(const AttnParams& p, hipStream_t s) {
    constexpr int T = 128 * TK, C = 128 * TC;
    constexpr int AROW = (C > T ? C : T) + 4;
    constexpr int KBUF = T * 36, SBUF = 32 * (T + 4);
    const size_t lds = (size_t)(32 * AROW + (KBUF > SBUF ? KBUF : SBUF)) * 4;
    static unsigned long long attr_set = 0ull;
    auto kern = attn_fused_kernel<TK, TC, KVP>;
    { hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(kern), attr_set, 160 * 1024); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3((T / 32) * ((p.B + 7) / 8) * 8), dim3(256), lds, s, p);
    return hipGetLastError();
}

}  // namespace

// per-image sum of the attention core's per-tile partial statistics part[B][nparts][C][2] (fp32) in tile order, in fp64: the GroupNorm statistics of the
// folded block's result (plain stores, deterministic)
__global__ __launch_bounds__(256) void partial_stats_kernel(const float* __restrict__ part, double* __restrict__ stats, int nparts, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0.0, q = 0.0;
        for (int k = 0; k < nparts; ++k) {
            const float2 v = *reinterpret_cast<const float2*>(part + (((size_t)b * nparts + k) * C + c) * 2);
            a += (double)v.x; q += (double)v.y;
        }
        double* st = stats + ((size_t)b * C + c) * 2;
        st[0] = a; st[1] = q;
    }
}

hipError_t launch_partial_stats(const float* part, double* stats, int B, int nparts, int C, hipStream_t s) {
    if (part == nullptr || stats == nullptr || B <= 0 || nparts <= 0 || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(partial_stats_kernel, dim3(B), dim3(256), 0, s, part, stats, nparts, C);
    return hipGetLastError();
}

bool attn_fused_supported(int T, int C) { return (T == 128 || (T >= 256 && T % 256 == 0 && T <= 4096)) && (C == 128 || C == 256); }

hipError_t launch_attn_fused(const AttnParams& p, hipStream_t s) {
    if (!attn_fused_supported(p.T, p.C)) return hipErrorInvalidValue;
    if (p.kv_packed) {
        if (p.T > 256) return p.C == 256 ? launch_attn_long_cfg<2, true>(p, s) : launch_attn_long_cfg<1, true>(p, s);
        if (p.T == 256) return p.C == 256 ? launch_attn_cfg<2, 2, true>(p, s) : launch_attn_cfg<2, 1, true>(p, s);
        return p.C == 256 ? launch_attn_cfg<1, 2, true>(p, s) : launch_attn_cfg<1, 1, true>(p, s);
    }
    if (p.T > 256) return p.C == 256 ? launch_attn_long_cfg<2, false>(p, s) : launch_attn_long_cfg<1, false>(p, s);     // online softmax over key blocks
    if (p.T == 256) return p.C == 256 ? launch_attn_cfg<2, 2, false>(p, s) : launch_attn_cfg<2, 1, false>(p, s);
    return p.C == 256 ? launch_attn_cfg<1, 2, false>(p, s) : launch_attn_cfg<1, 1, false>(p, s);
}

}  // namespace pf
