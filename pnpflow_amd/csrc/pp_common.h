// Device helpers shared by the persistent convolutions conv_pp.hip (Cout = 32) and conv_sp.hip (Cout = 64 / 128): vector types, the staging
// SiLU, the in-register 4 x 4 lane transpose of the epilogue, tile geometry.
#pragma once
#include "pf_common.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(4))) float* pp_float_cptr;

// same expressions as conv_mfma16.hip (bit-identical staging)
__device__ __forceinline__ float silu_pp(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// SiLU of four values with the four dependent chains (x * -log2(e) -> 2^x -> + 1 -> 1 / x -> * x) INTERLEAVED: under the register pressure of
// these kernels hipcc runs the chains one after the other through a single temporary (mul, exp, nop, add, rcp per element, every
// instruction waiting for the one before it - r4: the staging of a 16-channel patch was 2.5 k cycles alone); the same instructions in
// the same arithmetic order per element, so the values are bit-identical to silu_pp.  No wait states are needed: a transcendental
// result is read at least three instructions after it was issued.
__device__ __forceinline__ void silu4_pp(float4& v) {
    float t0, t1, t2, t3;
    asm("v_mul_f32 %4, 0xbfb8aa3b, %0\n\tv_mul_f32 %5, 0xbfb8aa3b, %1\n\tv_mul_f32 %6, 0xbfb8aa3b, %2\n\tv_mul_f32 %7, 0xbfb8aa3b, %3\n\t"
        "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
        "v_add_f32 %4, 1.0, %4\n\tv_add_f32 %5, 1.0, %5\n\tv_add_f32 %6, 1.0, %6\n\tv_add_f32 %7, 1.0, %7\n\t"
        "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7\n\t"
        "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5\n\tv_mul_f32 %2, %2, %6\n\tv_mul_f32 %3, %3, %7"
        : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
}

// fp16 hi / lo split of four fp32 values in six instructions: hi = RNE16(x) (v_cvt_pk_f16_f32, two values per instruction), lo = RNE16(x - hi)
// as ONE mixed-precision fma per value (v_fma_mix{lo,hi}_f16: fp16 hi * -1 + fp32 x, rounded once to the lower / upper half of the
// destination - x - hi is exact in fp32, so this is the same value as (_Float16)(x - (float)hi)).  hipcc's translation of the C++
// expression converts hi back to fp32, subtracts and converts again (and re-converts one pair singly): 13 instructions.
__device__ __forceinline__ void split4_pp(const float4& v, unsigned& h01, unsigned& h23, unsigned& l01, unsigned& l23) {
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}

// streaming (non-temporal) 16-byte accesses for tensors a launch touches once
typedef float pp_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(void* dst, const float4& v) { pp_f4v t_ = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t_, reinterpret_cast<pp_f4v*>(dst)); }
__device__ __forceinline__ float4 nt_load4(const void* src) { const pp_f4v t_ = __builtin_nontemporal_load(reinterpret_cast<const pp_f4v*>(src)); return make_float4(t_.x, t_.y, t_.z, t_.w); }

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// 4 x 4 transpose between the registers a0..a3 and the four lanes {k, k + 8, k + 16, k + 24} of a 32-lane half (k = lane & 7): in, a[i] of
// lane group g holds element (i, g); out, a[g'] of lane group i' holds element (i', g').  Lane bit 4 is exchanged with v_permlane16_swap
// (one instruction per register pair, gfx950), lane bit 3 with a DPP row rotation by 8 + selects.  With MFMA column n = 8 g + k carrying
// output channel 4 k + g (the weight image is packed that way) a lane ends up with FOUR CONSECUTIVE channels of one pixel, and the
// eight lanes k = 0..7 with the pixel's whole 128-byte row: stores and residual loads are 64 contiguous bytes per lane quad (the quad-
// local DPP transpose of the first version gave every lane of a quad a different pixel: 4 x the vector-memory requests, 107 of 255 us).
__device__ __forceinline__ void oct_transpose(float& a0, float& a1, float& a2, float& a3, bool bit3) {
    constexpr int ROR8 = 0x128;                           // row_ror:8 = lane ^ 8 inside a row of 16
    // (inline asm: with the builtin hipcc 7.2 folds the SECOND result of llvm.amdgcn.permlane16.swap onto the first in this function - seen
    // in the IR at -O1; the pads are the 2 wait states a VALU write needs before a v_permlane read, and before the DPP reads that follow)
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(b0), "+v"(b2), "+v"(b1), "+v"(b3));
    // (every DPP read is executed by ALL lanes before the selects)
    const float d0 = dpp_quad<ROR8>(b0), d1 = dpp_quad<ROR8>(b1), d2 = dpp_quad<ROR8>(b2), d3 = dpp_quad<ROR8>(b3);
    a0 = bit3 ? d1 : b0; a1 = bit3 ? b1 : d0; a2 = bit3 ? d3 : b2; a3 = bit3 ? b3 : d2;
}

struct PPTile { int b, oy0, ox0, edge; };      // edge bits: 1 top, 2 bottom, 4 left, 8 right (patch rows / columns outside the image)
template <int K> struct ic { static constexpr int value = K; };

}  // namespace pf
