// Small kernels of the U-Net velocity field that are not dense contractions:
// image-boundary convs (3->ch and ch->3, VALU), time embedding, softmax rows,
// stand-alone per-channel statistics.  Reference: pnpflow/models.py:253-299, 442-495.
#include "pf_common.h"

namespace pf {

__device__ __forceinline__ float silu_acc(float x) { return x / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------
// begin_conv: NCHW image -> NHWC [B][H][W][C], 3x3 pad 1 (models.py:358, 451), plus the per-channel
// (sum, sumsq) of the result for the first GroupNorm.  One thread per output pixel with all C=32 output
// channels in registers and the weights in LDS; the tile is transposed through LDS so that the NHWC rows are
// written as contiguous float4 runs, and the statistics are reduced per workgroup (64 fp64 atomics).
// HBM-bound: 128 B written per pixel.
// ------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void begin_conv_kernel(const EdgeConvParams p) {
    __shared__ float s_t[256][C + 1];
    __shared__ float s_part[8][C][2];
    const int tid = threadIdx.x;
    const float* __restrict__ wgt = p.w;          // wave-uniform addresses: scalar loads, no LDS traffic
    const float* __restrict__ bias = p.bias;
    const int HW = p.H * p.W;
    const int pix0 = blockIdx.x * 256;
    const int pix = pix0 + tid;
    const int b = blockIdx.y;
    const bool valid = pix < HW;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = valid ? bias[c] : 0.f;
    if (valid) {
        const int y = pix / p.W, x = pix % p.W;
        for (int ci = 0; ci < p.Cimg; ++ci) {
            const float* img = p.in + ((size_t)b * p.Cimg + ci) * HW;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                float v = 0.f;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = img[yy * p.W + xx];
                const float* __restrict__ w = wgt + (tap * p.Cimg + ci) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(v, w[c], acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) s_t[tid][c] = acc[c];
    __syncthreads();
    // the 256 pixels x C channels of this workgroup are one contiguous NHWC run
    float* o = p.out + ((size_t)b * HW + pix0) * C;
    const int nvalid = min(256, HW - pix0);
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        const int idx = k * 256 + tid, px = idx / (C / 4), q = idx % (C / 4);
        if (px < nvalid)
            *reinterpret_cast<float4*>(o + (size_t)px * C + q * 4) = make_float4(s_t[px][q * 4], s_t[px][q * 4 + 1], s_t[px][q * 4 + 2], s_t[px][q * 4 + 3]);
    }
    if (p.stats_out != nullptr) {
        const int c = tid % C, part = tid / C;        // 256 / C = 8 parts of 32 pixels (C = 32)
        float s1 = 0.f, s2 = 0.f;
        for (int px = part * (256 / (256 / C)); px < (part + 1) * (256 / (256 / C)); ++px) { const float v = s_t[px][c]; s1 += v; s2 += v * v; }
        s_part[part][c][0] = s1; s_part[part][c][1] = s2;
        __syncthreads();
        if (tid < 2 * C) {
            const int cc = tid >> 1, which = tid & 1;
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < 256 / C; ++k) tot += (double)s_part[k][cc][which];
            unsafeAtomicAdd(p.stats_out + ((size_t)b * C + cc) * 2 + which, tot);
        }
    }
}


// ------------------------------------------------------------------------------------
// begin_conv, round 5 (forward, C = 32, Cimg = 1 or 3): the Cimg x 9 -> 32 contraction per pixel on the matrix pipe, as end_conv2_kernel
// below - 32 pixels x K = 32 (27 used: input channel x tap) x 32 channels is one split-fp16 32 x 32 x 32 MFMA tile (image x 2^3, weights
// x 2^8; patches that reach beyond 2^12 are scaled down by their own exponent: fp32-equivalent over the fp32 range).  The round-1 kernel above spends 864 scalar-operand FMAs per pixel and runs at 0.25 of the
// write bandwidth (620 us for 1.34 GB; a lane-per-channel-quad v_pk_fma_f32 form measured 669 us - tools/ubench/edge_probe.hip); here a
// lane (pixel p = lane & 31, half h = lane >> 5) gathers its 16 im2col values from the LDS image patch, splits them, and the 16 results of
// its channel go through an LDS tile from which the workgroup writes whole 128-byte NHWC pixel rows.  Persistent: a workgroup walks a
// contiguous range of 16 x 16-pixel tiles, the next tile's patch is fetched into registers under the current tile's work, the
// GroupNorm statistics are carried per lane (fp32 per tile, fp64 across tiles) and flushed when the image changes.  The adjoint use
// (backward of end_conv: no weight image) stays on the round-1 kernel.
// ------------------------------------------------------------------------------------
typedef _Float16 bc_h8 __attribute__((ext_vector_type(8)));
typedef float bc_f16v __attribute__((ext_vector_type(16)));

template <int CIMG>
__global__ __launch_bounds__(256) void begin_conv2_kernel(const EdgeConvParams p) {
    constexpr int C = 32, PW = 18, NIMG = CIMG * PW * PW, K27 = 9 * CIMG, OP = 36, NLD = (NIMG + 255) / 256;
    __shared__ float s_img[NIMG + 4];      // [ci][18][18] + a zero slot (the padding of K to 32)
    __shared__ __attribute__((aligned(16))) float s_o[256 * OP];
    // operand range: the patch is split into fp16 hi / lo after a power-of-two scale.  2^3 for patches below 4 096 (every image:
    // the values are those of the fixed scale, bit for bit); a patch that reaches further is scaled down by its own exponent, as the other
    // convs' per-segment operand scales do - |x| up to the fp32 range instead of inf beyond 8 188.  s_amax[tile parity]: bit pattern of
    // the patch's max |x| (one LDS atomic per wave; the slot of the next tile is cleared between this tile's two barriers).
    __shared__ unsigned s_amax[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 15) / 16, per_img = tiles_x * tiles_y;
    const long total = (long)per_img * p.B;
    const long t0 = total * blockIdx.x / gridDim.x, t1 = total * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    if (tid == 0) { s_img[NIMG] = 0.f; s_amax[0] = 0u; s_amax[1] = 0u; }
    __syncthreads();
    // B fragments: [k-step][hi | lo][lane][8 halfs] (engine.hip packs MFMA k of (step s, half h, j) = 16 s + 8 h + j = input channel * 9 + tap)
    const bc_h8* wimg = reinterpret_cast<const bc_h8*>(p.w16);
    const bc_h8 wh0 = wimg[0 * 64 + lane], wl0 = wimg[1 * 64 + lane], wh1 = wimg[2 * 64 + lane], wl1 = wimg[3 * 64 + lane];
    const float bias_n = p.bias[pl];
    // LDS offsets of this lane's 16 im2col values relative to its pixel's patch origin (the zero slot for k >= 9 Cimg)
    int offk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = 16 * (i >> 3) + 8 * h + (i & 7);
        const int ci = k / 9, tap = k - 9 * ci;
        offk[i] = k < K27 ? (ci * PW + tap / 3) * PW + tap % 3 : -1;
    }
    const int q = tid & 7;
    double d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0};
    int cur_b = -1;
    auto flush = [&]() {
        if (p.stats_out == nullptr || cur_b < 0) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a = d1[j], s2 = d2[j];
            a += __shfl_xor(a, 8); s2 += __shfl_xor(s2, 8);
            a += __shfl_xor(a, 16); s2 += __shfl_xor(s2, 16);
            a += __shfl_xor(a, 32); s2 += __shfl_xor(s2, 32);
            if (lane < 8) {
                double* dst = p.stats_out + ((size_t)cur_b * C + 4 * q + j) * 2;
                unsafeAtomicAdd(dst, a); unsafeAtomicAdd(dst + 1, s2);
            }
            d1[j] = 0; d2[j] = 0;
        }
    };
    float nxt[NLD];
    auto fetch = [&](long t) {
        const int b = (int)(t / per_img), r = (int)(t - (long)b * per_img);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int ci = i / (PW * PW), rr = i - ci * (PW * PW), py = rr / PW, px = rr - py * PW;
            const int gy = ty * 16 - 1 + py, gx = tx * 16 - 1 + px;
            float v = 0.f;
            if (i < NIMG && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = p.in[(((size_t)b * CIMG + ci) * p.H + gy) * p.W + gx];
            nxt[k] = v;
        }
    };
    fetch(t0);
    for (long t = t0; t < t1; ++t) {
        const int b = (int)(t / per_img), r = (int)(t - (long)b * per_img);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int y0 = ty * 16, x0 = tx * 16;
        const int par = (int)((t - t0) & 1);
        if (b != cur_b) { flush(); cur_b = b; }
        {
            unsigned am = 0u;
#pragma unroll
            for (int k = 0; k < NLD; ++k) am = max(am, __float_as_uint(nxt[k]) & 0x7fffffffu);
            am = max(am, (unsigned)__shfl_xor((int)am, 1)); am = max(am, (unsigned)__shfl_xor((int)am, 2)); am = max(am, (unsigned)__shfl_xor((int)am, 4));
            am = max(am, (unsigned)__shfl_xor((int)am, 8)); am = max(am, (unsigned)__shfl_xor((int)am, 16)); am = max(am, (unsigned)__shfl_xor((int)am, 32));
            if (lane == 0) atomicMax(&s_amax[par], am);
        }
        __syncthreads();      // (the previous tile's patch and output tile have been read)
#pragma unroll
        for (int k = 0; k < NLD; ++k) if (tid + 256 * k < NIMG) s_img[tid + 256 * k] = nxt[k];
        if (tid == 0) s_amax[par ^ 1] = 0u;
        __syncthreads();
        // scale 2^3 while the patch stays below 2^12, else 2^(15 - e) / 2^... so that max |x| * scale < 2^15 (e = the biased exponent's excess)
        const int ex = (int)(s_amax[par] >> 23) - 127;                    // floor(log2(max |x|)) (255 - 127 for inf / NaN: the result is non-finite either way)
        const int sh = ex >= 12 ? ex - 11 : 0;                             // halvings of the fixed scale
        const float sc_in = __uint_as_float((unsigned)(127 + 3 - min(sh, 120)) << 23);
        const float sc_out = __uint_as_float((unsigned)(127 - 11 + min(sh, 120)) << 23);      // 1 / (2^8 * sc_in)
        if (t + 1 < t1) fetch(t + 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int mt = 2 * wv + mi;                       // M-tile: rows 2 mt, 2 mt + 1 of the tile
            const int base = (2 * mt + (pl >> 4)) * PW + (pl & 15);
            float a[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = s_img[offk[i] < 0 ? NIMG : base + offk[i]] * sc_in;
            bc_h8 ah0, al0, ah1, al1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ah0[j] = (_Float16)a[j]; al0[j] = (_Float16)(a[j] - (float)ah0[j]);
                ah1[j] = (_Float16)a[8 + j]; al1[j] = (_Float16)(a[8 + j] - (float)ah1[j]);
            }
            bc_f16v acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, wh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wl0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, wh1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wl1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wh1, acc, 0, 0, 0);
            // D: register i of lane (column = channel pl, half h) is pixel row 8 (i / 4) + 4 h + (i % 4) of the M-tile
#pragma unroll
            for (int i = 0; i < 16; ++i) s_o[(mt * 32 + 8 * (i >> 2) + 4 * h + (i & 3)) * OP + pl] = acc[i] * sc_out + bias_n;
        }
        __syncthreads();
        float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (k * 256 + tid) >> 3;              // pixel of the tile (row px / 16, column px % 16); this thread's channel quad is q
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            if (y < p.H && x < p.W) {
                const float4 v = *reinterpret_cast<const float4*>(&s_o[px * OP + 4 * q]);
                *reinterpret_cast<float4*>(p.out + (((size_t)b * p.H + y) * p.W + x) * C + 4 * q) = v;
                s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { d1[j] += (double)s1[j]; d2[j] += (double)s2[j]; }
    }
    flush();
}

hipError_t launch_begin_conv(const EdgeConvParams& p, hipStream_t s) {
    if (p.C != 32 || p.Cimg > 3) return hipErrorInvalidValue;
    const long tiles = (long)p.B * ((p.H + 15) / 16) * ((p.W + 15) / 16);
    const int cus = device_cu_count();      // (0 when the attribute query fails: the one-pixel-per-thread kernel then - ADVICE r5)
    if (cus > 0 && p.w16 != nullptr && (p.Cimg == 1 || p.Cimg == 3) && tiles >= cus) {      // (below one tile per CU the one-pixel-per-thread kernel is the faster one: 13 vs 24 us at 3 x 100 x 52)
        const long want = 3L * cus;
        const int grid = (int)(tiles < want ? tiles : want);
        if (p.Cimg == 3) hipLaunchKernelGGL(begin_conv2_kernel<3>, dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(begin_conv2_kernel<1>, dim3(grid), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    dim3 grid((p.H * p.W + 255) / 256, p.B);
    hipLaunchKernelGGL(begin_conv_kernel<32>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// end_conv: GroupNorm -> SiLU -> 3x3 conv C->Cimg, NHWC in, NCHW image out (models.py:428-433, 492).
// Two stages per 16x16 output tile:  (1) every pixel of the 18x18 halo patch is read once from HBM by one thread
// (its C channels stay in registers), normalised/activated, and contracted with the 9*Cimg weight vectors into
// t[pixel][tap][co] (the weights are wave-uniform: scalar loads, no LDS traffic);  (2) each output pixel gathers
// its 9 taps of t from LDS.  The activations never go through LDS (the previous version staged the 18x18xC patch
// and re-read it 9x together with the weights: LDS-bandwidth-bound at 6x this kernel's time).
// ------------------------------------------------------------------------------------
template <int C, int CIMG>
__global__ __launch_bounds__(256) void end_conv_kernel(const EdgeConvParams p) {
    constexpr int PW = 18, PP = PW * PW, NT_ = 9 * CIMG, TP = NT_ + 1;
    __shared__ float s_t[PP * TP];
    __shared__ float s_sc[C], s_sh[C];
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 15) / 16;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x;
    const bool raw = p.stats == nullptr;   // raw: plain 3x3 conv (used as the adjoint of begin_conv)
    if (tid < C) {
        float sc = 1.f, sh = 0.f;
        if (!raw) {
            const int g = tid / p.gn_cpg;
            double s = 0.0, ss = 0.0;
            for (int j = g * p.gn_cpg; j < (g + 1) * p.gn_cpg; ++j) {
                const double* st = p.stats + ((size_t)b * C + j) * 2;
                s += st[0]; ss += st[1];
            }
            const double N = (double)p.gn_cpg * p.H * p.W;
            const double mean = s / N;
            double var = ss / N - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
            sc = p.gamma[tid] * rstd;
            sh = p.beta[tid] - (float)mean * sc;
        }
        s_sc[tid] = sc; s_sh[tid] = sh;
    }
    __syncthreads();
    const int oy0 = ty * 16, ox0 = tx * 16;
    const float* __restrict__ w = p.w;
    for (int pix = tid; pix < PP; pix += 256) {         // wave-uniform trip count (waves 0-1 take a second pixel)
        const int gy = oy0 - 1 + pix / PW, gx = ox0 - 1 + pix % PW;
        float t[NT_];
#pragma unroll
        for (int k = 0; k < NT_; ++k) t[k] = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            const float4* src = reinterpret_cast<const float4*>(p.in + ((size_t)(b * p.H + gy) * p.W + gx) * C);
            float a[C];
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                const float4 v = src[q];
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
            if (!raw) {
#pragma unroll
                for (int c = 0; c < C; ++c) { const float u = a[c] * s_sc[c] + s_sh[c]; a[c] = u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
            }
#pragma unroll
            for (int k = 0; k < NT_; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) acc = fmaf(a[c], w[k * C + c], acc);
                t[k] = acc;
            }
        }
#pragma unroll
        for (int k = 0; k < NT_; ++k) s_t[pix * TP + k] = t[k];
    }
    __syncthreads();
    const int ly = tid / 16, lx = tid % 16;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float acc[CIMG];
#pragma unroll
    for (int co = 0; co < CIMG; ++co) acc[co] = p.bias[co];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* tp = s_t + ((ly + tap / 3) * PW + lx + tap % 3) * TP + tap * CIMG;
#pragma unroll
        for (int co = 0; co < CIMG; ++co) acc[co] += tp[co];
    }
    if (oy < p.H && ox < p.W) {
#pragma unroll
        for (int co = 0; co < CIMG; ++co) p.out[((size_t)(b * CIMG + co) * p.H + oy) * p.W + ox] = acc[co];
    }
}


// ------------------------------------------------------------------------------------
// end_conv, round 5 (C = 32, GroupNorm + SiLU input; the raw adjoint form stays on the kernel above): stage 1 - the 32 -> 9 Cimg
// contraction per halo pixel, 864 scalar-operand FMAs per pixel in the round-1 kernel - runs on the matrix pipe: 32 halo pixels x 32
// channels x 32 columns (27 used: tap x image channel) is ONE 32 x 32 x 32 MFMA tile, split-fp16 like every other conv of the net
// (a_lo w_hi + a_hi w_lo + a_hi w_hi; weights pre-scaled by 2^8, activations by 2^3: fp32-equivalent).  A lane (pixel p = lane & 31,
// half h = lane >> 5) reads its pixel's channels 16 h ... 16 h + 15 straight from the NHWC tensor (64 contiguous bytes), normalises /
// activates / splits them in registers - they ARE the A fragments (MFMA k of step s, half h, j = channel 16 h + 8 s + j: the weight image
// is packed to match) - and writes the 16 results of its column to the same LDS table t[pixel][tap][co] that stage 2 gathers from.
// ------------------------------------------------------------------------------------
typedef _Float16 ec_h8 __attribute__((ext_vector_type(8)));
typedef float ec_f16v __attribute__((ext_vector_type(16)));

template <int CIMG>
__global__ __launch_bounds__(256) void end_conv2_kernel(const EdgeConvParams p) {
    constexpr int C = 32, PW = 18, PP = PW * PW, NT_ = 9 * CIMG, TP = NT_ + 1, MTILES = (PP + 31) / 32;
    __shared__ float s_t[PP * TP];
    __shared__ float s_sc[C], s_sh[C];
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 15) / 16;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < C) {      // GroupNorm coefficients of this image (as the round-1 kernel)
        const int g = tid / p.gn_cpg;
        double sm = 0.0, ss = 0.0;
        for (int j = g * p.gn_cpg; j < (g + 1) * p.gn_cpg; ++j) {
            const double* st = p.stats + ((size_t)b * C + j) * 2;
            sm += st[0]; ss += st[1];
        }
        const double N = (double)p.gn_cpg * p.H * p.W;
        const double mean = sm / N;
        double var = ss / N - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
        const float sc = p.gamma[tid] * rstd;
        s_sc[tid] = sc; s_sh[tid] = p.beta[tid] - (float)mean * sc;
    }
    // weight fragments: [k-step][hi | lo][lane][8 halfs]
    const ec_h8* wimg = reinterpret_cast<const ec_h8*>(p.w16);
    const ec_h8 wh0 = wimg[0 * 64 + lane], wl0 = wimg[1 * 64 + lane], wh1 = wimg[2 * 64 + lane], wl1 = wimg[3 * 64 + lane];
    __syncthreads();
    const int pl = lane & 31, h = lane >> 5;
    float sc[16], sh[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { sc[j] = s_sc[16 * h + j]; sh[j] = s_sh[16 * h + j]; }
    const int oy0 = ty * 16, ox0 = tx * 16;
    for (int mt = wv; mt < MTILES; mt += 4) {
        const int hp = mt * 32 + pl;
        const int gy = oy0 - 1 + hp / PW, gx = ox0 - 1 + hp % PW;
        const bool valid = hp < PP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        float a[16];
        if (valid) {
            const float4* src = reinterpret_cast<const float4*>(p.in + ((size_t)(b * p.H + gy) * p.W + gx) * C + 16 * h);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) { const float4 v = src[q4]; a[4 * q4] = v.x; a[4 * q4 + 1] = v.y; a[4 * q4 + 2] = v.z; a[4 * q4 + 3] = v.w; }
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float u = a[j] * sc[j] + sh[j]; a[j] = u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)) * 8.0f; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = 0.f;
        }
        ec_h8 ah0, al0, ah1, al1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ah0[j] = (_Float16)a[j]; al0[j] = (_Float16)(a[j] - (float)ah0[j]);
            ah1[j] = (_Float16)a[8 + j]; al1[j] = (_Float16)(a[8 + j] - (float)ah1[j]);
        }
        ec_f16v acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, wh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wl0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, wh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wl1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wh1, acc, 0, 0, 0);
        // D: register i of lane (column n = lane & 31, half h) is row 8 (i / 4) + 4 h + (i % 4) of the tile
        if (pl < NT_) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                if (row < PP) s_t[row * TP + pl] = acc[i] * (1.0f / 2048.0f);
            }
        }
    }
    __syncthreads();
    const int ly = tid / 16, lx = tid % 16;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float o[CIMG];
#pragma unroll
    for (int co = 0; co < CIMG; ++co) o[co] = p.bias[co];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* tp = s_t + ((ly + tap / 3) * PW + lx + tap % 3) * TP + tap * CIMG;
#pragma unroll
        for (int co = 0; co < CIMG; ++co) o[co] += tp[co];
    }
    if (oy < p.H && ox < p.W) {
#pragma unroll
        for (int co = 0; co < CIMG; ++co) p.out[((size_t)(b * CIMG + co) * p.H + oy) * p.W + ox] = o[co];
    }
}

hipError_t launch_end_conv(const EdgeConvParams& p, hipStream_t s) {
    if (p.C != 32 || (p.Cimg != 1 && p.Cimg != 3)) return hipErrorInvalidValue;
    dim3 grid(p.B * ((p.H + 15) / 16) * ((p.W + 15) / 16));
    if (p.w16 != nullptr && p.stats != nullptr) {
        if (p.Cimg == 3) hipLaunchKernelGGL(end_conv2_kernel<3>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(end_conv2_kernel<1>, grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (p.Cimg == 3) hipLaunchKernelGGL((end_conv_kernel<32, 3>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((end_conv_kernel<32, 1>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Time embedding + every ResidualBlock's temb_proj in one launch
// (models.py:253-299 and :101): out[b][j] = bp[j] + Wp[j] . silu(MLP(sinusoidal(t[b]))).
// bp already contains conv1.bias so the conv epilogue adds a single vector.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temb_kernel(const TembParams p) {
    __shared__ float s_e[64];
    __shared__ float s_h[512];
    __shared__ float s_s[512];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ch = p.ch, tch = 4 * p.ch, half = ch / 2;
    if (tid < half) {
        // freq_i = exp(-i * ln(1e4)/(half-1))  (models.py:270-273), t used raw
        const float f = expf((float)tid * -(logf(10000.0f) / (float)(half - 1)));
        const float a = p.t[b] * f;
        s_e[tid] = sinf(a);
        s_e[half + tid] = cosf(a);
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b0[j];
        for (int k = 0; k < ch; ++k) acc = fmaf(p.w0[j * ch + k], s_e[k], acc);
        s_h[j] = silu_acc(acc);
    }
    __syncthreads();
    for (int j = tid; j < tch; j += 256) {
        float acc = p.b1[j];
        for (int k = 0; k < tch; ++k) acc = fmaf(p.w1[j * tch + k], s_h[k], acc);
        s_s[j] = silu_acc(acc);   // every consumer applies act(temb) first (models.py:101)
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j < p.total_out) {
        const float4* w = reinterpret_cast<const float4*>(p.wp + (size_t)j * tch);
        float acc = p.bp[j];
        for (int k = 0; k < tch / 4; ++k) {
            const float4 wv = w[k];
            acc = fmaf(wv.x, s_s[4 * k], acc); acc = fmaf(wv.y, s_s[4 * k + 1], acc);
            acc = fmaf(wv.z, s_s[4 * k + 2], acc); acc = fmaf(wv.w, s_s[4 * k + 3], acc);
        }
        p.out[(size_t)b * p.total_out + j] = acc;
    }
}

hipError_t launch_temb(const TembParams& p, hipStream_t s) {
    if (p.ch > 64 || 4 * p.ch > 512 || (p.ch & 1)) return hipErrorInvalidValue;
    dim3 grid((p.total_out + 255) / 256, p.B);
    hipLaunchKernelGGL(temb_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// softmax over the last dim, in place; one wave per row (models.py:154-155).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* data, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = data + row * cols;
    float m = -INFINITY;
    for (int i = lane; i < cols; i += 64) m = fmaxf(m, r[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) {
        const float e = expf(r[i] - m);
        r[i] = e;
        s += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < cols; i += 64) r[i] *= inv;
}

hipError_t launch_softmax_rows(float* data, int64_t rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, data, rows, cols);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// GroupNorm coefficients + operand scale of ONE conv launch, finalised once per image (models.py:33-38: GroupNorm(32 groups,
// eps 1e-6) over the - possibly concatenated - input of the conv).  Round 1 redid this fp64 finalisation in the prologue of
// every workgroup of the consuming launch (10 240 workgroups per 128^2-level launch each re-deriving the same 160 images'
// coefficients: the "15 % prologue" of profiles/r02_phase_trace_baseline.txt); now it is one micro-launch of B blocks and the
// conv prologue is two coalesced loads.
//   coef  sc[c] = gamma[c]*rstd_g, sh[c] = beta[c] - mean_g*sc[c]  (the statistics of a group may straddle two source tensors)
//   scale one power of two per K-segment, s * rms(staged operands of the segment) in [1/4, 64]: rms of a raw segment from its
//         (sum x^2) statistics, of the normalised ones from gamma/beta; s = 1 whenever the operands already are in that window
//         (every BASELINE net with weights of ordinary magnitude), so the guard costs one exact multiply and changes no result
//         there.  The conv kernel keeps its accumulator in the units of the segment it is walking (exact power-of-two rescale
//         at a segment switch) and undoes the last segment's scale in the epilogue.  Layout: scale[8b + si] = s, [8b + 4 + si] = 1/s
//   flags bit 0 when a statistic is not finite: an upstream activation overflowed (or was NaN) - reported by
//         pf_engine_check_numerics / at the end of the solver loops instead of propagating silently
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_coef_kernel(const GnCoefParams p) {
    __shared__ double s_st[2 * 1024];
    __shared__ double s_red[16];
    __shared__ int s_bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    bool bad = false;
    // (sum, sumsq) of every normalised channel, in GroupNorm channel order
    for (int c = tid; c < p.gn_C; c += 256) {
        double a = 0.0, q = 0.0;
        for (int si = 0; si < p.nseg; ++si)
            if (p.xform[si] != 0 && c >= p.gn_off[si] && c < p.gn_off[si] + p.C[si]) {
                const double* st = p.st[si] + ((size_t)b * p.C[si] + (c - p.gn_off[si])) * 2;
                a = st[0]; q = st[1];
            }
        bad |= !(isfinite(a) && isfinite(q));
        s_st[2 * c] = a; s_st[2 * c + 1] = q;
    }
    // mean square of each raw segment that carries statistics
    double raw_ss[3] = {0.0, 0.0, 0.0};
    for (int si = 0; si < p.nseg; ++si)
        if (p.xform[si] == 0 && p.st[si] != nullptr)
            for (int c = tid; c < p.C[si]; c += 256) {
                const double q = p.st[si][((size_t)b * p.C[si] + c) * 2 + 1];
                bad |= !isfinite(q);
                raw_ss[si] += q;
            }
    __syncthreads();
    double gn_ms = 0.0;
    const double inv_n = 1.0 / ((double)p.gn_cpg * (double)p.HW);
    for (int c = tid; c < p.gn_C; c += 256) {
        const int g0 = (c / p.gn_cpg) * p.gn_cpg;
        double sm = 0.0, ss = 0.0;
        for (int j = g0; j < g0 + p.gn_cpg; ++j) { sm += s_st[2 * j]; ss += s_st[2 * j + 1]; }
        const double mean = sm * inv_n;
        double var = ss * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float ga = p.gamma[c], be = p.beta[c];
        const float sc = ga * rstd;
        p.coef[((size_t)b * 2 + 0) * p.coef_stride + c] = sc;
        p.coef[((size_t)b * 2 + 1) * p.coef_stride + c] = be - (float)mean * sc;
        if (p.mu_out != nullptr) { p.mu_out[(size_t)b * p.gn_C + c] = (float)mean; p.rs_out[(size_t)b * p.gn_C + c] = rstd; }
        gn_ms += (double)ga * ga + (double)be * be;          // E[(gamma*xhat + beta)^2] with E[xhat] = 0, E[xhat^2] = 1 per group
    }
    if (bad) atomicOr(&s_bad, 1);
    if (p.scale != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gn_ms += __shfl_xor(gn_ms, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) raw_ss[k] += __shfl_xor(raw_ss[k], o);
        }
        if ((tid & 63) == 0) { s_red[(tid >> 6) * 4] = gn_ms; for (int k = 0; k < 3; ++k) s_red[(tid >> 6) * 4 + 1 + k] = raw_ss[k]; }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_bad && p.flags != nullptr) { atomicOr(p.flags, 1u); atomicCAS(p.flags + 1, 0u, (unsigned)(p.id + 1)); }
        if (p.scale != nullptr) {
            auto pow2_for = [](double ms) {            // power of two s with s*rms in [0.5, 1) when rms is outside [1/4, 64]; else 1
                float s = 1.0f;
                const float rms = (float)sqrt(ms);
                if (isfinite(rms) && rms > 0.f && (rms < 0.25f || rms > 64.f)) {
                    int e; frexpf(rms, &e);
                    e = e < -40 ? -40 : (e > 40 ? 40 : e);
                    s = ldexpf(1.0f, -e);
                }
                return s;
            };
            const double gs = s_red[0] + s_red[4] + s_red[8] + s_red[12];
            const float s_gn = p.gn_C > 0 ? pow2_for(gs / (double)p.gn_C) : 1.0f;
            for (int si = 0; si < 3; ++si) {
                float sv = 1.0f;
                if (si < p.nseg) {
                    if (p.xform[si] != 0) sv = s_gn;
                    else if (p.st[si] != nullptr)
                        sv = pow2_for((s_red[1 + si] + s_red[5 + si] + s_red[9 + si] + s_red[13 + si]) / ((double)p.C[si] * (double)p.HW));
                }
                p.scale[8 * b + si] = sv; p.scale[8 * b + 4 + si] = 1.0f / sv;
            }
        }
    }
}

hipError_t launch_gn_coef(const GnCoefParams& p, int B, hipStream_t s) {
    if (p.gn_C > 1024 || (p.gn_C > 0 && (p.coef == nullptr || p.gn_cpg <= 0))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_coef_kernel, dim3(B), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace pf
