// LPIPS (AlexNet, v0.1) as the reference computes it at its logging iterations (pnpflow/utils.py:677-724:
// lpips.LPIPS(net='alex')(clean, rec, normalize=True).mean()).
//
// Third-party algorithm: the `lpips` package (requirements.txt, unpinned; v0.1.4 is current) on torchvision's AlexNet features.
// Restated from their published definition:
//   in' = 2 in - 1 when normalize                      (lpips/lpips.py LPIPS.forward; the reference passes images that already are
//                                                       in [-1, 1] AND normalize=True - kept, utils.py:703-708)
//   x   = (in' - shift) / scale, shift = (-.030, -.088, -.188), scale = (.458, .448, .450)           (ScalingLayer)
//   f1 = relu(conv 3->64, 11x11, stride 4, pad 2)(x);  f2 = relu(conv 64->192, 5x5, pad 2)(maxpool3s2(f1));
//   f3 = relu(conv 192->384, 3x3, pad 1)(maxpool3s2(f2));  f4 = relu(conv 384->256, 3x3, pad 1)(f3);  f5 = relu(conv 256->256, 3x3, pad 1)(f4)
//   d  = sum_k  mean_{h,w}  sum_c  w_k[c] * ( f_k0[c] / (|f_k0| + 1e-10) - f_k1[c] / (|f_k1| + 1e-10) )^2       (unit-normalised over c)
// PARITY UNPINNED: neither lpips nor torchvision (nor their weight files) are in the image; the tests compare with the oracle's
// torch restatement on synthetic weights loaded under the published key names.
//
// A metric of the logging iterations, not of the restoration loop: plain fp32 direct convolutions (one thread = one output pixel x 8
// output channels, weights through wave-uniform scalar loads), ~0.2 GFLOP per 128^2 image and side.
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/pnpflow_hip.h"
#include "pf_common.h"

namespace pf {

constexpr int LP_CO = 8;        // output channels per thread

__global__ __launch_bounds__(256) void lp_scale_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int normalize, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / HW) % 3);
    const float shift = c == 0 ? -0.030f : (c == 1 ? -0.088f : -0.188f), scale = c == 0 ? 0.458f : (c == 1 ? 0.448f : 0.450f);
    float v = in[i];
    if (normalize) v = 2.0f * v - 1.0f;
    out[i] = (v - shift) / scale;
}

// out[b][co][oy][ox] = relu(bias[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] * in[b][ci][oy*S - P + ky][ox*S - P + kx])
__global__ __launch_bounds__(256) void lp_conv_relu_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Ci, int Hi, int Wi, int Co, int Ho, int Wo, int K, int S, int P) {
    const int cog = blockIdx.y, b = blockIdx.z;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Ho * Wo) return;
    const int oy = pix / Wo, ox = pix - oy * Wo;
    const int co0 = cog * LP_CO;
    float acc[LP_CO];
#pragma unroll
    for (int j = 0; j < LP_CO; ++j) acc[j] = bias[min(co0 + j, Co - 1)];
    const float* ib = in + (size_t)b * Ci * Hi * Wi;
    const int KK = K * K;
    for (int ci = 0; ci < Ci; ++ci) {
        const float* ip = ib + (size_t)ci * Hi * Wi;
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * S - P + ky;
            if (iy < 0 || iy >= Hi) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * S - P + kx;
                if (ix < 0 || ix >= Wi) continue;
                const float v = ip[iy * Wi + ix];
                const float* wp = w + ((size_t)co0 * Ci + ci) * KK + ky * K + kx;      // wave-uniform: scalar loads
#pragma unroll
                for (int j = 0; j < LP_CO; ++j) acc[j] = fmaf(wp[(size_t)min(j, Co - 1 - co0) * Ci * KK], v, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LP_CO; ++j)
        if (co0 + j < Co) out[(((size_t)b * Co + co0 + j) * Ho + oy) * Wo + ox] = fmaxf(acc[j], 0.0f);
}

// MaxPool2d(kernel 3, stride 2), no padding, floor mode
__global__ __launch_bounds__(256) void lp_maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int Hi, int Wi, int Ho, int Wo, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const size_t plane = i / ((size_t)Wo * Ho);
    const float* ip = in + plane * Hi * Wi + (size_t)(oy * 2) * Wi + ox * 2;
    float m = ip[0];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, ip[dy * Wi + dx]);
    out[i] = m;
}

// part[b * nparts + part_off + block] = (1 / HW) * sum over the block's pixels of sum_c lin[c] * (f0 / (|f0| + eps) - f1 / (|f1| + eps))^2
// (one thread per pixel, block-reduced).  The per-block partials of the five layers are summed in a FIXED order by lp_sum_kernel: the
// value of an image is bit-reproducible between runs and between 1-rank and N-rank jobs (a float atomicAdd across blocks is not).
__global__ __launch_bounds__(256) void lp_head_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ lin,
                                                     float* __restrict__ part, int nparts, int part_off, int C, int HW) {
    __shared__ float s_red[4];
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    float val = 0.0f;
    if (pix < HW) {
        const float* a = f0 + (size_t)b * C * HW + pix; const float* c = f1 + (size_t)b * C * HW + pix;
        float n0 = 0.f, n1 = 0.f;
        for (int k = 0; k < C; ++k) { const float u = a[(size_t)k * HW], v = c[(size_t)k * HW]; n0 = fmaf(u, u, n0); n1 = fmaf(v, v, n1); }
        const float r0 = 1.0f / (sqrtf(n0) + 1e-10f), r1 = 1.0f / (sqrtf(n1) + 1e-10f);
        for (int k = 0; k < C; ++k) { const float d = a[(size_t)k * HW] * r0 - c[(size_t)k * HW] * r1; val = fmaf(lin[k] * d, d, val); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)b * nparts + part_off + blockIdx.x] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)HW;
}

__global__ void lp_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc = 0.0f;
    for (int i = 0; i < nparts; ++i) acc += part[(size_t)b * nparts + i];
    out[b] = acc;
}

}  // namespace pf

using namespace pf;

struct pf_lpips {
    int device = 0;
    std::string err;
    std::map<std::string, float*> w;                 // device arrays by canonical name
    std::map<std::string, std::vector<int64_t>> shape;
    std::vector<void*> allocs;
    float* buf[2][7] = {{nullptr}};                  // per side: scaled input, f1, p1, f2, p2, f3..f5 share by ping-pong (see forward)
    size_t cap = 0;                                  // floats per activation buffer
    float* part = nullptr; size_t part_cap = 0;      // per-block partial sums of the five heads
};

namespace {
struct Layer { const char* name; int Ci, Co, K, S, P; };
const Layer LAYERS[5] = {{"features.0", 3, 64, 11, 4, 2}, {"features.3", 64, 192, 5, 1, 2}, {"features.6", 192, 384, 3, 1, 1},
                         {"features.8", 384, 256, 3, 1, 1}, {"features.10", 256, 256, 3, 1, 1}};
struct Guard { int prev = -1; explicit Guard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; hipSetDevice(d); } ~Guard() { if (prev >= 0) hipSetDevice(prev); } };
}

int pf_lpips_create(int device_id, pf_lpips** out) {
    if (!out) return PF_ERR_INVALID;
    auto* l = new pf_lpips(); l->device = device_id;
    *out = l;
    return PF_OK;
}

void pf_lpips_destroy(pf_lpips* l) {
    if (!l) return;
    Guard g(l->device);
    for (void* p : l->allocs) hipFree(p);
    delete l;
}

const char* pf_lpips_last_error(const pf_lpips* l) { return l ? l->err.c_str() : ""; }

// names: "features.{0,3,6,8,10}.{weight,bias}" (torchvision AlexNet) and "lin{0..4}" ([C] / [1][C][1][1] of lpips' NetLinLayer conv)
int pf_lpips_load_weight(pf_lpips* l, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!l || !name || !host || ndim < 1 || ndim > 4) return PF_ERR_INVALID;
    Guard g(l->device);
    std::vector<int64_t> want;
    const std::string n(name);
    for (int k = 0; k < 5; ++k) {
        const Layer& L = LAYERS[k];
        if (n == std::string(L.name) + ".weight") want = {L.Co, L.Ci, L.K, L.K};
        if (n == std::string(L.name) + ".bias") want = {L.Co};
        if (n == "lin" + std::to_string(k)) want = {L.Co};
    }
    if (want.empty()) { l->err = "unknown LPIPS weight " + n; return PF_ERR_INVALID; }
    int64_t cnt = 1, wcnt = 1;
    for (int i = 0; i < ndim; ++i) cnt *= shape[i];
    for (auto v : want) wcnt *= v;
    if (cnt != wcnt) { l->err = "shape mismatch for " + n; return PF_ERR_INVALID; }
    float* d = nullptr;
    if (hipMalloc(&d, (size_t)cnt * sizeof(float)) != hipSuccess) { l->err = "hipMalloc failed"; return PF_ERR_HIP; }
    if (hipMemcpy(d, host, (size_t)cnt * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { l->err = "hipMemcpy failed"; return PF_ERR_HIP; }
    l->allocs.push_back(d); l->w[n] = d; l->shape[n] = want;
    return PF_OK;
}

int pf_lpips_forward(pf_lpips* l, const float* img0, const float* img1, float* out, int B, int H, int W, int normalize, void* stream) {
    if (!l || !img0 || !img1 || !out || B <= 0 || H < 32 || W < 32) return PF_ERR_INVALID;
    Guard g(l->device);
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < 5; ++k)
        for (const char* suf : {".weight", ".bias"})
            if (!l->w.count(std::string(LAYERS[k].name) + suf)) { l->err = std::string("LPIPS weight not loaded: ") + LAYERS[k].name + suf; return PF_ERR_STATE; }
    for (int k = 0; k < 5; ++k) if (!l->w.count("lin" + std::to_string(k))) { l->err = "LPIPS weight not loaded: lin" + std::to_string(k); return PF_ERR_STATE; }
    // spatial sizes
    int h[8], w[8];       // 0 input, 1 f1, 2 p1, 3 f2, 4 p2, 5..7 f3..f5
    h[0] = H; w[0] = W;
    h[1] = (H + 4 - 11) / 4 + 1; w[1] = (W + 4 - 11) / 4 + 1;
    h[2] = (h[1] - 3) / 2 + 1; w[2] = (w[1] - 3) / 2 + 1;
    h[3] = h[2]; w[3] = w[2];
    h[4] = (h[3] - 3) / 2 + 1; w[4] = (w[3] - 3) / 2 + 1;
    h[5] = h[6] = h[7] = h[4]; w[5] = w[6] = w[7] = w[4];
    if (h[4] < 1 || w[4] < 1) return PF_ERR_INVALID;
    const int ch[8] = {3, 64, 64, 192, 192, 384, 256, 256};
    size_t need = 0;
    for (int i = 0; i < 8; ++i) need = std::max(need, (size_t)B * ch[i] * h[i] * w[i]);
    auto drop = [&](float*& p) {                     // free + forget: a failed regrow must leave nothing stale behind (no double free later)
        if (!p) return;
        hipFree(p);
        auto it = std::find(l->allocs.begin(), l->allocs.end(), (void*)p);
        if (it != l->allocs.end()) l->allocs.erase(it);
        p = nullptr;
    };
    if (need > l->cap) {
        hipStreamSynchronize(s);
        l->cap = 0;
        for (int side = 0; side < 2; ++side)
            for (int i = 0; i < 7; ++i) drop(l->buf[side][i]);
        for (int side = 0; side < 2; ++side)
            for (int i = 0; i < 7; ++i) {
                if (hipMalloc(&l->buf[side][i], need * sizeof(float)) != hipSuccess) { l->buf[side][i] = nullptr; l->err = "hipMalloc failed (activations)"; return PF_ERR_HIP; }
                l->allocs.push_back(l->buf[side][i]);
            }
        l->cap = need;
    }
    auto conv = [&](int k, const float* in, float* o, int hi, int wi, int ho, int wo) {
        const Layer& L = LAYERS[k];
        dim3 grid((ho * wo + 255) / 256, (L.Co + LP_CO - 1) / LP_CO, B);
        hipLaunchKernelGGL(lp_conv_relu_kernel, grid, dim3(256), 0, s, in, l->w.at(std::string(L.name) + ".weight"), l->w.at(std::string(L.name) + ".bias"), o,
                           L.Ci, hi, wi, L.Co, ho, wo, L.K, L.S, L.P);
    };
    auto pool = [&](const float* in, float* o, int C, int hi, int wi, int ho, int wo) {
        const size_t n = (size_t)B * C * ho * wo;
        hipLaunchKernelGGL(lp_maxpool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, o, hi, wi, ho, wo, n);
    };
    // features of both sides: buf[side][0] scaled input, [1] f1, [2] pool, [3] f2, [4] pool, [5] f3, [6] f4, [0] f5 (the input is dead by then)
    float* f[2][5];
    for (int side = 0; side < 2; ++side) {
        float** bf = l->buf[side];
        const size_t n0 = (size_t)B * 3 * H * W;
        hipLaunchKernelGGL(lp_scale_kernel, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, side == 0 ? img0 : img1, bf[0], H * W, normalize, n0);
        conv(0, bf[0], bf[1], h[0], w[0], h[1], w[1]);
        pool(bf[1], bf[2], 64, h[1], w[1], h[2], w[2]);
        conv(1, bf[2], bf[3], h[2], w[2], h[3], w[3]);
        pool(bf[3], bf[4], 192, h[3], w[3], h[4], w[4]);
        conv(2, bf[4], bf[5], h[4], w[4], h[5], w[5]);
        conv(3, bf[5], bf[6], h[5], w[5], h[6], w[6]);
        conv(4, bf[6], bf[0], h[6], w[6], h[7], w[7]);
        f[side][0] = bf[1]; f[side][1] = bf[3]; f[side][2] = bf[5]; f[side][3] = bf[6]; f[side][4] = bf[0];
    }
    const int fh[5] = {h[1], h[3], h[5], h[6], h[7]}, fw[5] = {w[1], w[3], w[5], w[6], w[7]};
    int nparts = 0, poff[5];
    for (int k = 0; k < 5; ++k) { poff[k] = nparts; nparts += (fh[k] * fw[k] + 255) / 256; }
    if ((size_t)B * nparts > l->part_cap) {
        hipStreamSynchronize(s);
        l->part_cap = 0;
        drop(l->part);
        if (hipMalloc(&l->part, (size_t)B * nparts * sizeof(float)) != hipSuccess) { l->part = nullptr; l->err = "hipMalloc failed (partials)"; return PF_ERR_HIP; }
        l->allocs.push_back(l->part);
        l->part_cap = (size_t)B * nparts;
    }
    for (int k = 0; k < 5; ++k) {
        const int HW = fh[k] * fw[k];
        hipLaunchKernelGGL(lp_head_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, f[0][k], f[1][k], l->w.at("lin" + std::to_string(k)), l->part, nparts, poff[k], LAYERS[k].Co, HW);
    }
    hipLaunchKernelGGL(lp_sum_kernel, dim3((B + 63) / 64), dim3(64), 0, s, l->part, out, nparts, B);
    if (hipGetLastError() != hipSuccess) { l->err = "LPIPS kernel launch failed"; return PF_ERR_HIP; }
    return PF_OK;
}
