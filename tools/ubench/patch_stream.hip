// Memory skeleton of a tiled 3x3 convolution over an NHWC fp32 activation (32 channels = 128 B per pixel): what rate does the
// ACCESS PATTERN reach with no arithmetic at all?  (round 4: conv_mfma16_kernel and the structurally unrelated conv_pp_kernel take
// the same 238 / 254 us on the 128^2 x 32-channel level - whatever bounds them is common to both.)
//
// Each workgroup walks a contiguous range of tiles; per tile every thread requests its float4 share of the (TH + 2 HALO) x (TW + 2 HALO)
// pixel patch (+ the residual tile), sums the values it received for the PREVIOUS tile (so that the loads are consumed one tile late, as
// a register-prefetching conv does) and stores the TH x TW output tile.  Variants: tile shape, halo on / off, residual on / off,
// workgroups per CU, prefetch depth 1 or 2.
// build: hipcc --offload-arch=gfx950 -O3 -o patch_stream patch_stream.hip ; run: ./patch_stream [H W B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int TH, int TW, int HALO, bool RES, int DEPTH>
__global__ __launch_bounds__(256) void patch_kernel(const float* __restrict__ in, const float* __restrict__ res, float* __restrict__ out, int B, int H, int W) {
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO, NP = PH * PW, NF = (NP * 8 + 255) / 256, NO = TH * TW * 8 / 256;
    const int t = threadIdx.x, qi = t & 7, p0 = t >> 3;
    const int tiles_x = W / TW, tiles_y = H / TH, T = B * tiles_x * tiles_y;
    const int G = gridDim.x, rg = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int t0 = (int)((long)rg * T / G), t1 = (int)((long)(rg + 1) * T / G);
    float4 ra[DEPTH][NF], rr[DEPTH][NO];
    auto issue = [&](int tile, float4 (&a)[NF], float4 (&r)[NO]) {
        tile = min(tile, t1 - 1);
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int pp = min(p0 + 32 * i, NP - 1), py = pp / PW, px = pp - py * PW;
            const int gy = min(max(ty * TH - HALO + py, 0), H - 1), gx = min(max(tx * TW - HALO + px, 0), W - 1);
            a[i] = *reinterpret_cast<const float4*>(in + (((size_t)b * H + gy) * W + gx) * 32 + qi * 4);
        }
        if (RES) {
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const int pp = p0 + 32 * i, py = pp / TW, px = pp - py * TW;
                r[i] = *reinterpret_cast<const float4*>(res + (((size_t)b * H + ty * TH + py) * W + tx * TW + px) * 32 + qi * 4);
            }
        }
    };
    auto consume = [&](int tile, float4 (&a)[NF], float4 (&r)[NO]) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NF; ++i) { s.x += a[i].x; s.y += a[i].y; s.z += a[i].z; s.w += a[i].w; }
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            float4 v = s;
            if (RES) { v.x += r[i].x; v.y += r[i].y; v.z += r[i].z; v.w += r[i].w; }
            const int pp = p0 + 32 * i, py = pp / TW, px = pp - py * TW;
            *reinterpret_cast<float4*>(out + (((size_t)b * H + ty * TH + py) * W + tx * TW + px) * 32 + qi * 4) = v;
        }
    };
    if (t1 <= t0) return;
    issue(t0, ra[0], rr[0]);
    if (DEPTH == 2) issue(t0 + 1, ra[1], rr[1]);
    for (int tile = t0; tile < t1; tile += DEPTH) {
        // consume slot 0, refill it DEPTH tiles ahead; then slot 1
        float4 ca[NF], cr[NO];
#pragma unroll
        for (int i = 0; i < NF; ++i) ca[i] = ra[0][i];
#pragma unroll
        for (int i = 0; i < NO; ++i) cr[i] = rr[0][i];
        issue(tile + DEPTH, ra[0], rr[0]);
        consume(tile, ca, cr);
        if (DEPTH == 2 && tile + 1 < t1) {
#pragma unroll
            for (int i = 0; i < NF; ++i) ca[i] = ra[1][i];
#pragma unroll
            for (int i = 0; i < NO; ++i) cr[i] = rr[1][i];
            issue(tile + 1 + DEPTH, ra[1], rr[1]);
            consume(tile + 1, ca, cr);
        }
    }
}

template <int TH, int TW, int HALO, bool RES, int DEPTH>
static void run(const char* name, const float* in, const float* res, float* out, int B, int H, int W, int wg_per_cu) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((patch_kernel<TH, TW, HALO, RES, DEPTH>), dim3(grid), dim3(256), 0, 0, in, res, out, B, H, W);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((patch_kernel<TH, TW, HALO, RES, DEPTH>), dim3(grid), dim3(256), 0, 0, in, res, out, B, H, W);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)B * H * W * 128.0 * (RES ? 3 : 2);
    printf("%-44s wg/CU %d  %8.1f us   algorithmic %6.0f GB/s\n", name, wg_per_cu, us, bytes / (us * 1e-6) / 1e9);
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 128, W = argc > 2 ? atoi(argv[2]) : 128, B = argc > 3 ? atoi(argv[3]) : 160;
    const size_t n = (size_t)B * H * W * 32;
    float *in, *res, *out;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&res, n * 4); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(in, 0, n * 4); (void)hipMemset(res, 0, n * 4); (void)hipMemset(out, 0, n * 4);
    printf("tensor %d x %d x %d x 32 fp32 = %.0f MB\n", B, H, W, n * 4 / 1e6);
    for (int wg : {2, 4, 8}) {
        run<8, 16, 1, false, 1>("8x16 tile + halo, no residual, depth 1", in, res, out, B, H, W, wg);
        run<8, 16, 1, true, 1>("8x16 tile + halo, residual, depth 1", in, res, out, B, H, W, wg);
        run<8, 16, 0, false, 1>("8x16 tile, NO halo, no residual, depth 1", in, res, out, B, H, W, wg);
        run<16, 16, 1, false, 1>("16x16 tile + halo, no residual, depth 1", in, res, out, B, H, W, wg);
        run<16, 16, 1, true, 1>("16x16 tile + halo, residual, depth 1", in, res, out, B, H, W, wg);
        run<8, 32, 1, false, 1>("8x32 tile + halo, no residual, depth 1", in, res, out, B, H, W, wg);
        run<4, 64, 1, false, 1>("4x64 tile + halo, no residual, depth 1", in, res, out, B, H, W, wg);
        run<8, 16, 1, false, 2>("8x16 tile + halo, no residual, depth 2", in, res, out, B, H, W, wg);
        run<8, 16, 1, true, 2>("8x16 tile + halo, residual, depth 2", in, res, out, B, H, W, wg);
    }
    return 0;
}
