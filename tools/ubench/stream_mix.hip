// HBM streaming ceilings for the access mixes of the restoration path (round 3): float4 per lane, grid-stride, tensors of 64 MB ...
// 1.34 GB (the 256^2 x 32-channel activation at U-Net batch 160 is 1.34 GB: far beyond the 256 MiB Infinity Cache).
//   1R+1W  copy                        (what MI355X_MICROARCH.md quotes: 6.3 TB/s)
//   2R+1W  the conv of a residual-free block / a grad step
//   3R+1W  conv + residual / GroupNorm-backward second stage
//   3R+1W-inplace  (reads a, b, out; writes out): gn_bwd_post<false, true>
// build: hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip ; run: ./stream_mix
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

template <int NR, bool INPLACE>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = a[i];
        if (NR >= 2) { const float4 w = b[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (NR >= 3) { const float4 w = INPLACE ? out[i] : c[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        out[i] = v;
    }
}

// the same mixes with the work split the way gn_bwd_post splits it: every workgroup owns one contiguous chunk (32 KB of each tensor)
template <int NR, bool INPLACE>
__global__ __launch_bounds__(256) void mix_chunk_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float4* out, size_t n4, int chunk4) {
    const size_t base = (size_t)blockIdx.x * chunk4;
    for (int j = threadIdx.x; j < chunk4; j += 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = min(base + j + k * 256, n4 - 1);
            v[k] = a[i];
            if (NR >= 2) { const float4 w = b[i]; v[k].x += w.x; v[k].y += w.y; v[k].z += w.z; v[k].w += w.w; }
            if (NR >= 3) { const float4 w = INPLACE ? out[i] : c[i]; v[k].x += w.x; v[k].y += w.y; v[k].z += w.z; v[k].w += w.w; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const size_t i = base + j + k * 256; if (i < n4) out[i] = v[k]; }
    }
}
template <int NR, bool INPLACE>
static double run_chunk(const float4* a, const float4* b, const float4* c, float4* o, size_t n4, int chunk4) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (int)((n4 + chunk4 - 1) / chunk4);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix_chunk_kernel<NR, INPLACE>), dim3(grid), dim3(256), 0, 0, a, b, c, o, n4, chunk4);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((mix_chunk_kernel<NR, INPLACE>), dim3(grid), dim3(256), 0, 0, a, b, c, o, n4, chunk4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)(NR + 1) * n4 * 16 * reps / (ms * 1e-3) / 1e9;
}

template <int NR, bool INPLACE>
static double run(const float4* a, const float4* b, const float4* c, float4* o, size_t n4, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix_kernel<NR, INPLACE>), dim3(grid), dim3(256), 0, 0, a, b, c, o, n4);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((mix_kernel<NR, INPLACE>), dim3(grid), dim3(256), 0, 0, a, b, c, o, n4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)(NR + 1) * n4 * 16 * reps / (ms * 1e-3) / 1e9;
}

// `stream_mix quick <bytes>`: one JSON line for ONE tensor size (what bench.py quotes as roofline.streaming_ceiling_gbs, measured in the
// run that reports it): the best of two grid-stride grids and the 16 KB-per-workgroup split, per read / write mix
static int quick(size_t bytes) {
    const size_t n4 = bytes / 16;
    float4 *a, *b, *c, *o;
    if (hipMalloc(&a, n4 * 16) != hipSuccess || hipMalloc(&b, n4 * 16) != hipSuccess || hipMalloc(&c, n4 * 16) != hipSuccess || hipMalloc(&o, n4 * 16) != hipSuccess) return 2;
    (void)hipMemset(a, 0, n4 * 16); (void)hipMemset(b, 0, n4 * 16); (void)hipMemset(c, 0, n4 * 16); (void)hipMemset(o, 0, n4 * 16);
    double best[3] = {0, 0, 0};
    for (int grid : {8192, 32768}) {
        best[0] = fmax(best[0], run<1, false>(a, b, c, o, n4, grid)); best[1] = fmax(best[1], run<2, false>(a, b, c, o, n4, grid)); best[2] = fmax(best[2], run<3, false>(a, b, c, o, n4, grid));
    }
    const int c4 = 16 * 1024 / 16;
    best[0] = fmax(best[0], run_chunk<1, false>(a, b, c, o, n4, c4)); best[1] = fmax(best[1], run_chunk<2, false>(a, b, c, o, n4, c4)); best[2] = fmax(best[2], run_chunk<3, false>(a, b, c, o, n4, c4));
    printf("{\"tensor_bytes\": %zu, \"1R+1W\": %.0f, \"2R+1W\": %.0f, \"3R+1W\": %.0f}\n", bytes, best[0], best[1], best[2]);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "quick") return quick((size_t)atoll(argv[2]));
    const size_t sizes_mb[] = {64, 268, 1342};
    for (size_t mb : sizes_mb) {
        const size_t n4 = mb * 1000 * 1000 / 16;
        float4 *a, *b, *c, *o;
        (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16); (void)hipMalloc(&c, n4 * 16); (void)hipMalloc(&o, n4 * 16);
        (void)hipMemset(a, 0, n4 * 16); (void)hipMemset(b, 0, n4 * 16); (void)hipMemset(c, 0, n4 * 16); (void)hipMemset(o, 0, n4 * 16);
        for (int grid : {2048, 8192, 32768}) {
            printf("tensor %5zu MB  grid %6d   1R+1W %7.0f   2R+1W %7.0f   3R+1W %7.0f   3R+1W in place %7.0f   GB/s\n", mb, grid,
                   run<1, false>(a, b, c, o, n4, grid), run<2, false>(a, b, c, o, n4, grid), run<3, false>(a, b, c, o, n4, grid), run<3, true>(a, b, c, o, n4, grid));
        }
        for (int chunk_kb : {16, 32, 128, 1024}) {
            const int c4 = chunk_kb * 1024 / 16;
            printf("tensor %5zu MB  chunk %4d KB/workgroup   1R+1W %7.0f   2R+1W %7.0f   3R+1W %7.0f   3R+1W in place %7.0f   GB/s\n", mb, chunk_kb,
                   run_chunk<1, false>(a, b, c, o, n4, c4), run_chunk<2, false>(a, b, c, o, n4, c4), run_chunk<3, false>(a, b, c, o, n4, c4), run_chunk<3, true>(a, b, c, o, n4, c4));
        }
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(o);
    }
    return 0;
}
