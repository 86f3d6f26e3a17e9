// Microbenchmark: issue behaviour of v_mfma_f32_32x32x2_f32 with NACC independent accumulators per
// wave and W waves per SIMD (blocks of 256*W threads, 256 blocks = 1 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wps) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, 256 * wps>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    k<NACC><<<256, 256 * wps>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_simd = (double)iters * 8 * NACC * wps;
    const double flops = n_mfma_per_simd * 1024 * 4096.0;
    printf("NACC=%d waves/SIMD=%d : %.3f ms  %.1f TF/s  %.1f ns per MFMA per SIMD (64 cyc @2.4GHz = 26.7 ns)\n", NACC, wps, ms,
           flops / ms / 1e9, ms * 1e6 / n_mfma_per_simd);
    hipFree(out);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<1>(2); run<1>(4); run<2>(2); run<4>(2);
    return 0;
}
