// Microbenchmark: what the matrix pipe of THIS box delivers NOW under a pure v_mfma_f32_32x32x16_f16 load - the instruction every conv
// kernel of the engine multiplies with (conv_pp / conv_sp / conv_dma / conv_mfma16: three of them per fp32-equivalent product).  bench.py
// runs `mfma_f16_chain quick` beside its roofline (roofline.mfma_ceiling, measured_this_run) as it runs stream_mix for the HBM side: the
// MI355X clock under a dense MFMA load is power-managed (1.6 - 2.1 GHz in the conv kernels' traces against the 2.4 GHz of the 2.5 PFLOP/s
// datasheet peak) and differs from box to box by a few percent, so a class's executed-MFMA rate is also reported as a fraction of this
// number (a 2 % kernel gain must stay visible when the box is 3 % slower).
//   one workgroup of 256 threads per CU = ONE wave per SIMD (conv_sp's occupancy), NACC = 4 independent 32 x 32 accumulator chains per wave:
//   the pipe issues back to back (32 cycles per instruction); operands are non-trivial fp16 bit patterns (toggle rate matters for power).
//   `quick`: ~0.6 s of warm-up launches (the power manager settles), then ~1.2 s measured; one JSON line.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void chain_kernel(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // operands: pseudo-random fp16 values of magnitude 2^-6 .. 2^-5 with random signs (products ~1e-3, sums stay far inside fp32)
    f16x8 a[2], b[2];
    unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned short bits_a = (unsigned short)(((h >> 16) & 0x8000u) | 0x2400u | ((h >> 8) & 0x3ffu));
            h = h * 1664525u + 1013904223u;
            const unsigned short bits_b = (unsigned short)(((h >> 16) & 0x8000u) | 0x2400u | ((h >> 8) & 0x3ffu));
            a[v][j] = __builtin_bit_cast(_Float16, bits_a); b[v][j] = __builtin_bit_cast(_Float16, bits_b);
        }
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + i) & 1], b[(u >> 1) & 1], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

int main(int argc, char** argv) {
    const bool quick = argc >= 2 && std::string(argv[1]) == "quick";
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 2;
    constexpr int NACC = 4;
    float* out; unsigned long long* cyc;
    if (hipMalloc(&out, (size_t)cus * 256 * sizeof(float)) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 2;
    const int iters = 20000;                                  // 20 000 x 32 MFMAs x 32 cycles = 20.5 M cycles ~ 10 ms per launch
    const double flops_per_launch = (double)cus * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch_ms = [&]() {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel<NACC>, dim3(cus), dim3(256), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    const double warm_s = quick ? 0.6 : 2.0, meas_s = quick ? 1.2 : 4.0;
    double t = 0; int nwarm = 0;
    while (t < warm_s * 1e3) { t += launch_ms(); ++nwarm; }
    std::vector<double> tf; t = 0;
    unsigned long long cycles = 0;
    while (t < meas_s * 1e3) {
        const double ms = launch_ms(); t += ms;
        tf.push_back(flops_per_launch / (ms * 1e-3) / 1e12);
        (void)hipMemcpy(&cycles, cyc, 8, hipMemcpyDeviceToHost);
    }
    if (hipGetLastError() != hipSuccess || tf.empty()) return 3;
    std::sort(tf.begin(), tf.end());
    const double med = tf[tf.size() / 2], lo = tf.front(), hi = tf.back();
    // clock inferred from the issue rate: 1024 flop per cycle and SIMD when the pipe issues back to back
    const double ghz = med * 1e12 / ((double)cus * 4 * 1024.0) / 1e9;
    // cycles the shader clock counted for one launch's loop on wave 0 against the ideal 32 per instruction
    const double cyc_per_mfma = (double)cycles / ((double)iters * 8 * NACC);
    printf("{\"tflops_f16_dense\": %.1f, \"min\": %.1f, \"max\": %.1f, \"launches\": %zu, \"warmup_launches\": %d, \"cus\": %d, \"waves_per_simd\": 1, \"accumulator_chains\": %d, "
           "\"inferred_clock_ghz\": %.3f, \"counter_ticks_per_mfma\": %.2f, \"frac_of_2500\": %.4f}\n",
           med, lo, hi, tf.size(), nwarm, cus, NACC, ghz, cyc_per_mfma, med / 2500.0);
    return 0;
}
