// Probe: how v_mfma_f32_32x32x16_f16 and the f32->f16 conversions treat fp16 subnormals (sign, flushing).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f16_subnormal mfma_f16_subnormal.hip && ./mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(const uint16_t* abits, int n, float* out_mfma, const float* xs, int nx, uint16_t* out_cvt, float* out_lo) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
        if (lane == 0) { uint16_t u = abits[t]; _Float16 h; __builtin_memcpy(&h, &u, 2); a[0] = h; }   // A[m=0][k=0]
        if (lane < 32) b[0] = (_Float16)1.0f;                                                             // B[k=0][n=lane]
        f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        if (lane == 0) out_mfma[t] = c[0];                                                                // D[0][0]
    }
    if (lane < nx) {
        const float x = xs[lane];
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        uint16_t hb, lb; __builtin_memcpy(&hb, &h, 2); __builtin_memcpy(&lb, &l, 2);
        out_cvt[2 * lane] = hb; out_cvt[2 * lane + 1] = lb; out_lo[lane] = (float)h + (float)l;
    }
}
int main() {
    const uint16_t bits[] = {0x8200, 0x0200, 0x81FF, 0x8001, 0x0001, 0x83FF, 0x8400, 0x0400};
    const int n = sizeof(bits) / 2;
    const float xs[] = {1295.5f / 16384.f, 1295.501f / 16384.f, 1295.4999f / 16384.f, 0.07907104f, -3.0517578125e-05f, -3.0458e-05f, 0.0791321f};
    const int nx = sizeof(xs) / 4;
    uint16_t* d_bits; float *d_out, *d_xs, *d_lo; uint16_t* d_cvt;
    hipMalloc(&d_bits, sizeof(bits)); hipMalloc(&d_out, n * 4); hipMalloc(&d_xs, sizeof(xs)); hipMalloc(&d_cvt, nx * 4); hipMalloc(&d_lo, nx * 4);
    hipMemcpy(d_bits, bits, sizeof(bits), hipMemcpyHostToDevice); hipMemcpy(d_xs, xs, sizeof(xs), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_bits, n, d_out, d_xs, nx, d_cvt, d_lo);
    float out[16]; uint16_t cvt[32]; float lo[16];
    hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(cvt, d_cvt, nx * 4, hipMemcpyDeviceToHost); hipMemcpy(lo, d_lo, nx * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) printf("mfma A=0x%04x * 1.0 -> %.10e\n", bits[t], out[t]);
    for (int i = 0; i < nx; ++i) printf("x=%.10e  hi=0x%04x lo=0x%04x  hi+lo=%.10e  err=%.3e\n", xs[i], cvt[2 * i], cvt[2 * i + 1], lo[i], lo[i] - xs[i]);
    return 0;
}
