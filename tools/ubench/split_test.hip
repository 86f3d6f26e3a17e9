// Bit-exactness of pp_common.h's asm helpers against their C++ definitions (GPU, 1 M random values incl. subnormal / huge ones):
// split4_pp vs (_Float16)x, (_Float16)(x - (float)hi); silu4_pp vs silu_pp.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pnpflow_amd/csrc -o split_test split_test.hip
#include "../../pnpflow_amd/csrc/pp_common.h"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace pf;
__global__ void k(const float4* in, int n, unsigned* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = in[i];
    unsigned h01, h23, l01, l23;
    split4_pp(v, h01, h23, l01, l23);
    float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short hr[4], lr[4];
    for (int j = 0; j < 4; ++j) {
        float xv = x[j]; asm volatile("" : "+v"(xv));
        _Float16 h = (_Float16)xv; _Float16 l = (_Float16)(xv - (float)h);
        hr[j] = __builtin_bit_cast(unsigned short, h); lr[j] = __builtin_bit_cast(unsigned short, l);
    }
    const unsigned rh01 = hr[0] | ((unsigned)hr[1] << 16), rh23 = hr[2] | ((unsigned)hr[3] << 16);
    const unsigned rl01 = lr[0] | ((unsigned)lr[1] << 16), rl23 = lr[2] | ((unsigned)lr[3] << 16);
    if (h01 != rh01 || h23 != rh23 || l01 != rl01 || l23 != rl23) atomicAdd(bad, 1u);
    float4 s = v; silu4_pp(s);
    const float r[4] = {silu_pp(v.x), silu_pp(v.y), silu_pp(v.z), silu_pp(v.w)};
    const float g[4] = {s.x, s.y, s.z, s.w};
    for (int j = 0; j < 4; ++j) if (__builtin_bit_cast(unsigned, r[j]) != __builtin_bit_cast(unsigned, g[j]) && !(r[j] != r[j] && g[j] != g[j])) atomicAdd(bad + 1, 1u);
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h((size_t)n * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const int e = (int)((s >> 8) % 60) - 40;                 // 2^-40 .. 2^19
        s = s * 1664525u + 1013904223u;
        const float m = (float)(s >> 8) / 16777216.0f * 2.0f - 1.0f;
        h[i] = ldexpf(m, e);
    }
    float4* d; unsigned* bad; (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&bad, 8); (void)hipMemset(bad, 0, 8);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n, bad);
    unsigned hb[2]; (void)hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("split4_pp mismatching float4: %u of %d; silu4_pp mismatching values: %u of %d\n", hb[0], n, hb[1], 4 * n);
    return (hb[0] || hb[1]) ? 1 : 0;
}
