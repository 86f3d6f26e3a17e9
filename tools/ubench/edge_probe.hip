// Stand-alone probe of the image-boundary convs (pnpflow_amd/csrc/unet_misc.hip): end_conv_kernel (round 1, VALU) against end_conv2_kernel (MFMA) on random
// tensors - outputs compared, launches timed - begin_conv_kernel timed, and the chip's pure float4 write / read streaming times of the same tensor.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pnpflow_amd/csrc -o edge_probe edge_probe.hip      run: ./edge_probe [H W B Cimg]
#include "../../pnpflow_amd/csrc/unet_misc.hip"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace pf;

// what the chip does for a pure streaming WRITE / READ of the same tensor (float4 per lane, grid-stride): the ceiling of begin_conv / end_conv
__global__ __launch_bounds__(256) void fill_kernel(float4* dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void drain_kernel(const float4* src, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}

template <class F> static float time_us(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    float best = 1e30f;
    for (int batch = 0; batch < 3; ++batch) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best * 1e3f / 5;
}
static float frand(unsigned& st) { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.f - 1.f; }

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 256, W = argc > 2 ? atoi(argv[2]) : 256, B = argc > 3 ? atoi(argv[3]) : 160, CI = argc > 4 ? atoi(argv[4]) : 3;
    const int C = 32; const size_t npix = (size_t)B * H * W;
    unsigned st = 12345u;
    std::vector<float> himg(npix * CI), hact(npix * C), hwb((size_t)9 * CI * C), hbias(C), hwe((size_t)9 * CI * C), hbe(4), hg(C), hbt(C);
    for (auto& v : himg) v = frand(st); for (auto& v : hact) v = 3.f * frand(st) + 0.5f;
    for (auto& v : hwb) v = 0.2f * frand(st); for (auto& v : hbias) v = 0.1f * frand(st);
    for (auto& v : hbe) v = 0.1f * frand(st); for (auto& v : hg) v = 1.f + 0.3f * frand(st); for (auto& v : hbt) v = 0.2f * frand(st);
    // end conv weights in the reference's layout [Cimg][ch][3][3], repacked as engine.hip does: VALU form [tap][co][ch], MFMA image
    std::vector<float> wend((size_t)CI * C * 9); for (auto& v : wend) v = 0.1f * frand(st);
    for (int co = 0; co < CI; ++co) for (int ci = 0; ci < C; ++ci) for (int tap = 0; tap < 9; ++tap) hwe[((size_t)tap * CI + co) * C + ci] = wend[((size_t)co * C + ci) * 9 + tap];
    std::vector<_Float16> wm((size_t)2 * 2 * 64 * 8, (_Float16)0.f);
    for (int sk = 0; sk < 2; ++sk) for (int ln = 0; ln < 64; ++ln) for (int j = 0; j < 8; ++j) {
        const int n = ln & 31, hh = ln >> 5, cc = 16 * hh + 8 * sk + j;
        if (n >= 9 * CI) continue;
        const int tap = n / CI, co = n % CI;
        const float wv = wend[((size_t)co * C + cc) * 9 + tap] * 256.0f;
        const _Float16 hi = (_Float16)wv, lo = (_Float16)(wv - (float)hi);
        wm[(((size_t)sk * 2 + 0) * 64 + ln) * 8 + j] = hi; wm[(((size_t)sk * 2 + 1) * 64 + ln) * 8 + j] = lo;
    }
    float *img, *act, *act2, *wb, *bias, *we, *be, *g, *bt, *o1, *o2; double *st1, *st2; void* w16;
    (void)hipMalloc(&img, himg.size() * 4); (void)hipMalloc(&act, hact.size() * 4); (void)hipMalloc(&act2, hact.size() * 4);
    (void)hipMalloc(&wb, hwb.size() * 4); (void)hipMalloc(&bias, 128); (void)hipMalloc(&we, hwe.size() * 4); (void)hipMalloc(&be, 16);
    (void)hipMalloc(&g, 128); (void)hipMalloc(&bt, 128); (void)hipMalloc(&o1, himg.size() * 4); (void)hipMalloc(&o2, himg.size() * 4);
    (void)hipMalloc(&st1, (size_t)B * C * 16); (void)hipMalloc(&st2, (size_t)B * C * 16); (void)hipMalloc(&w16, wm.size() * 2);
    (void)hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(wb, hwb.data(), hwb.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, hbias.data(), 128, hipMemcpyHostToDevice); (void)hipMemcpy(we, hwe.data(), hwe.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(be, hbe.data(), 16, hipMemcpyHostToDevice); (void)hipMemcpy(g, hg.data(), 128, hipMemcpyHostToDevice); (void)hipMemcpy(bt, hbt.data(), 128, hipMemcpyHostToDevice);
    (void)hipMemcpy(w16, wm.data(), wm.size() * 2, hipMemcpyHostToDevice);

    // ---- begin conv: round-1 VALU kernel against begin_conv2_kernel (MFMA) ---------------------------------------------------------------
    std::vector<_Float16> wmb((size_t)2 * 2 * 64 * 8, (_Float16)0.f);      // engine.hip's packing: k = ci * 9 + tap, weights [tap][ci][C] in hwb
    for (int sk = 0; sk < 2; ++sk) for (int ln = 0; ln < 64; ++ln) for (int j = 0; j < 8; ++j) {
        const int n = ln & 31, hh = ln >> 5, kk = 16 * sk + 8 * hh + j;
        if (kk >= 9 * CI) continue;
        const int ci = kk / 9, tap = kk % 9;
        const float wv = hwb[((size_t)tap * CI + ci) * C + n] * 256.0f;
        const _Float16 hi = (_Float16)wv, lo = (_Float16)(wv - (float)hi);
        wmb[(((size_t)sk * 2 + 0) * 64 + ln) * 8 + j] = hi; wmb[(((size_t)sk * 2 + 1) * 64 + ln) * 8 + j] = lo;
    }
    void* w16b; (void)hipMalloc(&w16b, wmb.size() * 2); (void)hipMemcpy(w16b, wmb.data(), wmb.size() * 2, hipMemcpyHostToDevice);
    EdgeConvParams p{}; p.in = img; p.w = wb; p.bias = bias; p.B = B; p.H = H; p.W = W; p.Cimg = CI; p.C = C;
    auto old_begin = [&](float* out, double* stats) { EdgeConvParams q = p; q.out = out; q.stats_out = stats; q.w16 = nullptr; (void)launch_begin_conv(q, 0); };
    auto new_begin = [&](float* out, double* stats) { EdgeConvParams q = p; q.out = out; q.stats_out = stats; q.w16 = w16b; (void)launch_begin_conv(q, 0); };
    (void)hipMemset(st1, 0, (size_t)B * C * 16); (void)hipMemset(st2, 0, (size_t)B * C * 16);
    old_begin(act, st1); new_begin(act2, st2); (void)hipDeviceSynchronize();
    {
        std::vector<float> a(npix * C), b2(npix * C); std::vector<double> s1((size_t)B * C * 2), s2((size_t)B * C * 2);
        (void)hipMemcpy(a.data(), act, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b2.data(), act2, a.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(s1.data(), st1, s1.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(s2.data(), st2, s2.size() * 8, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0; for (size_t i = 0; i < a.size(); ++i) { rmax = fmax(rmax, fabs(a[i])); const double d = fabs((double)a[i] - b2[i]); if (!(d <= emax)) emax = d; }
        double srel = 0; for (size_t i = 0; i < s1.size(); ++i) srel = fmax(srel, fabs(s1[i] - s2[i]) / fmax(1.0, fabs(s1[i])));
        printf("begin_conv %dx%dx%d Cimg %d: max|MFMA - VALU| = %.3e, max|VALU| = %.3e; statistics max rel diff %.2e  %s\n", B, H, W, CI, emax, rmax, srel, emax <= 2e-6 * rmax && srel < 1e-4 ? "OK" : "FAIL");
    }
    printf("begin_conv: round-1 kernel %.1f us, begin_conv2_kernel %.1f us (%.2f GB written)\n", time_us([&] { old_begin(act, st1); }), time_us([&] { new_begin(act2, st2); }), npix * C * 4 / 1e9);
    for (int g : {1024, 4096, 16384})
        printf("pure float4 streaming of the %.2f GB tensor, grid %5d: write %.1f us, read %.1f us\n", npix * C * 4 / 1e9, g,
               time_us([&] { hipLaunchKernelGGL(fill_kernel, dim3(g), dim3(256), 0, 0, reinterpret_cast<float4*>(act2), npix * C / 4); }),
               time_us([&] { hipLaunchKernelGGL(drain_kernel, dim3(g), dim3(256), 0, 0, reinterpret_cast<const float4*>(act2), npix * C / 4, o1); }));
    // ---- end conv (GroupNorm + SiLU form) ---------------------------------------------------------------------------------------------------
    (void)hipMemcpy(act, hact.data(), hact.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> hst((size_t)B * C * 2);
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) { hst[((size_t)b * C + c) * 2] = 0.5 * H * W; hst[((size_t)b * C + c) * 2 + 1] = (3.0 + 0.25) * H * W; }      // mean 0.5, E[x^2] = var 3 + 0.25
    (void)hipMemcpy(st1, hst.data(), hst.size() * 8, hipMemcpyHostToDevice);
    EdgeConvParams e{}; e.in = act; e.w = we; e.bias = be; e.B = B; e.H = H; e.W = W; e.Cimg = CI; e.C = C; e.stats = st1; e.gamma = g; e.beta = bt; e.gn_cpg = 1; e.gn_eps = 1e-6f;
    auto old_end = [&](float* out) { EdgeConvParams q = e; q.out = out; q.w16 = nullptr; (void)launch_end_conv(q, 0); };
    auto new_end = [&](float* out) { EdgeConvParams q = e; q.out = out; q.w16 = w16; (void)launch_end_conv(q, 0); };
    old_end(o1); new_end(o2); (void)hipDeviceSynchronize();
    {
        std::vector<float> a(npix * CI), b2(npix * CI);
        (void)hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b2.data(), o2, a.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0; for (size_t i = 0; i < a.size(); ++i) { rmax = fmax(rmax, fabs(a[i])); const double d = fabs((double)a[i] - b2[i]); if (!(d <= emax)) emax = d; }
        printf("end_conv: max|MFMA - VALU| = %.3e, max|VALU| = %.3e  %s\n", emax, rmax, emax <= 2e-6 * rmax ? "OK" : "FAIL");
    }
    printf("end_conv: round-1 kernel %.1f us, end_conv2_kernel %.1f us (%.2f GB read)\n", time_us([&] { old_end(o1); }), time_us([&] { new_end(o2); }), npix * C * 4 / 1e9);
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    return 0;
}
