// Stand-alone timing of conv_pp_kernel (pnpflow_amd/csrc/conv_pp.hip) on synthetic tensors, with s_memtime stamps of workgroup 0's phases
// (compiled in by PP_PROBE_BUILD; the library has none).  The one-ingredient-removed timings of profiles/r04_level0_probes.md came from a
// PROBE template parameter that existed during round 4 (commit c2787d4) and was removed from the kernel afterwards.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pnpflow_amd/csrc -o conv_pp_probe conv_pp_probe.hip ; run: ./conv_pp_probe [H W B nch]
#define PP_PROBE_BUILD 1
#include "../../pnpflow_amd/csrc/conv_pp.hip"
#ifndef PP_TEAM_BARRIERS      // builds with docs/experiments/r06_conv_pp_team_barriers.patch applied define it (the patched kernel needs 128 B of LDS for its team counters)
constexpr int PP_TBAR_BYTES = 0;
#endif
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>
using namespace pf;

template <int N9, bool RES, int TEAMS = 2>
static float run(const PPParams& p0, int H, int W) {
    auto kern = conv_pp_kernel<1, N9, 0, RES, TEAMS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PPParams p = p0;
    int lx = 0; while ((16 << lx) < W) ++lx;
    int ly = 0; while ((8 << ly) < H) ++ly;
    p.lx = lx; p.ly = ly; p.rot = getenv("ROT") ? atoi(getenv("ROT")) : 5;
    const size_t lds = (size_t)N9 * 36864 + TEAMS * pp_patch_bytes(1) + (TEAMS == 2 ? PP_TBAR_BYTES : 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(TEAMS == 2 ? 256 : 512), dim3(256 * TEAMS), lds, 0, p);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(TEAMS == 2 ? 256 : 512), dim3(256 * TEAMS), lds, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 128, W = argc > 2 ? atoi(argv[2]) : 128, B = argc > 3 ? atoi(argv[3]) : 160, nch = argc > 4 ? atoi(argv[4]) : 1;
    const size_t n = (size_t)B * H * W * 32;
    float *in, *in2, *res, *out, *coef, *scale, *addv; double* stats; void* wimg;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&in2, n * 4); (void)hipMalloc(&res, n * 4); (void)hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.f - 1.f;
    (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(in2, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&coef, (size_t)B * 2 * 1024 * 4); std::vector<float> c((size_t)B * 2 * 1024, 0.5f); (void)hipMemcpy(coef, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&scale, (size_t)B * 8 * 4); std::vector<float> sc((size_t)B * 8, 1.0f); (void)hipMemcpy(scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&addv, (size_t)B * 32 * 4); (void)hipMemset(addv, 0, (size_t)B * 32 * 4);
    (void)hipMalloc(&stats, (size_t)B * 64 * 8); (void)hipMemset(stats, 0, (size_t)B * 64 * 8);
    (void)hipMalloc(&wimg, 36864); std::vector<_Float16> w(18432); for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(((int)(i * 40503u >> 4) % 200 - 100) * 0.01f); (void)hipMemcpy(wimg, w.data(), 36864, hipMemcpyHostToDevice);
    PPParams p{};
    for (int i = 0; i < nch; ++i) { p.ch[i] = PPChunk{}; p.ch[i].src = i == 0 ? in : in2; p.ch[i].wimg = wimg; p.ch[i].cstride = 32; p.ch[i].coff = 0; p.ch[i].xform = 2; p.ch[i].gn_c0 = 32 * i; p.ch[i].seg = i; }
    p.n9 = nch; p.n1 = 0; p.B = B; p.H = H; p.W = W; p.out = out; p.addvec = addv; p.addvec_bs = 32;
    p.res_scale = 1.f; p.stats_out = stats; p.out_scale = 1.f; p.coef = coef; p.coef_stride = 1024; p.scale = scale;
    unsigned long long* dbg; (void)hipMalloc(&dbg, (2 * 64 * 8 + 64) * 8); (void)hipMemset(dbg, 0, (2 * 64 * 8 + 64) * 8);
    p.residual = nullptr;
    if (nch == 1) {
        printf("%d x %d x %d x 32, 1 chunk: one 8-wave workgroup per CU %7.1f us | two 4-wave workgroups per CU %7.1f us | with residual %7.1f / %7.1f us\n", B, H, W,
               run<1, false, 2>(p, H, W), run<1, false, 1>(p, H, W), (p.residual = res, run<1, true, 2>(p, H, W)), run<1, true, 1>(p, H, W));
        p.residual = getenv("STAMPS") && atoi(getenv("STAMPS")) > 1 ? res : nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pf::g_pp_dbg), &dbg, sizeof(dbg));      // stamps on from here
        { const int skip = getenv("SKIP") ? atoi(getenv("SKIP")) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(pf::g_pp_dbg_skip), &skip, sizeof(skip)); }
        const float us = p.residual ? run<1, true, 2>(p, H, W) : run<1, false, 2>(p, H, W);
        std::vector<unsigned long long> hs(2 * 64 * 8 + 64); (void)hipMemcpy(hs.data(), dbg, hs.size() * 8, hipMemcpyDeviceToHost);
        { double tk = 0; int nw = 0; for (int w = 0; w < 8; ++w) if (hs[2 * 64 * 8 + 2 * w + 1] > hs[2 * 64 * 8 + 2 * w]) { tk += (double)(hs[2 * 64 * 8 + 2 * w + 1] - hs[2 * 64 * 8 + 2 * w]); ++nw; }
          if (nw) printf("walk of %d sampled workgroups: %.0f shader-clock ticks on average = %.3f GHz over the launch's %.1f us\n", nw, tk / nw, tk / nw / (us * 1e3), us); }
        printf("stamped (8-wave workgroup, residual %d): %.1f us.  workgroup 0, cycles per step: - | epilogue | transform | requests | barrier | mfma | barrier\n", p.residual ? 1 : 0, us);
        for (int tm = 0; tm < 2; ++tm)
            for (int sidx = 8; sidx < 16; ++sidx) {
                const unsigned long long* q = &hs[(tm * 64 + sidx) * 8];
                printf("team %d step %2d  start %8llu : %6llu | %6llu | %6llu | %6llu | %6llu | %6llu | %6llu\n", tm, sidx, q[0] - hs[(tm * 64 + 8) * 8], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[7] - q[6]);
            }
    } else {
        // multi-chunk: the 8-wave form (TEAMS = 2, what the library launches) against the one-team form of the same kernel (one 4-wave workgroup per
        // CU: same arithmetic in the same order, so the outputs must be BIT-equal) - the check of the team barriers (round 6)
        auto checksum = [&]() {
            std::vector<float> o(n); (void)hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
            unsigned long long hsh = 1469598103934665603ull; double sum = 0;
            for (size_t i = 0; i < n; ++i) { unsigned u; memcpy(&u, &o[i], 4); hsh = (hsh ^ u) * 1099511628211ull; sum += o[i]; }
            return std::make_pair(hsh, sum);
        };
        float us2 = 0, us1 = 0;
        std::pair<unsigned long long, double> c2, c1;
        // (the clock manager needs ~40 ms of load to settle: the first launches of a process run 15-20 % slower than the rest)
        for (int k = 0; k < 6; ++k) { if (nch == 2) run<2, false, 2>(p, H, W); else run<3, false, 2>(p, H, W); }
        if (nch == 2) {
            (void)hipMemset(out, 0, n * 4); us2 = run<2, false, 2>(p, H, W); c2 = checksum();
            (void)hipMemset(out, 0, n * 4); us1 = run<2, false, 1>(p, H, W); c1 = checksum();
        } else {
            (void)hipMemset(out, 0, n * 4); us2 = run<3, false, 2>(p, H, W); c2 = checksum();
            (void)hipMemset(out, 0, n * 4); us1 = run<3, false, 1>(p, H, W); c1 = checksum();
        }
        printf("%d x %d x %d x 32, %d chunks: 8-wave form %7.1f us | one 4-wave workgroup per CU %7.1f us | outputs %s (fnv %016llx / %016llx, sum %.6e)\n", B, H, W, nch, us2, us1,
               c2.first == c1.first ? "BIT-EQUAL" : "DIFFER", c2.first, c1.first, c2.second);
        // a few more launches of the 8-wave form: timing spread
        for (int k = 0; k < 4; ++k) printf("  8-wave form again: %7.1f us\n", nch == 2 ? run<2, false, 2>(p, H, W) : run<3, false, 2>(p, H, W));
    }
    return 0;
}
