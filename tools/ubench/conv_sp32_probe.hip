// Stand-alone timing of conv_pp128_kernel (pnpflow_amd/csrc/conv_sp32.hip) on synthetic tensors, with s_memtime stamps of workgroup 0's
// phases.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pnpflow_amd/csrc -o conv_sp32_probe conv_sp32_probe.hip
// run: ./conv_sp32_probe [H W B nch res first_step n_steps]
#define PP_PROBE_BUILD 1
#include "../../pnpflow_amd/csrc/conv_sp32.hip"
#include <cstdio>
#include <vector>
using namespace pf;

static int nchg = 2;
template <bool RES>
static float run(const PPParams& p0, int H, int W) {
    nchg = p0.n9;
    auto kern = conv_sp32_kernel<RES>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PPParams p = p0;
    int lx = 0; while ((16 << lx) < W) ++lx;
    int ly = 0; while ((16 << ly) < H) ++ly;
    p.lx = lx; p.ly = ly; p.rot = 5;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(256), s32_lds(nchg), 0, p);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(256), s32_lds(nchg), 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 64, W = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 160, nch = argc > 4 ? atoi(argv[4]) : 8;
    const int useres = argc > 5 ? atoi(argv[5]) : 0, s0 = argc > 6 ? atoi(argv[6]) : 4, ns = argc > 7 ? atoi(argv[7]) : 2 * nch + 2;
    const size_t n = (size_t)B * H * W * 32;
    float *in, *res, *out, *coef, *scale, *addv; double* stats; void* wimg;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&res, n * 4); (void)hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.f - 1.f;
    (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&coef, (size_t)B * 2 * 1024 * 4); std::vector<float> c((size_t)B * 2 * 1024, 0.5f); (void)hipMemcpy(coef, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&scale, (size_t)B * 8 * 4); std::vector<float> sc((size_t)B * 8, 1.0f); (void)hipMemcpy(scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&addv, (size_t)B * 128 * 4); (void)hipMemset(addv, 0, (size_t)B * 128 * 4);
    (void)hipMalloc(&stats, (size_t)B * 256 * 8); (void)hipMemset(stats, 0, (size_t)B * 256 * 8);
    (void)hipMalloc(&wimg, (size_t)73728 * 24); std::vector<_Float16> w((size_t)73728 / 2 * 24); for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(((int)(i * 40503u >> 4) % 200 - 100) * 0.01f); (void)hipMemcpy(wimg, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    PPParams p{};
    for (int i = 0; i < nch; ++i) { p.ch[i] = PPChunk{}; p.ch[i].src = in; p.ch[i].wimg = (char*)wimg + (size_t)18432 * i; p.ch[i].cstride = 32; p.ch[i].coff = 16 * (i & 1); p.ch[i].xform = 2; p.ch[i].gn_c0 = 16 * i; p.ch[i].seg = 0; }
    p.n9 = nch; p.n1 = 0; p.cout = 32; p.B = B; p.H = H; p.W = W; p.out = out; p.addvec = addv; p.addvec_bs = 32;
    p.res_scale = 1.f; p.stats_out = stats; p.out_scale = 1.f; p.coef = coef; p.coef_stride = 1024; p.scale = scale;
    p.residual = useres ? res : nullptr;
    unsigned long long* dbg; (void)hipMalloc(&dbg, 64 * 16 * 8); (void)hipMemset(dbg, 0, 64 * 16 * 8);
    auto go = [&]() -> float { return useres ? run<true>(p, H, W) : run<false>(p, H, W); };
    const float us_plain = go();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pf::g_sp32_dbg), &dbg, sizeof(dbg));      // stamps on from here
    const float us = go();
    printf("without stamps: %.1f us\n", us_plain);
    std::vector<unsigned long long> hs(64 * 16); (void)hipMemcpy(hs.data(), dbg, hs.size() * 8, hipMemcpyDeviceToHost);
    printf("%d x %d x %d x 32, %d chunks of 16, residual %d: %.1f us.  workgroup 0 wave 0, cycles per chunk: row 0 | row 1 | wait + barrier | row 2 (+ epilogue piece) | to next\n", B, H, W, nch, useres, us);
    for (int sidx = s0; sidx < s0 + ns && sidx < 63; ++sidx) {
        const unsigned long long* q = &hs[sidx * 16];
        printf("chunk %2d start %8llu: %5llu %5llu | %5llu | %5llu | %5llu\n", sidx, q[0] - hs[s0 * 16], q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], hs[(sidx + 1) * 16] - q[4]);
    }
    return 0;
}
