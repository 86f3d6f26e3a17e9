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
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>
using namespace pf;

template <int N9, bool RES, int TEAMS = 2, int TERMS = 3>
static float run(const PPParams& p0, int H, int W) {
    auto kern = conv_pp_kernel<1, N9, 0, RES, TEAMS, TERMS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PPParams p = p0;
    int lx = 0; while ((16 << lx) < W) ++lx;
    int ly = 0; while ((8 << ly) < H) ++ly;
    p.lx = lx; p.ly = ly; p.rot = getenv("ROT") ? atoi(getenv("ROT")) : 5;
    const size_t lds = (size_t)N9 * pp_w9(TERMS) + TEAMS * pp_patch_bytes(1) + (TEAMS == 2 ? PP_TBAR_BYTES : 0);
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

// Reference of the kernel's arithmetic on a sample of output pixels (every `step`-th pixel): the same operands - GroupNorm + SiLU + operand scale,
// fp16 hi / lo split - and the same products (terms 3: a_lo w_hi + a_hi w_lo + a_hi w_hi; terms 1: a_hi w_hi), summed in fp32 in another order; weights
// read back from the packed image ([k16-step = tap * 2 + j][hi | lo][k-half][column][8 halfs], MFMA column 8 g + k = output channel 4 k + g).
__global__ void ref_kernel_pp(PPParams p, int terms, int step, int nsamp, float* ref) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsamp * 32) return;
    const int n = idx & 31, sidx = idx >> 5;
    const long pix = (long)sidx * step;
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), b = (int)(pix / ((long)p.W * p.H));
    const int col = (n & 3) * 8 + (n >> 2);
    float acc = 0.f;
    for (int c = 0; c < p.n9; ++c) {
        const PPChunk& k = p.ch[c];
        const float* cb = p.coef + ((size_t)b * 2 * p.coef_stride + k.gn_c0);
        const float asc = p.scale[8 * b + k.seg];
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
            const float* src = k.src + ((size_t)(b * p.H + yy) * p.W + xx) * k.cstride + k.coff;
            for (int kk = 0; kk < 32; ++kk) {
                float v = src[kk] * cb[kk] + cb[p.coef_stride + kk];
                v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                v *= asc;
                const _Float16 ah = (_Float16)v, al = (_Float16)(v - (float)ah);
                const int j = kk >> 4, kh = (kk >> 3) & 1, i = kk & 7;
                const _Float16* wt = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(k.wimg) + (tap * 2 + j) * 2048);
                const float wh = (float)wt[(kh * 32 + col) * 8 + i], wl = (float)wt[512 + (kh * 32 + col) * 8 + i];
                acc += terms == 3 ? (float)al * wh + (float)ah * wl + (float)ah * wh : (float)ah * wh;
            }
        }
    }
    const float oscale = p.out_scale * (1.0f / 256.0f) * p.scale[8 * b + 4 + p.ch[p.n9 - 1].seg];
    float o = acc * oscale + (p.addvec ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f);
    if (p.residual) o += p.res_scale * p.residual[((size_t)(b * p.H + y) * p.W + x) * 32 + n];
    ref[idx] = o;
}

static bool parity(const PPParams& p, int terms, const float* out, const char* what) {
    const int step = 7, nsamp = (int)(((long)p.B * p.H * p.W + step - 1) / step);
    float* ref; (void)hipMalloc(&ref, (size_t)nsamp * 32 * 4);
    hipLaunchKernelGGL(ref_kernel_pp, dim3((nsamp * 32 + 255) / 256), dim3(256), 0, 0, p, terms, step, nsamp, ref);
    std::vector<float> hr((size_t)nsamp * 32), ho((size_t)p.B * p.H * p.W * 32);
    (void)hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    (void)hipFree(ref);
    double emax = 0, rmax = 0;
    for (int sI = 0; sI < nsamp; ++sI)
        for (int n_ = 0; n_ < 32; ++n_) {
            const double r = hr[(size_t)sI * 32 + n_], o = ho[(size_t)sI * step * 32 + n_];
            if (fabs(r) > rmax) rmax = fabs(r);
            if (!(fabs(r - o) <= emax)) emax = fabs(r - o);
        }
    const bool ok = emax <= 2e-5 * rmax;
    printf("PARITY %s (%s, terms %d): max|kernel - reference| = %.3e over %d sampled pixels x 32 channels, max|reference| = %.3e\n", ok ? "OK" : "FAIL", what, terms, emax, nsamp, rmax);
    return ok;
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
    if (getenv("PARITY")) {
        // every form the library launches at this chunk count, both split modes, against the reference kernel
        bool ok = true;
        auto clr = [&]() { (void)hipMemset(out, 0, n * 4); };
        if (nch == 1) {
            clr(); run<1, false, 1, 3>(p, H, W); ok &= parity(p, 3, out, "1 chunk");
            clr(); run<1, false, 1, 1>(p, H, W); ok &= parity(p, 1, out, "1 chunk");
            p.residual = res;
            clr(); run<1, true, 1, 3>(p, H, W); ok &= parity(p, 3, out, "1 chunk + residual");
            clr(); run<1, true, 1, 1>(p, H, W); ok &= parity(p, 1, out, "1 chunk + residual");
        } else if (nch == 2) {
            clr(); run<2, false, 2, 3>(p, H, W); ok &= parity(p, 3, out, "2 chunks, 8-wave form");
            clr(); run<2, false, 1, 1>(p, H, W); ok &= parity(p, 1, out, "2 chunks");
        } else {
            clr(); run<3, false, 2, 3>(p, H, W); ok &= parity(p, 3, out, "3 chunks, 8-wave form");
            clr(); run<3, false, 1, 1>(p, H, W); ok &= parity(p, 1, out, "3 chunks");
        }
        printf("%s\n", ok ? "ALL PARITY OK" : "PARITY FAILED");
        return ok ? 0 : 1;
    }
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
