// Stand-alone timing + parity of conv_sp_kernel (pnpflow_amd/csrc/conv_sp.hip) on synthetic tensors, with s_memtime stamps of workgroup 0's
// phases.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pnpflow_amd/csrc -o conv_sp_probe conv_sp_probe.hip
// run: ./conv_sp_probe [H W B nch res first_step n_steps cout terms]
#define PP_PROBE_BUILD 1
#include "../../pnpflow_amd/csrc/conv_sp.hip"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace pf;

template <int MT, int NT, bool RES, int TERMS = 3>
static float run(const PPParams& p0, int H, int W) {
    auto kern = conv_sp_kernel<MT, NT, RES, TERMS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    PPParams p = p0;
    int lx = 0; while ((16 << lx) < W) ++lx;
    int ly = 0; while ((sp_rows(MT) << ly) < H) ++ly;
    p.lx = lx; p.ly = ly; p.rot = 5;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(256), sp_lds(MT, NT, TERMS), 0, p);
    const int reps = 10;
    float best = 1e30f;
    for (int batch = 0; batch < 4; ++batch) {      // best of four batches of ten launches (run-to-run noise of one batch: +-3 %)
        (void)hipEventRecord(e0);
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(256), sp_lds(MT, NT, TERMS), 0, p);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return best * 1e3f / reps;
}

// Reference of the kernel's arithmetic on a sample of output pixels (every `step`-th pixel of the launch): the same operands - GroupNorm + SiLU
// + operand scale, fp16 hi / lo split - and the same three products per term (a_lo w_hi + a_hi w_lo + a_hi w_hi), summed in fp32 in
// another order; weights read back from the packed image ([tap][hi | lo][N-tile][lane-linear 1 KiB], MFMA column 8 g + k = channel 4 k + g).
__global__ void ref_kernel(PPParams p, int nt_n, int step, int nsamp, float* ref, int terms) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int cout = 32 * nt_n;
    if (idx >= nsamp * cout) return;
    const int n = idx % cout, sidx = idx / cout;
    const long pix = (long)sidx * step;
    const int x = (int)(pix % p.W), y = (int)((pix / p.W) % p.H), b = (int)(pix / ((long)p.W * p.H));
    const int nt = n >> 5, ch = n & 31, col = (ch & 3) * 8 + (ch >> 2);
    const int tapb = (terms == 3 ? 2 : 1) * nt_n * 1024;      // terms 1 (precision mode 2): hi-only tap images, the a_hi w_hi product alone
    float acc = 0.f;
    for (int c = 0; c < p.n9; ++c) {
        const PPChunk& k = p.ch[c];
        const float* cb = p.coef + ((size_t)b * 2 * p.coef_stride + k.gn_c0);
        const float asc = p.scale[8 * b + k.seg];
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
            const float* src = k.src + ((size_t)(b * p.H + yy) * p.W + xx) * k.cstride + k.coff;
            const _Float16* wt = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(k.wimg) + tap * tapb);
            for (int kk = 0; kk < 16; ++kk) {
                float v = src[kk] * cb[kk] + cb[p.coef_stride + kk];
                v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                v *= asc;
                const _Float16 ah = (_Float16)v, al = (_Float16)(v - (float)ah);
                const int lane = col + 32 * (kk >> 3), j = kk & 7;
                const float wh = (float)wt[(nt * 1024 + lane * 16) / 2 + j], wl = terms == 3 ? (float)wt[(nt_n * 1024 + nt * 1024 + lane * 16) / 2 + j] : 0.f;
                acc += terms == 3 ? (float)al * wh + (float)ah * wl + (float)ah * wh : (float)ah * wh;
            }
        }
    }
    const float oscale = p.out_scale * (1.0f / 256.0f) * p.scale[8 * b + 4 + p.ch[p.n9 - 1].seg];
    float o = acc * oscale + (p.addvec ? p.addvec[(size_t)b * p.addvec_bs + n] : 0.f);
    if (p.residual) o += p.res_scale * p.residual[((size_t)(b * p.H + y) * p.W + x) * cout + n];
    ref[idx] = o;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 64, W = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 160, nch = argc > 4 ? atoi(argv[4]) : 8;
    const int useres = argc > 5 ? atoi(argv[5]) : 0, s0 = argc > 6 ? atoi(argv[6]) : 4, ns = argc > 7 ? atoi(argv[7]) : 2 * nch + 2;
    const int cout = argc > 8 ? atoi(argv[8]) : 128;      // 128: <2, 4> (16 x 16-pixel tiles), 64: <4, 2> (32 x 16)
    const int terms = argc > 9 ? atoi(argv[9]) : 3;       // 3: the fp32-equivalent split; 1: precision mode 2 (hi-only operands and weight images)
    const size_t wbytes = (cout == 128 ? 73728 : 36864) / (terms == 3 ? 1 : 2);
    const size_t n = (size_t)B * H * W * 128;
    float *in, *res, *out, *coef, *scale, *addv; double* stats; void* wimg;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&res, n * 4); (void)hipMalloc(&out, n * 4);
    std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.f - 1.f;
    (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&coef, (size_t)B * 2 * 1024 * 4); std::vector<float> c((size_t)B * 2 * 1024, 0.5f); (void)hipMemcpy(coef, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&scale, (size_t)B * 8 * 4); std::vector<float> sc((size_t)B * 8, 1.0f); (void)hipMemcpy(scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&addv, (size_t)B * 128 * 4); (void)hipMemset(addv, 0, (size_t)B * 128 * 4);
    (void)hipMalloc(&stats, (size_t)B * 256 * 8); (void)hipMemset(stats, 0, (size_t)B * 256 * 8);
    (void)hipMalloc(&wimg, wbytes * 24); std::vector<_Float16> w(wbytes / 2 * 24); for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(((int)(i * 40503u >> 4) % 200 - 100) * 0.01f); (void)hipMemcpy(wimg, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    PPParams p{};
    for (int i = 0; i < nch; ++i) { p.ch[i] = PPChunk{}; p.ch[i].src = in; p.ch[i].wimg = (char*)wimg + wbytes * i; p.ch[i].cstride = 128; p.ch[i].coff = 16 * (i & 7); p.ch[i].xform = 2; p.ch[i].gn_c0 = 16 * i; p.ch[i].seg = 0; }
    p.n9 = nch; p.n1 = 0; p.cout = cout; p.B = B; p.H = H; p.W = W; p.out = out; p.addvec = addv; p.addvec_bs = 128;
    p.res_scale = 1.f; p.stats_out = stats; p.out_scale = 1.f; p.coef = coef; p.coef_stride = 1024; p.scale = scale;
    p.residual = useres ? res : nullptr;
    unsigned long long* dbg; (void)hipMalloc(&dbg, 64 * 16 * 8); (void)hipMemset(dbg, 0, 64 * 16 * 8);
    auto go = [&]() -> float {
        if (terms == 1) return cout == 128 ? (useres ? run<2, 4, true, 1>(p, H, W) : run<2, 4, false, 1>(p, H, W)) : (useres ? run<4, 2, true, 1>(p, H, W) : run<4, 2, false, 1>(p, H, W));
        return cout == 128 ? (useres ? run<2, 4, true>(p, H, W) : run<2, 4, false>(p, H, W)) : (useres ? run<4, 2, true>(p, H, W) : run<4, 2, false>(p, H, W));
    };
    const float us_plain = go();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pf::g_sp_dbg), &dbg, sizeof(dbg));      // stamps on from here
    const float us = go();
    printf("without stamps: %.1f us\n", us_plain);
    {   // parity on a sample of pixels
        const int step = 7, nsamp = (int)(((long)B * H * W + step - 1) / step);
        float* ref; (void)hipMalloc(&ref, (size_t)nsamp * cout * 4);
        hipLaunchKernelGGL(ref_kernel, dim3((nsamp * cout + 255) / 256), dim3(256), 0, 0, p, cout / 32, step, nsamp, ref, terms);
        std::vector<float> hr((size_t)nsamp * cout), ho((size_t)B * H * W * cout);
        (void)hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0; long bad = -1;
        for (int sI = 0; sI < nsamp; ++sI)
            for (int n_ = 0; n_ < cout; ++n_) {
                const double r = hr[(size_t)sI * cout + n_], o = ho[(size_t)sI * step * cout + n_];
                if (fabs(r) > rmax) rmax = fabs(r);
                if (!(fabs(r - o) <= emax)) { emax = fabs(r - o); bad = (long)sI * step; }
            }
        printf("PARITY %s: max|kernel - reference| = %.3e over %d sampled pixels x %d channels, max|reference| = %.3e (worst pixel %ld)\n",
               emax <= 2e-5 * rmax ? "OK" : "FAIL", emax, nsamp, cout, rmax, bad);
    }
    std::vector<unsigned long long> hs(64 * 16); (void)hipMemcpy(hs.data(), dbg, hs.size() * 8, hipMemcpyDeviceToHost);
    printf("%d x %d x %d x 128, %d chunks of 16, residual %d: %.1f us.  workgroup 0 wave 0, cycles per chunk: taps 0-3 + wait + barrier | taps 4-7 + wait + barrier | tap 8 + epilogue\n", B, H, W, nch, useres, us);
    for (int sidx = s0; sidx < s0 + ns && sidx < 63; ++sidx) {
        const unsigned long long* q = &hs[sidx * 16];
        printf("chunk %2d start %8llu: taps0-3", sidx, q[0] - hs[s0 * 16]);
        for (int k = 1; k <= 4; ++k) printf(" %5llu", q[k] - q[k - 1]);
        printf(" | wait %5llu bar %5llu | tap4 %5llu | taps5-7 %5llu %5llu %5llu | wait %5llu bar %5llu | tap8 %5llu | to next %5llu\n", q[13] - q[4], q[5] - q[13], q[6] - q[5], q[7] - q[6], q[8] - q[7], q[9] - q[8],
               q[14] - q[9], q[10] - q[14], q[11] - q[10], hs[(sidx + 1) * 16] - q[12]);
    }
    return 0;
}
