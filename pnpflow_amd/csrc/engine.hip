// Host side of libpnpflow_hip.so: architecture walk, weight repacking, activation
// planning, kernel sequencing, the PnP-Flow outer loop (optionally as one hipGraph per
// outer iteration) and the C ABI declared in include/pnpflow_hip.h.
//
// Mirrors (file:line relative to the reference repository):
//   UNet.__init__/forward          pnpflow/models.py:302-495
//   PNP_FLOW.solve_ip inner loop   pnpflow/methods/pnp_flow.py:93, 102-121
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pnpflow_hip.h"
#include "pf_common.h"

using namespace pf;

namespace {

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; bool loaded = false; };

struct ResDesc {
    std::string prefix;      // "...a_block."
    int cin0 = 0, cin1 = 0;  // cin1 > 0: input is cat[h, skip]
    int cout = 0;
    bool attn = false;
    std::string attn_prefix;
};
struct LevelDesc { std::vector<ResDesc> blocks; std::string resample; int res_ch = 0; };

struct Tensor { float* p = nullptr; int C = 0, H = 0, W = 0; double* stats = nullptr; };

enum OpKind { OP_MEMSET, OP_TEMB, OP_BEGIN, OP_CONV, OP_SOFTMAX, OP_ATTN, OP_STATS, OP_END, OP_GN_COEF, OP_PREP,
              // NCSN++ net (engine_ncsnpp.inc)
              OP_NX_TEMB, OP_NX_IMG_IN, OP_NX_FIR, OP_NX_IMG_OUT,
              // backward-only
              OP_GN_FWD_COEF, OP_GN_BWD_PRE, OP_GN_BWD_COEF, OP_GN_BWD_POST, OP_TRANSPOSE, OP_SOFTMAX_BWD, OP_SUMPOOL };
struct Op {
    OpKind kind;
    ConvParams cp; int stride = 1, up = 0;
    int dma = 0;                                    // OP_CONV: 1 = conv_dma.hip (its operands were written by the OP_PREP in front of it)
    int use_pp = 0; PPParams ppp{};                 // OP_CONV: 1 = conv_pp.hip (persistent two-team kernel of the 32-channel level), 2 = conv_sp.hip (64- / 128-channel levels)
    PrepParams pp{};
    EdgeConvParams ep;
    TembParams tp;
    void* ptr = nullptr; size_t bytes = 0;          // memset
    float* sm = nullptr; int64_t sm_rows = 0; int sm_cols = 0;
    AttnParams ap{};
    GnCoefParams gp{};
    FirParams fp{};
    NxTembParams ntp{};
    size_t flops = 0;
    // generic slots of the backward helper ops
    const void* P[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void* O = nullptr;
    int I[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float F = 0.f;
};

enum TapeKind { TP_RES, TP_ATTN, TP_DOWN, TP_UP,
                // NCSN++ (engine_ncsnpp.inc): BigGAN block, input conv, input-pyramid FIR step, output-skip conv (+ pyramid FIR step)
                TP_NX_RES, TP_NX_CONV_IN, TP_NX_PYR_DOWN, TP_NX_OUT_SKIP };
struct TapeRec {
    TapeKind kind;
    const ResDesc* r = nullptr;
    std::string pfx;
    Tensor in0, in1, h1, out, qkv, S, o;
    bool has_in1 = false;
    // attention variants: GroupNorm groups and the (x + h) * s of AttnBlockpp
    int groups = 32; float s = 1.0f;
    // NCSN++ records
    int idx = 0;                       // module index (all_modules.<idx>)
    Tensor hact, xres, pyr, pyr_old;   // FIR views of a resampling block; Combine input; output pyramid before this level
    bool resample = false, up = false, has_pyr = false, has_old = false;
};

struct Tap { std::string name; Tensor t; };

struct Plan {
    int B = 0;
    uint64_t last_used = 0;
    int64_t bytes = 0;               // device bytes of this plan's activation buffers + statistics slab
    std::vector<Op> ops;
    std::vector<void*> allocs;
    std::vector<Tap> taps;
    size_t gemm_flops = 0;
    // per-launch GroupNorm coefficients / operand scales (one buffer: the launches of a plan are stream-ordered) and the
    // numeric-health flag word their finalisation kernel sets
    float* coef = nullptr; float* scale = nullptr; unsigned int* flags = nullptr;
    static constexpr int COEF_STRIDE = 1024;
    // retained-activation plans (VJP): forward tape + backward op list
    bool retain = false;
    std::vector<TapeRec> tape;
    Tensor t_begin, t_last;
    std::vector<Op> bops;
    size_t bwd_flops = 0;
    float* vec_scaled = nullptr; float* vjp_scale = nullptr; unsigned int* vjp_amax = nullptr;   // VJP input normalisation
    // retained forward: (mean, rstd) [B][gn_C] of every GroupNorm a conv launch applies, keyed by the norm's gamma vector on the device - written by that launch's
    // gn_coef_kernel, read by the backward's GroupNorm stages (round 6: replaces one gn_fwd_coeffs launch per GroupNorm and Euler step)
    std::map<const float*, std::pair<float*, float*>> gn_ret;
    float* nx_sigma = nullptr;         // NCSN++ retained forward: the divisor t * t_scale of its output, per image (read by the backward)
};

struct SolverBufs {
    int B = 0; size_t n = 0, ny = 0; int steps = 0, ns = 0;
    float *x = nullptr, *z = nullptr, *zt = nullptr, *v = nullptr, *scratch = nullptr, *y = nullptr;
    unsigned long long* rng = nullptr;       // device [seed, stream_base, elem_offset]: read by the interpolation kernel, so that one
                                             // captured graph serves every batch / shard
    int64_t bytes = 0;
    float *t_all = nullptr, *coef_all = nullptr, *t_cur = nullptr, *coef_cur = nullptr;
    int* iter = nullptr;
};

}  // namespace

struct NxMod {                 // one entry of NCSNpp.all_modules (ncsnpp.py:72-206), in construction order
    enum Kind { FOURIER, LINEAR, CONV_IN, RES, ATTN, COMBINE, GN, CONV_OUT } kind;
    int in_ch = 0, out_ch = 0;   // RES: in_ch is the (concatenated) input width
    bool up = false, down = false;
};

struct pf_engine {
    int device = 0;
    pf_unet_cfg cfg{};
    int arch = 0;                    // 0: the OT U-Net (pnpflow/models.py); 1: NCSN++ (pnpflow/image_generation/models/ncsnpp.py)
    pf_ncsnpp_cfg ncfg{};
    std::vector<NxMod> nx_mods;
    float solver_time_scale = 1.0f;  // model label = t * this inside the solver loops (methods/pnp_flow.py:23-27)
    std::string err;
    std::vector<std::pair<std::string, std::vector<int64_t>>> expected;   // state_dict order
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    // architecture
    std::vector<LevelDesc> down, up;
    ResDesc mid0, mid2; std::string mid_attn;
    int mid_ch = 0;
    std::vector<std::pair<std::string, int>> res_order;   // every ResidualBlock prefix (+cout), in forward order
    // device weights
    std::map<std::string, float*> dev;     // cache of uploaded / packed arrays
    struct W16Src { std::string name; int lo, hi; };
    std::map<const void*, W16Src> w16_src;  // split-fp16 repack -> the host tensor slice it was made from (for the hi-only repack of the single-term mode)
    std::vector<void*> weight_allocs;
    int temb_total = 0;
    std::map<std::string, int> temb_off;   // ResBlock prefix -> offset in the stacked temb projection
    std::map<int, std::unique_ptr<Plan>> plans;
    Plan* last_plan = nullptr;       // plan of the most recent forward (activation taps are read from it)
    Plan* retained_plan = nullptr;   // plan of the most recent pf_unet_forward_retain (pf_unet_backward must walk the same one)
    int retained_B = 0;
    uint64_t plan_clock = 0;         // LRU stamp source for the plan cache
    int64_t bytes = 0;               // device bytes currently held by this engine (weights, activation plans, solver buffers)
    // cached hipGraph of one PnP-Flow outer iteration (re-used while the captured arguments stay the same)
    struct GraphKey { const void* plan; int kind, half, sf, ntaps; const void* mask; const void* taps; const void* noise;
                      int num_samples, batch_samples, noise_model, B; };
    GraphKey gkey{}; hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    int precision = 1;   // 1 (default): split-fp16 (3 x f16 MFMA, fp32-equivalent) for the packed-weight convs; 0: exact fp32 MFMA;
                         // 2: single fp16 MFMA per product (fp16 operands, fp32 accumulate - TF32-class, include/pnpflow_hip.h)
    SolverBufs sb;
    // OT-ODE loop (pf_ot_ode_restore): iterate, velocity, solve output, J^T vec, per-iteration schedule tables, cached graph
    struct OdeBufs { int B = 0; size_t n = 0, ny = 0; int steps = 0; bool blur = false;
                     float *x = nullptr, *vt = nullptr, *vec = nullptr, *g = nullptr, *y = nullptr, *scratch = nullptr;
                     float *tab = nullptr /* [4][steps]: t, 1-t, r_t^2, coef */, *cur = nullptr /* [4][B] */; int* iter = nullptr; int64_t bytes = 0; } ob;
    struct OdeKey { const void* plan; int kind, half, sf, ntaps; const void* mask; const void* taps; int B; float sigma2, delta; int pad_; };   // compared with memcmp: no implicit padding (static_assert below)
    OdeKey okey{}; hipGraph_t ograph = nullptr; hipGraphExec_t ogexec = nullptr;
    hipStream_t work_stream = nullptr;   // used when the caller passes the NULL stream and asks for graph replay
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<const Op*> ev_ops;   // op of each recorded event pair (profiling detail)
    size_t ev_used = 0;
    int64_t prof_launches = 0; double prof_ms = 0.0, prof_flops = 0.0;
};

static thread_local std::string g_create_err;

// PNPFLOW_HIP_POISON=1 (tests): every fresh device allocation of the engine is filled with 0xFF bytes (NaNs as fp32 / fp64), so that a
// read of memory the engine never wrote shows up as a numeric-health error or a parity failure instead of depending on what
// the allocation happened to hold (a fresh box hands out zeroed HBM; the second process on a box does not)
static int poison_mask() { static const int m = getenv("PNPFLOW_HIP_POISON") ? atoi(getenv("PNPFLOW_HIP_POISON")) : 0; return m == 1 ? 7 : m; }      // 1 = everything; else bit 0 plans, bit 1 PnP-Flow buffers, bit 2 OT-ODE buffers
static bool poison_enabled() { return poison_mask() != 0; }
static void poison(void* p, size_t bytes, int group = 1) {
    // (device-wide synchronisation on both sides: the fill runs on the NULL stream, which does not order with the non-blocking streams
    //  the solvers hand over - without it the fill can land AFTER the first kernels that write the buffer)
    if ((poison_mask() & group) && p) { hipDeviceSynchronize(); hipMemset(p, 0xFF, bytes); hipDeviceSynchronize(); }
}

// selects the engine's device for the duration of an ABI call and restores the caller's current device afterwards
struct DeviceGuard {
    int prev = -1; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; err = hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
#define USE_DEVICE(e) DeviceGuard _dg((e)->device); HIPCHK(e, _dg.err)

#define LAUNCHCHK(call)                                                        \
    do { hipError_t _r = (call); if (_r != hipSuccess) return _r == hipErrorInvalidValue ? PF_ERR_INVALID : PF_ERR_HIP; } while (0)


#define HIPCHK(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) {                                                                \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_r);                      \
            return PF_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

// --------------------------------------------------------------------------------------
// architecture walk (models.py:337-436) -> descriptors + expected state_dict entries
// --------------------------------------------------------------------------------------
static bool in_attn_res(const pf_unet_cfg& c, int h) {
    for (int i = 0; i < c.num_attn_resolutions; ++i) if (c.attn_resolutions[i] == h) return true;
    return false;
}

static void expect(pf_engine* e, const std::string& n, std::vector<int64_t> s) { e->expected.emplace_back(n, std::move(s)); }
static void expect_lin(pf_engine* e, const std::string& p, int i, int o) { expect(e, p + "weight", {o, i}); expect(e, p + "bias", {o}); }
static void expect_conv(pf_engine* e, const std::string& p, int i, int o, int k) { expect(e, p + "weight", {o, i, k, k}); expect(e, p + "bias", {o}); }
static void expect_gn(pf_engine* e, const std::string& p, int c) { expect(e, p + "weight", {c}); expect(e, p + "bias", {c}); }
static void expect_res(pf_engine* e, const std::string& p, int i, int o, int tch) {
    expect_lin(e, p + "temb_proj.", tch, o); expect_gn(e, p + "norm1.", i); expect_conv(e, p + "conv1.", i, o, 3);
    expect_gn(e, p + "norm2.", o); expect_conv(e, p + "conv2.", o, o, 3);
    if (i != o) expect_conv(e, p + "shortcut.", i, o, 1);
}
static void expect_attn(pf_engine* e, const std::string& p, int c) {
    expect_conv(e, p + "attn_q.", c, c, 1); expect_conv(e, p + "attn_k.", c, c, 1);
    expect_conv(e, p + "attn_v.", c, c, 1); expect_conv(e, p + "proj_out.", c, c, 1);
    expect_gn(e, p + "norm.", c);
}

static int build_arch(pf_engine* e) {
    const pf_unet_cfg& c = e->cfg;
    const int ch = c.ch, nlev = c.num_levels, nres = c.num_res_blocks, tch = 4 * ch;
    char buf[128];
    expect_lin(e, "temb_net.main.0.", ch, tch); expect_lin(e, "temb_net.main.2.", tch, tch);
    expect_conv(e, "begin_conv.", c.input_channels, ch, 3);
    std::vector<int> chans{ch};
    int cur = ch, h = c.input_height;
    for (int lvl = 0; lvl < nlev; ++lvl) {
        LevelDesc L;
        const int o = ch * c.ch_mult[lvl];
        for (int blk = 0; blk < nres; ++blk) {
            ResDesc r;
            snprintf(buf, sizeof buf, "down_modules.%d.%da_%da_block.", lvl, lvl, blk); r.prefix = buf;
            r.cin0 = cur; r.cout = o;
            expect_res(e, r.prefix, cur, o, tch);
            if (in_attn_res(c, h)) {
                r.attn = true;
                snprintf(buf, sizeof buf, "down_modules.%d.%da_%db_attn.", lvl, lvl, blk); r.attn_prefix = buf;
                expect_attn(e, r.attn_prefix, o);
            }
            L.blocks.push_back(r); chans.push_back(o); cur = o;
        }
        if (lvl != nlev - 1) {
            snprintf(buf, sizeof buf, "down_modules.%d.%db_downsample.", lvl, lvl); L.resample = buf; L.res_ch = o;
            expect_conv(e, L.resample, o, o, 3);
            h /= 2; chans.push_back(o);
        }
        e->down.push_back(L);
    }
    e->mid_ch = cur;
    e->mid0.prefix = "mid_modules.0."; e->mid0.cin0 = cur; e->mid0.cout = cur;
    e->mid2.prefix = "mid_modules.2."; e->mid2.cin0 = cur; e->mid2.cout = cur;
    e->mid_attn = "mid_modules.1.";
    expect_res(e, e->mid0.prefix, cur, cur, tch); expect_attn(e, e->mid_attn, cur); expect_res(e, e->mid2.prefix, cur, cur, tch);
    for (int idx = 0; idx < nlev; ++idx) {
        const int lvl = nlev - 1 - idx;
        LevelDesc L;
        const int o = ch * c.ch_mult[lvl];
        for (int blk = 0; blk < nres + 1; ++blk) {
            ResDesc r;
            snprintf(buf, sizeof buf, "up_modules.%d.%da_%da_block.", idx, lvl, blk); r.prefix = buf;
            r.cin0 = cur; r.cin1 = chans.back(); chans.pop_back(); r.cout = o;
            expect_res(e, r.prefix, r.cin0 + r.cin1, o, tch);
            if (in_attn_res(c, h)) {
                r.attn = true;
                snprintf(buf, sizeof buf, "up_modules.%d.%da_%db_attn.", idx, lvl, blk); r.attn_prefix = buf;
                expect_attn(e, r.attn_prefix, o);
            }
            L.blocks.push_back(r); cur = o;
        }
        if (lvl != 0) {
            snprintf(buf, sizeof buf, "up_modules.%d.%db_upsample.up_conv.", idx, lvl); L.resample = buf; L.res_ch = o;
            expect_conv(e, L.resample, o, o, 3);
            h *= 2;
        }
        e->up.push_back(L);
    }
    if (!chans.empty()) { e->err = "internal: skip stack not empty"; return PF_ERR_INVALID; }
    expect_gn(e, "end_conv.0.", cur); expect_conv(e, "end_conv.2.", cur, c.output_channels, 3);
    if (cur != ch) { e->err = "internal: final width != ch"; return PF_ERR_INVALID; }
    // forward-order list of residual blocks for the stacked temb projection
    for (auto& L : e->down) for (auto& r : L.blocks) e->res_order.emplace_back(r.prefix, r.cout);
    e->res_order.emplace_back(e->mid0.prefix, e->mid0.cout); e->res_order.emplace_back(e->mid2.prefix, e->mid2.cout);
    for (auto& L : e->up) for (auto& r : L.blocks) e->res_order.emplace_back(r.prefix, r.cout);
    int off = 0;
    for (auto& pr : e->res_order) { e->temb_off[pr.first] = off; off += pr.second; }
    e->temb_total = off;
    return PF_OK;
}

// --------------------------------------------------------------------------------------
// device weights
// --------------------------------------------------------------------------------------
static const HostTensor& W(pf_engine* e, const std::string& n) { return e->host.at(n); }

static float* upload(pf_engine* e, const std::string& key, const std::vector<float>& v) {
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    float* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(v.size(), 4) * sizeof(float)) != hipSuccess) return nullptr;
    e->bytes += (int64_t)(std::max<size_t>(v.size(), 4) * sizeof(float));
    hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
    e->weight_allocs.push_back(d);
    e->dev[key] = d;
    return d;
}

// OIHW conv weight, input channels [lo,hi) -> fragment-major [chunk][tap][kstep(2)][Cout][8]
// (zero padded K tail): the B fragment of one wave (32 channels x 8 k) is contiguous.
static float* packed_conv(pf_engine* e, const std::string& wname, int lo, int hi) {
    const std::string key = wname + "#" + std::to_string(lo) + ":" + std::to_string(hi);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    const int C = hi - lo, nchunk = (C + CONV_KC - 1) / CONV_KC;
    std::vector<float> out((size_t)nchunk * kk * O * CONV_KC, 0.f);
    for (int chn = 0; chn < nchunk; ++chn)
        for (int tap = 0; tap < kk; ++tap)
            for (int n = 0; n < O; ++n)
                for (int k = 0; k < CONV_KC; ++k) {
                    const int c = chn * CONV_KC + k;
                    if (c < C)
                        out[((((size_t)chn * kk + tap) * 2 + k / 8) * O + n) * 8 + k % 8] = t.data[((size_t)n * I + lo + c) * kk + tap];
                }
    return upload(e, key, out);
}

// split-fp16 repack of the same slice: [chunk][tap][hi | lo][Cout][16] halfs, values pre-scaled by 2^8
static const void* packed_conv16(pf_engine* e, const std::string& wname, int lo, int hi) {
    const std::string key = wname + "#h" + std::to_string(lo) + ":" + std::to_string(hi);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    const int C = hi - lo, nchunk = (C + CONV_KC - 1) / CONV_KC;
    std::vector<_Float16> out((size_t)nchunk * kk * O * 32, (_Float16)0.f);
    for (int chn = 0; chn < nchunk; ++chn)
        for (int tap = 0; tap < kk; ++tap)
            for (int n = 0; n < O; ++n)
                for (int k = 0; k < CONV_KC; ++k) {
                    const int c = chn * CONV_KC + k;
                    if (c >= C) continue;
                    const float w = t.data[((size_t)n * I + lo + c) * kk + tap] * 256.0f;
                    const _Float16 h = (_Float16)w;
                    const _Float16 l = (_Float16)(w - (float)h);
                    // block (16-channel slice, tap) = [hi halves: Cout x 16][lo halves: Cout x 16]: the hi (lo) fragment load of a wave
                    // (32 output channels x 32 B) is ONE contiguous 1 KiB run - with hi and lo interleaved per output channel every
                    // fragment load touched 2 KiB of half-used lines (round 3: the texture addresser is the busiest unit of the
                    // 32-channel level, profiles/r03_pmc_level0_counters.md)
                    const size_t blk = ((size_t)chn * kk + tap) * O * 32;
                    out[blk + (size_t)n * 16 + k] = h; out[blk + (size_t)O * 16 + (size_t)n * 16 + k] = l;
                }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    const void* d = upload(e, key, raw);
    if (d) e->w16_src[d] = {wname, lo, hi};
    return d;
}

// hi-only repack of the same slice for the single-term (precision mode 2) LDS-DMA kernel: [slice16][tap][Cout][16 halfs], values
// pre-scaled by 2^8 like the split repack (so that the epilogue's 2^-8 is shared) - no low halves are stored, fetched or multiplied
static const void* packed_conv16h(pf_engine* e, const std::string& wname, int lo, int hi) {
    const std::string key = wname + "#h1_" + std::to_string(lo) + ":" + std::to_string(hi);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    const int C = hi - lo, nchunk = (C + CONV_KC - 1) / CONV_KC;
    std::vector<_Float16> out((size_t)nchunk * kk * O * 16, (_Float16)0.f);
    for (int chn = 0; chn < nchunk; ++chn)
        for (int tap = 0; tap < kk; ++tap)
            for (int n = 0; n < O; ++n)
                for (int k = 0; k < CONV_KC; ++k) {
                    const int c = chn * CONV_KC + k;
                    if (c >= C) continue;
                    out[(((size_t)chn * kk + tap) * O + n) * 16 + k] = (_Float16)(t.data[((size_t)n * I + lo + c) * kk + tap] * 256.0f);
                }
    std::vector<float> raw((out.size() + 1) / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    return upload(e, key, raw);
}

// LDS weight image of ONE 32-channel K-chunk for conv_pp.hip (Cout = 32): [k16-step s = tap * 2 + j][hi | lo][k-half][column][8 halfs], channel of
// (s, k-half, i) = lo + j * 16 + k-half * 8 + i; the same x 2^8 pre-scale and hi / lo split as packed_conv16, so the products are the same
static const void* packed_conv_pp(pf_engine* e, const std::string& wname, int lo) {
    const std::string key = wname + "#pp" + std::to_string(lo);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    std::vector<_Float16> out((size_t)kk * 2 * 2 * 2 * O * 8, (_Float16)0.f);
    for (int tap = 0; tap < kk; ++tap)
        for (int j = 0; j < 2; ++j)
            for (int kh = 0; kh < 2; ++kh)
                for (int n = 0; n < O; ++n)
                    for (int i = 0; i < 8; ++i) {
                        const int c = lo + j * 16 + kh * 8 + i;
                        const int oc = 4 * (n & 7) + (n >> 3);      // MFMA column n = 8 g + k carries output channel 4 k + g (conv_pp.hip's epilogue transpose)
                        const float w = t.data[((size_t)oc * I + c) * kk + tap] * 256.0f;
                        const _Float16 h = (_Float16)w;
                        const _Float16 l = (_Float16)(w - (float)h);
                        const size_t blk = (size_t)(tap * 2 + j) * 2 * (2 * O * 8);
                        out[blk + ((size_t)kh * O + n) * 8 + i] = h;
                        out[blk + (size_t)2 * O * 8 + ((size_t)kh * O + n) * 8 + i] = l;
                    }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    return upload(e, key, raw);
}

// LDS weight image of ONE 16-channel K-chunk for conv_sp.hip at Cout = 64 (3x3): [tap][hi | lo][N-tile][k-half][column][8 halfs], input channel
// of (k-half, i) = lo + k-half * 8 + i, output channel of (N-tile, column n) = 32 N-tile + 4 (n & 7) + (n >> 3); same x 2^8 pre-scale and split
static const void* packed_conv_sp64(pf_engine* e, const std::string& wname, int lo) {
    const std::string key = wname + "#sp64_" + std::to_string(lo);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    if (O != 64 || kk != 9) return nullptr;
    std::vector<_Float16> out((size_t)9 * 2 * 2 * 2 * 32 * 8, (_Float16)0.f);
    for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < 2; ++nt)
            for (int kh = 0; kh < 2; ++kh)
                for (int n = 0; n < 32; ++n)
                    for (int i = 0; i < 8; ++i) {
                        const int c = lo + kh * 8 + i;
                        const int oc = 32 * nt + 4 * (n & 7) + (n >> 3);
                        const float w = t.data[((size_t)oc * I + c) * 9 + tap] * 256.0f;
                        const _Float16 h = (_Float16)w;
                        const _Float16 l = (_Float16)(w - (float)h);
                        const size_t inner = ((size_t)kh * 32 + n) * 8 + i;
                        out[(((size_t)tap * 2 + 0) * 2 + nt) * 512 + inner] = h;
                        out[(((size_t)tap * 2 + 1) * 2 + nt) * 512 + inner] = l;
                    }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    return upload(e, key, raw);
}

// LDS weight image of ONE 16-channel K-chunk for conv_sp.hip at Cout = 128 (3x3): [tap][hi | lo][N-tile 0..3][k-half][column][8 halfs] = 9 tap
// slots of 8 KiB; input channel of (k-half, i) = lo + k-half * 8 + i, output channel of (N-tile, column n) = 32 N-tile + 4 (n & 7) + (n >> 3);
// same x 2^8 pre-scale and split as packed_conv16
static const void* packed_conv_sp128(pf_engine* e, const std::string& wname, int lo) {
    const std::string key = wname + "#sp128_" + std::to_string(lo);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    if (O != 128 || kk != 9) return nullptr;
    std::vector<_Float16> out((size_t)9 * 2 * 4 * 512, (_Float16)0.f);
    for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < 4; ++nt)
            for (int kh = 0; kh < 2; ++kh)
                for (int n = 0; n < 32; ++n)
                    for (int i = 0; i < 8; ++i) {
                        const int c = lo + kh * 8 + i;
                        const int oc = 32 * nt + 4 * (n & 7) + (n >> 3);
                        const float w = t.data[((size_t)oc * I + c) * 9 + tap] * 256.0f;
                        const _Float16 h = (_Float16)w;
                        const _Float16 l = (_Float16)(w - (float)h);
                        const size_t inner = ((size_t)kh * 32 + n) * 8 + i;
                        out[(((size_t)tap * 2 + 0) * 4 + nt) * 512 + inner] = h;
                        out[(((size_t)tap * 2 + 1) * 4 + nt) * 512 + inner] = l;
                    }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    return upload(e, key, raw);
}

// hi-only image of the same chunk for conv_sp's TERMS = 1 form (precision mode 2): [tap][N-tile][k-half][column][8 halfs] = 9 taps of NT KiB, values
// RNE16(w x 2^8) - the hi halves of the images above, i.e. the weights conv_mfma16's TERMS = 1 form multiplies with
static const void* packed_conv_sp_h(pf_engine* e, const std::string& wname, int lo, int cout) {
    const std::string key = wname + "#sph" + std::to_string(cout) + "_" + std::to_string(lo);
    auto it = e->dev.find(key);
    if (it != e->dev.end()) return it->second;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1], kk = (int)(t.shape[2] * t.shape[3]);
    if (O != cout || (cout != 64 && cout != 128) || kk != 9) return nullptr;
    const int NT = cout / 32;
    std::vector<_Float16> out((size_t)9 * NT * 512, (_Float16)0.f);
    for (int tap = 0; tap < 9; ++tap)
        for (int nt = 0; nt < NT; ++nt)
            for (int kh = 0; kh < 2; ++kh)
                for (int n = 0; n < 32; ++n)
                    for (int i = 0; i < 8; ++i) {
                        const int c = lo + kh * 8 + i;
                        const int oc = 32 * nt + 4 * (n & 7) + (n >> 3);
                        out[((size_t)tap * NT + nt) * 512 + ((size_t)kh * 32 + n) * 8 + i] = (_Float16)(t.data[((size_t)oc * I + c) * 9 + tap] * 256.0f);
                    }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * sizeof(_Float16));
    return upload(e, key, raw);
}

static void fill_packed_seg(ConvSeg& s, const float* w, int taps, int Cout) {
    (void)taps; (void)Cout;
    s.w = w; s.w_mode = 0; s.w_bs = 0; s.w_cs = 0; s.w_ts = 0; s.w_ns = 0; s.w_ks = 0; s.w16 = nullptr;
}

// --------------------------------------------------------------------------------------
// plan construction
// --------------------------------------------------------------------------------------
struct Builder {
    pf_engine* e; Plan* plan; int B; bool ok = true;
    std::multimap<size_t, void*> free_list;
    std::map<void*, size_t> sizes;
    size_t stats_bytes = 0;
    std::vector<std::pair<double**, size_t>> stats_fix;   // (where to store, offset) resolved after the walk

    float* acquire(size_t nfloats) {
        const size_t bytes = ((nfloats * sizeof(float) + 255) / 256) * 256;
        auto it = free_list.find(bytes);
        if (it != free_list.end()) { void* p = it->second; free_list.erase(it); return (float*)p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; e->err = "hipMalloc failed (activations)"; return nullptr; }
        poison(p, bytes);
        plan->allocs.push_back(p); sizes[p] = bytes; plan->bytes += (int64_t)bytes; e->bytes += (int64_t)bytes;
        return (float*)p;
    }
    bool keep = getenv("PNPFLOW_HIP_KEEP_ACTIVATIONS") != nullptr;   // never reuse: taps stay readable / activations retained for the VJP
    void release(float* p) { if (p && !keep) free_list.emplace(sizes[p], p); }
    void recycle(float* p) { if (p) free_list.emplace(sizes[p], p); }   // backward temporaries are always recycled
    Tensor make(int C, int H, int W, bool stats) {
        Tensor t; t.C = C; t.H = H; t.W = W; t.p = acquire((size_t)B * H * W * C);
        if (stats) { t.stats = (double*)(uintptr_t)(stats_bytes + 1); stats_bytes += (size_t)B * C * 2 * sizeof(double); }
        return t;
    }
};

static ConvParams base_params(int B, int H, int W, int Hs, int Ws, const Tensor& out) {
    ConvParams p{};
    p.B = B; p.H = H; p.W = W; p.Hs = Hs; p.Ws = Ws; p.Cout = out.C; p.out = out.p; p.out_cstride = out.C;
    p.out_scale = 1.0f; p.res_scale = 1.0f; p.stats_out = out.stats; p.gn_eps = 1e-6f;
    return p;
}

static void add_seg(ConvParams& p, const Tensor& t, int xform, int taps, int gn_off) {
    ConvSeg& s = p.seg[p.nseg++];
    s.src = t.p; s.C = t.C; s.cstride = t.C; s.coff = 0; s.xform = xform; s.taps = taps; s.gn_off = gn_off; s.stats = t.stats;
}

// GroupNorm coefficients (and, in the split-fp16 mode, the power-of-two operand scale) of a conv launch are finalised by a
// micro-launch right before it; returns the ConvParams with coef / scale filled in
static void ensure_coef(Builder& bd) {
    Plan* plan = bd.plan;
    if (plan->coef) return;
    plan->coef = bd.acquire((size_t)bd.B * 2 * Plan::COEF_STRIDE);
    plan->scale = bd.acquire((size_t)bd.B * 8);
    plan->flags = reinterpret_cast<unsigned int*>(bd.acquire(64));
    if (plan->flags) { hipMemset(plan->flags, 0, 64); hipDeviceSynchronize(); }      // (NULL-stream fill: not ordered with the non-blocking streams the launches use)
}

static ConvParams with_coef(Builder& bd, ConvParams p, std::vector<Op>& ops) {
    Plan* plan = bd.plan;
    ensure_coef(bd);
    bool raw_stats = false, packed16 = true;
    for (int i = 0; i < p.nseg; ++i) {
        raw_stats |= p.seg[i].xform == 0 && p.seg[i].stats != nullptr;
        packed16 &= p.seg[i].w_mode == 0 && p.seg[i].w16 != nullptr;
    }
    const bool want_scale = bd.e->precision != 0 && packed16 && (raw_stats || p.gn_C > 0);
    if (p.gn_C == 0 && !want_scale) return p;
    Op op{}; op.kind = OP_GN_COEF;
    GnCoefParams& g = op.gp;
    g.nseg = p.nseg;
    for (int i = 0; i < p.nseg; ++i) { g.st[i] = p.seg[i].stats; g.C[i] = p.seg[i].C; g.xform[i] = p.seg[i].xform; g.gn_off[i] = p.seg[i].gn_off; }
    g.gn_C = p.gn_C; g.gn_cpg = p.gn_cpg; g.HW = p.Hs * p.Ws; g.eps = p.gn_eps; g.gamma = p.gamma; g.beta = p.beta;
    g.coef = plan->coef; g.coef_stride = Plan::COEF_STRIDE; g.scale = want_scale ? plan->scale : nullptr; g.flags = plan->flags; g.id = (int)ops.size();
    static const bool ret_env = !(getenv("PNPFLOW_HIP_GN_RETAIN") && atoi(getenv("PNPFLOW_HIP_GN_RETAIN")) == 0);      // test-only A/B switch
    if (plan->retain && ret_env && p.gn_C > 0 && &ops == &plan->ops && !plan->gn_ret.count(p.gamma)) {
        float* mu = bd.acquire((size_t)bd.B * p.gn_C); float* rs = bd.acquire((size_t)bd.B * p.gn_C);
        if (mu && rs) { g.mu_out = mu; g.rs_out = rs; plan->gn_ret[p.gamma] = {mu, rs}; }
    }
    ops.push_back(op);
    p.coef = plan->coef; p.coef_stride = Plan::COEF_STRIDE; p.scale = want_scale ? plan->scale : nullptr;
    return p;
}

// LDS-DMA path (conv_dma.hip): if the launch qualifies, the operands of its K-segments are pre-transformed by one OP_PREP in front of it
// (after the OP_GN_COEF that finalises its GroupNorm coefficients / operand scales).  The a16 buffers are launch-local temporaries.
static bool attach_dma(Builder& bd, ConvParams& p, int stride, int up, std::vector<Op>& ops) {
    pf_engine* e = bd.e;
    if (e->precision == 0) return false;
    const int terms = e->precision == 2 ? 1 : 3;
    if (terms == 1)
        for (int i = 0; i < p.nseg; ++i) {
            auto it = e->w16_src.find(p.seg[i].w16);
            if (p.seg[i].w_mode != 0 || it == e->w16_src.end()) return false;
            if ((it->second.hi - it->second.lo) % 64 != 0) return false;
            p.seg[i].w16h = packed_conv16h(e, it->second.name, it->second.lo, it->second.hi);
        }
    if (!conv_dma_supported(p, stride, up, terms)) return false;
    Op op{}; op.kind = OP_PREP;
    PrepParams& q = op.pp;
    q.nseg = p.nseg; q.B = p.B; q.Hs = p.Hs; q.Ws = p.Ws; q.terms = terms;
    q.coef = p.coef; q.coef_stride = p.coef_stride; q.scale = p.scale;
    std::vector<float*> tmp;
    for (int i = 0; i < p.nseg; ++i) {
        ConvSeg& sg = p.seg[i];
        float* a = bd.acquire(conv_dma_a16_bytes(p.B, sg.C, p.Hs, p.Ws, terms) / sizeof(float));
        if (!a) return false;
        tmp.push_back(a);
        sg.a16 = a;
        q.src[i] = sg.src; q.dst[i] = a; q.C[i] = sg.C; q.cstride[i] = sg.cstride; q.coff[i] = sg.coff; q.xform[i] = sg.xform; q.gn_off[i] = sg.gn_off;
    }
    ops.push_back(op);
    for (float* a : tmp) bd.recycle(a);      // stream order: the next launch that may get these bytes starts after this conv has finished
    return true;
}

// conv_pp.hip (persistent two-team kernel of the 32-channel level: 32-channel K-chunks) / conv_sp.hip (one wave per SIMD, the 64- and 128-channel
// levels: 16-channel K-chunks, weights streamed through LDS slots): the launch becomes a list of K-chunks in the kernel argument
static int attach_pp(Builder& bd, const ConvParams& p, int stride, int up, PPParams& q) {      // 0: no; 1: conv_pp; 2: conv_sp
    pf_engine* e = bd.e;
    if (e->precision == 0) return 0;
    // precision mode 2 (one MFMA per product): the hi-only TERMS = 1 forms of both persistent kernels (round 6)
    const int terms = e->precision == 2 ? 1 : 3;
    const bool sp = (p.Cout == 128 || p.Cout == 64) && conv_sp_supported(p, stride, up, terms);
    if (!sp && !(p.Cout == 32 && conv_pp_supported(p, stride, up, terms))) return 0;
    q = PPParams{};
    q.cout = p.Cout;
    const int kc = sp ? 16 : 32;
    int n = 0;
    for (int i = 0; i < p.nseg; ++i) {
        const ConvSeg& sg = p.seg[i];
        auto it = e->w16_src.find(sg.w16);
        if (it == e->w16_src.end()) return 0;
        for (int cc = 0; cc < sg.C / kc; ++cc) {
            PPChunk& k = q.ch[n++];
            k.src = sg.src; k.cstride = sg.cstride; k.coff = sg.coff + cc * kc; k.xform = sg.xform;
            k.gn_c0 = sg.gn_off + cc * kc; k.seg = i;
            k.wimg = !sp ? packed_conv_pp(e, it->second.name, it->second.lo + cc * kc)
                     : terms == 1 ? packed_conv_sp_h(e, it->second.name, it->second.lo + cc * kc, p.Cout)
                     : p.Cout == 128 ? packed_conv_sp128(e, it->second.name, it->second.lo + cc * kc) : packed_conv_sp64(e, it->second.name, it->second.lo + cc * kc);
            if (!k.wimg) return 0;
            (sg.taps == 9 ? q.n9 : q.n1) += 1;
        }
    }
    q.B = p.B; q.H = p.H; q.W = p.W;
    q.out = p.out; q.addvec = p.addvec; q.addvec_bs = p.addvec_bs; q.residual = p.residual; q.res_scale = p.res_scale;
    q.stats_out = p.stats_out; q.out_scale = p.out_scale; q.coef = p.coef; q.coef_stride = p.coef_stride; q.scale = p.scale;
    return sp ? 2 : 1;
}

static void push_conv(Builder& bd, const ConvParams& p0, int stride = 1, int up = 0) {
    ConvParams p = with_coef(bd, p0, bd.plan->ops);
    Op op{}; op.kind = OP_CONV;
    op.use_pp = attach_pp(bd, p, stride, up, op.ppp);
    op.dma = (!op.use_pp && attach_dma(bd, p, stride, up, bd.plan->ops)) ? 1 : 0;
    op.cp = p; op.stride = stride; op.up = up; op.flops = conv_flops(p);
    bd.plan->gemm_flops += op.flops;
    bd.plan->ops.push_back(op);
}

// ResidualBlock (models.py:94-113).  in1.p != nullptr: input is cat[in0, in1].
static Tensor res_block(Builder& bd, const ResDesc& r, const Tensor& in0, const Tensor* in1, float* temb_vec) {
    pf_engine* e = bd.e; const int B = bd.B, H = in0.H, Wd = in0.W;
    const int cin = in0.C + (in1 ? in1->C : 0);
    // conv1: GN1+SiLU on the (concatenated) input, + bias + temb projection
    Tensor h1 = bd.make(r.cout, H, Wd, true);
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, h1);
        add_seg(p, in0, 2, 9, 0);
        fill_packed_seg(p.seg[0], packed_conv(e, r.prefix + "conv1.weight", 0, in0.C), 9, r.cout);
        p.seg[0].w16 = packed_conv16(e, r.prefix + "conv1.weight", 0, in0.C);
        if (in1) {
            add_seg(p, *in1, 2, 9, in0.C);
            fill_packed_seg(p.seg[1], packed_conv(e, r.prefix + "conv1.weight", in0.C, cin), 9, r.cout);
            p.seg[1].w16 = packed_conv16(e, r.prefix + "conv1.weight", in0.C, cin);
        }
        p.gn_C = cin; p.gn_cpg = cin / 32;
        p.gamma = upload(e, r.prefix + "norm1.weight", W(e, r.prefix + "norm1.weight").data);
        p.beta = upload(e, r.prefix + "norm1.bias", W(e, r.prefix + "norm1.bias").data);
        p.addvec = temb_vec + e->temb_off.at(r.prefix); p.addvec_bs = e->temb_total;
        push_conv(bd, p);
    }
    // conv2: GN2+SiLU, + bias, + shortcut (identity residual or 1x1 conv folded in as K-segments)
    Tensor out = bd.make(r.cout, H, Wd, true);
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, out);
        add_seg(p, h1, 2, 9, 0);
        fill_packed_seg(p.seg[0], packed_conv(e, r.prefix + "conv2.weight", 0, r.cout), 9, r.cout);
        p.seg[0].w16 = packed_conv16(e, r.prefix + "conv2.weight", 0, r.cout);
        p.gn_C = r.cout; p.gn_cpg = r.cout / 32;
        p.gamma = upload(e, r.prefix + "norm2.weight", W(e, r.prefix + "norm2.weight").data);
        p.beta = upload(e, r.prefix + "norm2.bias", W(e, r.prefix + "norm2.bias").data);
        std::vector<float> bias = W(e, r.prefix + "conv2.bias").data;
        if (cin != r.cout) {
            const std::string sw = r.prefix + "shortcut.weight";
            add_seg(p, in0, 0, 1, 0);
            fill_packed_seg(p.seg[p.nseg - 1], packed_conv(e, sw, 0, in0.C), 1, r.cout);
            p.seg[p.nseg - 1].w16 = packed_conv16(e, sw, 0, in0.C);
            if (in1) {
                add_seg(p, *in1, 0, 1, 0);
                fill_packed_seg(p.seg[p.nseg - 1], packed_conv(e, sw, in0.C, cin), 1, r.cout);
                p.seg[p.nseg - 1].w16 = packed_conv16(e, sw, in0.C, cin);
            }
            const auto& sb = W(e, r.prefix + "shortcut.bias").data;
            for (size_t i = 0; i < bias.size(); ++i) bias[i] += sb[i];
        } else {
            p.residual = in0.p; p.res_cstride = in0.C;
        }
        p.addvec = upload(e, r.prefix + "conv2.bias+sc", bias); p.addvec_bs = 0;
        push_conv(bd, p);
    }
    bd.release(h1.p);
    if (bd.plan->retain) {
        TapeRec tr; tr.kind = TP_RES; tr.r = &r; tr.in0 = in0; tr.has_in1 = in1 != nullptr; if (in1) tr.in1 = *in1; tr.h1 = h1; tr.out = out;
        bd.plan->tape.push_back(tr);
    }
    return out;
}

// SelfAttention (models.py:145-162)
// groups / s: the NCSN++ variant (AttnBlockpp, layerspp.py:61-94) normalises over min(C/4, 32) groups and returns (x + h) * s with
// s = 1/sqrt(2); its NIN weights reach this function as 1x1-conv aliases whose proj_out weight/bias are NOT pre-scaled
static Tensor attn_block(Builder& bd, const std::string& pfx, const Tensor& x, int groups = 32, float s_out = 1.0f) {
    pf_engine* e = bd.e; const int B = bd.B, H = x.H, Wd = x.W, C = x.C, HW = H * Wd;
    // qkv = 1x1 convs of GroupNorm(x) (no activation), stacked: [B][HW][3C]
    std::string key = pfx + "qkv";
    if (!e->dev.count(key + ".w")) {
        HostTensor cat; cat.shape = {3 * C, C, 1, 1}; cat.data.reserve((size_t)3 * C * C);
        std::vector<float> bias;
        for (const char* nm : {"attn_q.", "attn_k.", "attn_v."}) {
            const auto& w = W(e, pfx + nm + "weight").data; cat.data.insert(cat.data.end(), w.begin(), w.end());
            const auto& bb = W(e, pfx + nm + "bias").data; bias.insert(bias.end(), bb.begin(), bb.end());
        }
        e->host[key + ".w"] = cat; e->host[key + ".w"].loaded = true;
        upload(e, key + ".b", bias);
        e->dev[key + ".w"] = packed_conv(e, key + ".w", 0, C);
    }
    static const bool fuse_env = !(getenv("PNPFLOW_HIP_FUSED_ATTN") && atoi(getenv("PNPFLOW_HIP_FUSED_ATTN")) == 0);
    static const bool fold_env = !(getenv("PNPFLOW_HIP_ATTN_FOLD") && atoi(getenv("PNPFLOW_HIP_ATTN_FOLD")) == 0);      // test-only A/B switch
    const bool fused = fuse_env && e->precision != 0 && !bd.plan->retain && attn_fused_supported(HW, C);
    // Round 6: proj_out folded into the value projection.  out = x + proj_out(P v) and nothing non-linear sits between the two products, so
    // proj_out(P v) = P (v Wp^T) + bp:  v' = GroupNorm(x) (Wp Wv)^T + Wp bv  replaces v in the stacked q,k,v conv (weights merged once on the host, in
    // double), the fused core adds bp and x in its epilogue and sums the result's GroupNorm statistics - the proj_out launch (and the round trip of P v
    // through HBM) is gone.  Exact algebra: the two forms differ by fp32 rounding.  The SelfAttention of the OT U-Net only (s_out = 1); the retained forward of
    // the VJP keeps the reference's structure (its backward walks it).
    const bool fold = fused && fold_env && s_out == 1.0f;
    static const bool pack_env = !(getenv("PNPFLOW_HIP_ATTN_PACK") && atoi(getenv("PNPFLOW_HIP_ATTN_PACK")) == 0);      // test-only A/B switch
    const bool kv_packed = fused && pack_env;
    if (fold) {
        key = pfx + "qkvf";
        if (!e->dev.count(key + ".w")) {
            const auto& wv = W(e, pfx + "attn_v.weight").data; const auto& bv = W(e, pfx + "attn_v.bias").data;
            const auto& wp = W(e, pfx + "proj_out.weight").data;
            HostTensor cat; cat.shape = {3 * C, C, 1, 1}; cat.data.reserve((size_t)3 * C * C);
            std::vector<float> bias;
            for (const char* nm : {"attn_q.", "attn_k."}) {
                const auto& w = W(e, pfx + nm + "weight").data; cat.data.insert(cat.data.end(), w.begin(), w.end());
                const auto& bb = W(e, pfx + nm + "bias").data; bias.insert(bias.end(), bb.begin(), bb.end());
            }
            std::vector<double> row((size_t)C);
            for (int o = 0; o < C; ++o) {
                std::fill(row.begin(), row.end(), 0.0);
                double bsum = 0.0;
                for (int m = 0; m < C; ++m) {
                    const double a = (double)wp[(size_t)o * C + m];
                    const float* wr = wv.data() + (size_t)m * C;
                    for (int i = 0; i < C; ++i) row[i] += a * (double)wr[i];
                    bsum += a * (double)bv[m];
                }
                for (int i = 0; i < C; ++i) cat.data.push_back((float)row[i]);
                bias.push_back((float)bsum);
            }
            e->host[key + ".w"] = cat; e->host[key + ".w"].loaded = true;
            upload(e, key + ".b", bias);
            e->dev[key + ".w"] = packed_conv(e, key + ".w", 0, C);
        }
    }
    Tensor qkv = bd.make(3 * C, H, Wd, false);
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, qkv);
        add_seg(p, x, 1, 1, 0);
        fill_packed_seg(p.seg[0], e->dev.at(key + ".w"), 1, 3 * C);
        p.seg[0].w16 = packed_conv16(e, key + ".w", 0, C);
        p.gn_C = C; p.gn_cpg = C / groups;
        p.gamma = upload(e, pfx + "norm.weight", W(e, pfx + "norm.weight").data);
        p.beta = upload(e, pfx + "norm.bias", W(e, pfx + "norm.bias").data);
        p.addvec = e->dev.at(key + ".b"); p.addvec_bs = 0;
        // the fused core's keys and values leave this conv as packed fp16 (hi, lo) pairs: split once per element here instead of once per 32-query tile there
        p.pack_from_p1 = kv_packed ? C + 1 : 0;
        push_conv(bd, p);
    }
    if (fold) {
        Tensor out = bd.make(C, H, Wd, true);
        Op op{}; op.kind = OP_ATTN;
        op.ap.qkv = qkv.p; op.ap.out = out.p; op.ap.B = B; op.ap.T = HW; op.ap.C = C; op.ap.scale = 1.0f / sqrtf((float)C); op.ap.kv_packed = kv_packed ? 1 : 0;
        op.ap.bias = upload(e, pfx + "proj_out.bias", W(e, pfx + "proj_out.bias").data); op.ap.residual = x.p;
        // the result's GroupNorm statistics: the core stores the fp32 partial sums of its 32-query tiles (plain stores), a micro-launch adds them per image
        // in tile order (deterministic; fp64 atomics from the tiles: +30 us per launch, a pass over the tensor itself: 14 us)
        float* part = bd.acquire((size_t)B * (HW / 32) * C * 2);
        op.ap.stats_part = part;
        op.flops = (size_t)4 * B * HW * HW * C;
        bd.plan->ops.push_back(op);
        { Op st{}; st.kind = OP_STATS; st.P[0] = part; st.O = out.stats; st.I[0] = HW / 32; st.I[1] = C; bd.plan->ops.push_back(st); }
        bd.recycle(part);
        bd.release(qkv.p);
        return out;
    }
    Tensor S{}, o = bd.make(C, H, Wd, false);
    if (fused) {
        // S = q k^T / sqrt(C), softmax, O = P v in one launch (attention.hip); the exact-fp32 mode and the retained
        // forward of the VJP (whose backward reads P) keep the three-launch path below
        Op op{}; op.kind = OP_ATTN;
        op.ap.qkv = qkv.p; op.ap.out = o.p; op.ap.B = B; op.ap.T = HW; op.ap.C = C; op.ap.scale = 1.0f / sqrtf((float)C); op.ap.kv_packed = kv_packed ? 1 : 0;
        op.flops = (size_t)4 * B * HW * HW * C;
        bd.plan->ops.push_back(op);
    } else {
    // S[b][i][j] = C^-1/2 * sum_c q[i][c] k[j][c]      (a 1-tap "conv" whose weights are k)
    S = bd.make(HW, H, Wd, false);
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, S);
        ConvSeg& s = p.seg[p.nseg++];
        s.src = qkv.p; s.C = C; s.cstride = 3 * C; s.coff = 0; s.xform = 0; s.taps = 1; s.stats = nullptr;
        s.w = qkv.p + C; s.w_mode = 1; s.w_bs = (int64_t)HW * 3 * C; s.w_cs = 0; s.w_ts = 0; s.w_ns = 3 * C; s.w_ks = 1;
        p.out_scale = 1.0f / sqrtf((float)C);
        push_conv(bd, p);
    }
    { Op op{}; op.kind = OP_SOFTMAX; op.sm = S.p; op.sm_rows = (int64_t)B * HW; op.sm_cols = HW; bd.plan->ops.push_back(op); }
    // O[b][i][c] = sum_j A[i][j] v[j][c]
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, o);
        ConvSeg& s = p.seg[p.nseg++];
        s.src = S.p; s.C = HW; s.cstride = HW; s.coff = 0; s.xform = 0; s.taps = 1; s.stats = nullptr;
        s.w = qkv.p + 2 * C; s.w_mode = 1; s.w_bs = (int64_t)HW * 3 * C; s.w_ks = 3 * C; s.w_cs = 0; s.w_ts = 0; s.w_ns = 1;
        push_conv(bd, p);
    }
    }
    // out = x + proj_out(O)
    Tensor out = bd.make(C, H, Wd, true);
    {
        ConvParams p = base_params(B, H, Wd, H, Wd, out);
        add_seg(p, o, 0, 1, 0);
        fill_packed_seg(p.seg[0], packed_conv(e, pfx + "proj_out.weight", 0, C), 1, C);
        p.seg[0].w16 = packed_conv16(e, pfx + "proj_out.weight", 0, C);
        if (s_out == 1.0f) {
            p.addvec = upload(e, pfx + "proj_out.bias", W(e, pfx + "proj_out.bias").data);
        } else {
            std::vector<float> bs = W(e, pfx + "proj_out.bias").data;
            for (auto& v : bs) v *= s_out;
            p.addvec = upload(e, pfx + "proj_out.bias*s", bs);
            p.out_scale = s_out; p.res_scale = s_out;
        }
        p.addvec_bs = 0;
        p.residual = x.p; p.res_cstride = C;
        push_conv(bd, p);
    }
    bd.release(qkv.p); if (S.p) bd.release(S.p); bd.release(o.p);
    if (bd.plan->retain) {
        TapeRec tr; tr.kind = TP_ATTN; tr.pfx = pfx; tr.in0 = x; tr.qkv = qkv; tr.S = S; tr.o = o; tr.out = out; tr.groups = groups; tr.s = s_out;
        bd.plan->tape.push_back(tr);
    }
    return out;
}

// Phase form of Upsample (models.py:70-91: nearest x2, then a 3x3 conv): output pixel (2i + dy, 2j + dx) sees the upsampled rows 2i + dy - 1 ..
// 2i + dy + 1, i.e. source rows {i-1, i, i} for dy = 0 and {i, i, i+1} for dy = 1 (columns alike) - a 2 x 2 conv of the SOURCE image per phase whose
// tap (ty, tx) carries the sum of the 3 x 3 weights that read the same source pixel: rows R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}.
// 16 instead of 36 multiply-adds per source pixel.  Host tensor [phase * Cout + oc][c][ty * 2 + tx] (sums in double, rounded once to fp32), registered
// under `wname + "#phase"` so that the ordinary packers (packed_conv16 / packed_conv16h) build its images.
static std::string phase_weight(pf_engine* e, const std::string& wname) {
    const std::string key = wname + "#phase";
    if (e->host.count(key)) return key;
    const HostTensor& t = W(e, wname);
    const int O = (int)t.shape[0], I = (int)t.shape[1];
    HostTensor ph; ph.shape = {4 * O, I, 2, 2}; ph.data.assign((size_t)4 * O * I * 4, 0.f); ph.loaded = true;
    static const int R0[2][2] = {{0, 1}, {0, 2}}, R1[2][2] = {{1, 3}, {2, 3}};      // [d][t] -> first / one-past-last 3x3 index:  d = 0: {0}, {1,2};  d = 1: {0,1}, {2}
    for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx)
        for (int oc = 0; oc < O; ++oc) for (int c = 0; c < I; ++c)
            for (int ty = 0; ty < 2; ++ty) for (int tx = 0; tx < 2; ++tx) {
                double sum = 0.0;
                for (int ky = R0[dy][ty]; ky < R1[dy][ty]; ++ky)
                    for (int kx = R0[dx][tx]; kx < R1[dx][tx]; ++kx) sum += (double)t.data[((size_t)oc * I + c) * 9 + ky * 3 + kx];
                ph.data[(((size_t)(dy * 2 + dx) * O + oc) * I + c) * 4 + ty * 2 + tx] = (float)sum;
            }
    e->host[key] = std::move(ph);
    return key;
}

// the upsampling conv on conv_dma's UP = 2 form, when the shape qualifies (split-fp16 modes only; the exact-fp32 mode and small / ragged sizes keep
// the 9-tap form on the upsampled view)
static bool push_conv_up_phase(Builder& bd, const ConvParams& p9, const std::string& wname) {
    pf_engine* e = bd.e;
    if (e->precision == 0) return false;
    const int terms = e->precision == 2 ? 1 : 3;
    ConvParams q = p9;
    ConvSeg& sg = q.seg[0];
    if (!conv_dma_phase_shape_ok(q, terms)) return false;      // (before any weight image is built)
    const std::string key = phase_weight(e, wname);
    sg.taps = 4; sg.w = nullptr; sg.w16 = packed_conv16(e, key, 0, sg.C);
    if (!sg.w16) return false;
    ConvParams p = with_coef(bd, q, bd.plan->ops);
    Op op{}; op.kind = OP_CONV;
    op.dma = attach_dma(bd, p, 1, 2, bd.plan->ops) ? 1 : 0;
    if (!op.dma) { bd.ok = false; e->err = "phase upsampling conv: conv_dma refused a launch it had accepted"; return false; }
    op.cp = p; op.stride = 1; op.up = 2; op.flops = conv_flops(p);
    bd.plan->gemm_flops += op.flops;
    bd.plan->ops.push_back(op);
    return true;
}

static Tensor resample_conv(Builder& bd, const std::string& pfx, const Tensor& x, bool down) {
    pf_engine* e = bd.e; const int B = bd.B;
    const int H = down ? x.H / 2 : x.H * 2, Wd = down ? x.W / 2 : x.W * 2;
    Tensor out = bd.make(x.C, H, Wd, true);
    ConvParams p = base_params(B, H, Wd, x.H, x.W, out);
    add_seg(p, x, 0, 9, 0);
    p.addvec = upload(e, pfx + "bias", W(e, pfx + "bias").data); p.addvec_bs = 0;
    if (down || !push_conv_up_phase(bd, p, pfx + "weight")) {
        fill_packed_seg(p.seg[0], packed_conv(e, pfx + "weight", 0, x.C), 9, x.C);
        p.seg[0].w16 = packed_conv16(e, pfx + "weight", 0, x.C);
        push_conv(bd, p, down ? 2 : 1, down ? 0 : 1);
    }
    if (bd.plan->retain) {
        TapeRec tr; tr.kind = down ? TP_DOWN : TP_UP; tr.pfx = pfx; tr.in0 = x; tr.out = out;
        bd.plan->tape.push_back(tr);
    }
    return out;
}

static void fix_stats(Op& op, double* slab) {
    auto fx = [&](const double*& p) { if (p) p = (const double*)((char*)slab + ((uintptr_t)p - 1)); };
    auto fxm = [&](double*& p) { if (p) p = (double*)((char*)slab + ((uintptr_t)p - 1)); };
    if (op.kind == OP_CONV) { for (int i = 0; i < op.cp.nseg; ++i) fx(op.cp.seg[i].stats); fxm(op.cp.stats_out); fxm(op.cp.gnb_sum); if (op.use_pp) op.ppp.stats_out = op.cp.stats_out; }
    if (op.kind == OP_GN_COEF) for (int i = 0; i < op.gp.nseg; ++i) fx(op.gp.st[i]);
    if (op.kind == OP_STATS) { double* d = (double*)op.O; fxm(d); op.O = d; }
    if (op.kind == OP_NX_FIR) fxm(op.fp.stats_raw);
    if (op.kind == OP_END) fx(op.ep.stats);
    if (op.kind == OP_BEGIN) fxm(op.ep.stats_out);
    if (op.kind == OP_GN_FWD_COEF) { const double* a = (const double*)op.P[0]; const double* b2 = (const double*)op.P[1]; fx(a); fx(b2); op.P[0] = a; op.P[1] = b2; }
    if (op.kind == OP_GN_BWD_PRE || op.kind == OP_GN_BWD_COEF) { const double* a = (const double*)op.P[6]; fx(a); op.P[6] = a; }
}

static int build_backward(pf_engine* e, Plan* plan, struct Builder& bd);
static int nx_build_backward(pf_engine* e, Plan* plan, struct Builder& bd);

// the OT U-Net's forward walk (models.py:442-495): op 1 = time embedding, ...
static int unet_walk(pf_engine* e, Builder& bd, Plan* plan) {
    const int B = bd.B;
    const pf_unet_cfg& c = e->cfg;
    const int H0 = c.input_height, ch = c.ch, tch = 4 * ch;
    float* temb_vec = bd.acquire((size_t)B * e->temb_total);
    {
        // stacked temb projections; bias = temb_proj.bias + conv1.bias
        if (!e->dev.count("temb.wp")) {
            std::vector<float> wp, bp;
            for (auto& pr : e->res_order) {
                const auto& w = W(e, pr.first + "temb_proj.weight").data; wp.insert(wp.end(), w.begin(), w.end());
                const auto& b1 = W(e, pr.first + "temb_proj.bias").data; const auto& b2 = W(e, pr.first + "conv1.bias").data;
                for (size_t i = 0; i < b1.size(); ++i) bp.push_back(b1[i] + b2[i]);
            }
            upload(e, "temb.wp", wp); upload(e, "temb.bp", bp);
        }
        Op op{}; op.kind = OP_TEMB;
        op.tp.w0 = upload(e, "temb_net.main.0.weight", W(e, "temb_net.main.0.weight").data);
        op.tp.b0 = upload(e, "temb_net.main.0.bias", W(e, "temb_net.main.0.bias").data);
        op.tp.w1 = upload(e, "temb_net.main.2.weight", W(e, "temb_net.main.2.weight").data);
        op.tp.b1 = upload(e, "temb_net.main.2.bias", W(e, "temb_net.main.2.bias").data);
        op.tp.wp = e->dev.at("temb.wp"); op.tp.bp = e->dev.at("temb.bp");
        op.tp.out = temb_vec; op.tp.B = B; op.tp.ch = ch; op.tp.total_out = e->temb_total;
        (void)tch;
        plan->ops.push_back(op);
    }
    // begin conv (models.py:451) + stand-alone channel statistics of its output
    std::vector<Tensor> hs;
    {
        Tensor t0 = bd.make(ch, H0, H0, true);
        const HostTensor& w = W(e, "begin_conv.weight");   // [ch][Cimg][3][3] -> [tap][ci][ch]
        const int ci_n = c.input_channels;
        std::vector<float> wp((size_t)9 * ci_n * ch);
        for (int n = 0; n < ch; ++n) for (int ci = 0; ci < ci_n; ++ci) for (int tap = 0; tap < 9; ++tap)
            wp[((size_t)tap * ci_n + ci) * ch + n] = w.data[((size_t)n * ci_n + ci) * 9 + tap];
        Op op{}; op.kind = OP_BEGIN;
        op.ep.out = t0.p; op.ep.w = upload(e, "begin_conv.packed", wp);
        op.ep.bias = upload(e, "begin_conv.bias", W(e, "begin_conv.bias").data);
        op.ep.B = B; op.ep.H = H0; op.ep.W = H0; op.ep.Cimg = ci_n; op.ep.C = ch;
        op.ep.stats_out = t0.stats;              // (sum, sumsq) per channel for the first GroupNorm, reduced in the same kernel
        if (ch == 32 && e->precision != 0 && (ci_n == 1 || ci_n == 3)) {
            // begin_conv2_kernel's B fragments: MFMA k of (step s, half hh, j) = 16 s + 8 hh + j = input channel * 9 + tap (zero beyond 9 Cimg),
            // column n = output channel, x 2^8 and split like every packed conv weight
            std::vector<_Float16> wm((size_t)2 * 2 * 64 * 8, (_Float16)0.f);
            for (int sk = 0; sk < 2; ++sk) for (int ln = 0; ln < 64; ++ln) for (int j = 0; j < 8; ++j) {
                const int n = ln & 31, hh = ln >> 5, kk = 16 * sk + 8 * hh + j;
                if (kk >= 9 * ci_n) continue;
                const int ci = kk / 9, tap = kk % 9;
                const float wv = w.data[((size_t)n * ci_n + ci) * 9 + tap] * 256.0f;
                const _Float16 hi = (_Float16)wv, lo = (_Float16)(wv - (float)hi);
                wm[(((size_t)sk * 2 + 0) * 64 + ln) * 8 + j] = hi; wm[(((size_t)sk * 2 + 1) * 64 + ln) * 8 + j] = lo;
            }
            std::vector<float> raw(wm.size() / 2);
            memcpy(raw.data(), wm.data(), wm.size() * sizeof(_Float16));
            op.ep.w16 = upload(e, "begin_conv.mfma16", raw);
        }
        plan->ops.push_back(op);
        hs.push_back(t0);
        plan->t_begin = t0;
        plan->taps.push_back({"begin_conv", t0});
    }
    char nm[64];
    // down path (models.py:452-467)
    for (size_t lvl = 0; lvl < e->down.size(); ++lvl) {
        const LevelDesc& L = e->down[lvl];
        for (size_t blk = 0; blk < L.blocks.size(); ++blk) {
            Tensor h = res_block(bd, L.blocks[blk], hs.back(), nullptr, temb_vec);
            if (L.blocks[blk].attn) { Tensor a = attn_block(bd, L.blocks[blk].attn_prefix, h); bd.release(h.p); h = a; }
            hs.push_back(h);
            snprintf(nm, sizeof nm, "down%zu_%zu", lvl, blk); plan->taps.push_back({nm, h});
        }
        if (!L.resample.empty()) {
            hs.push_back(resample_conv(bd, L.resample, hs.back(), true));
            snprintf(nm, sizeof nm, "downsample%zu", lvl); plan->taps.push_back({nm, hs.back()});
        }
    }
    // middle (models.py:470-471)
    Tensor h = hs.back();
    bool h_owned = false;   // hs tensors are released when popped
    {
        Tensor a = res_block(bd, e->mid0, h, nullptr, temb_vec);
        Tensor b2 = attn_block(bd, e->mid_attn, a); bd.release(a.p);
        Tensor c2 = res_block(bd, e->mid2, b2, nullptr, temb_vec); bd.release(b2.p);
        h = c2; h_owned = true;
        plan->taps.push_back({"mid", h});
    }
    // up path (models.py:474-488)
    for (size_t idx = 0; idx < e->up.size(); ++idx) {
        const LevelDesc& L = e->up[idx];
        const int lvl = (int)e->up.size() - 1 - (int)idx;
        for (size_t blk = 0; blk < L.blocks.size(); ++blk) {
            Tensor skip = hs.back(); hs.pop_back();
            Tensor o = res_block(bd, L.blocks[blk], h, &skip, temb_vec);
            if (h_owned) bd.release(h.p);
            bd.release(skip.p);
            if (L.blocks[blk].attn) { Tensor a = attn_block(bd, L.blocks[blk].attn_prefix, o); bd.release(o.p); o = a; }
            h = o; h_owned = true;
            snprintf(nm, sizeof nm, "up%d_%zu", lvl, blk); plan->taps.push_back({nm, h});
        }
        if (!L.resample.empty()) {
            Tensor o = resample_conv(bd, L.resample, h, false);
            bd.release(h.p); h = o;
            snprintf(nm, sizeof nm, "upsample%d", lvl); plan->taps.push_back({nm, h});
        }
    }
    // end (models.py:492)
    {
        const HostTensor& w = W(e, "end_conv.2.weight");   // [Cimg][ch][3][3] -> [tap][co][ch]
        const int co_n = c.output_channels;
        std::vector<float> wp((size_t)9 * co_n * ch);
        for (int co = 0; co < co_n; ++co) for (int ci = 0; ci < ch; ++ci) for (int tap = 0; tap < 9; ++tap)
            wp[((size_t)tap * co_n + co) * ch + ci] = w.data[((size_t)co * ch + ci) * 9 + tap];
        Op op{}; op.kind = OP_END;
        op.ep.in = h.p; op.ep.w = upload(e, "end_conv.packed", wp);
        op.ep.bias = upload(e, "end_conv.2.bias", W(e, "end_conv.2.bias").data);
        op.ep.B = B; op.ep.H = H0; op.ep.W = H0; op.ep.Cimg = co_n; op.ep.C = ch;
        op.ep.stats = h.stats; op.ep.gamma = upload(e, "end_conv.0.weight", W(e, "end_conv.0.weight").data);
        op.ep.beta = upload(e, "end_conv.0.bias", W(e, "end_conv.0.bias").data);
        op.ep.gn_cpg = ch / 32; op.ep.gn_eps = 1e-6f;
        if (ch == 32 && e->precision != 0) {
            // end_conv2_kernel's B fragments: MFMA k of (step s, half hh, j) = channel 16 hh + 8 s + j, column n = tap * Cimg + co (zero beyond
            // 9 Cimg), x 2^8 and split like every packed conv weight
            std::vector<_Float16> wm((size_t)2 * 2 * 64 * 8, (_Float16)0.f);
            for (int sk = 0; sk < 2; ++sk) for (int ln = 0; ln < 64; ++ln) for (int j = 0; j < 8; ++j) {
                const int n = ln & 31, hh = ln >> 5, cc = 16 * hh + 8 * sk + j;
                if (n >= 9 * co_n) continue;
                const int tap = n / co_n, co = n % co_n;
                const float wv = w.data[((size_t)co * ch + cc) * 9 + tap] * 256.0f;
                const _Float16 hi = (_Float16)wv, lo = (_Float16)(wv - (float)hi);
                wm[(((size_t)sk * 2 + 0) * 64 + ln) * 8 + j] = hi; wm[(((size_t)sk * 2 + 1) * 64 + ln) * 8 + j] = lo;
            }
            std::vector<float> raw(wm.size() / 2);
            memcpy(raw.data(), wm.data(), wm.size() * sizeof(_Float16));
            op.ep.w16 = upload(e, "end_conv.mfma16", raw);
        }
        plan->ops.push_back(op);
        plan->t_last = h;
    }
    return PF_OK;
}

#include "engine_ncsnpp.inc"

static int build_plan(pf_engine* e, int B, bool retain, Plan** out_plan) {
    const int key = (B * 2 + (retain ? 1 : 0)) * 3 + e->precision;      // the precision mode selects kernels at build time (one plan per mode: pf_unet_backward must meet the mode of its retained forward)
    auto it = e->plans.find(key);
    if (it != e->plans.end()) { *out_plan = e->last_plan = it->second.get(); it->second->last_used = ++e->plan_clock; return PF_OK; }
    // bounded cache: a plan owns its activation buffers (GBs at the BASELINE sizes), so the least recently used one is
    // dropped before a ninth is built (never the retained one a pf_unet_backward may still walk, nor the graph's)
    while (e->plans.size() >= 8) {
        auto victim = e->plans.end();
        for (auto jt = e->plans.begin(); jt != e->plans.end(); ++jt) {
            if (jt->second.get() == e->retained_plan || jt->second.get() == e->gkey.plan || jt->second.get() == e->okey.plan) continue;
            if (victim == e->plans.end() || jt->second->last_used < victim->second->last_used) victim = jt;
        }
        if (victim == e->plans.end()) break;
        hipDeviceSynchronize();
        for (void* p : victim->second->allocs) hipFree(p);
        e->bytes -= victim->second->bytes;
        if (e->last_plan == victim->second.get()) e->last_plan = nullptr;
        e->plans.erase(victim);
    }
    auto plan = std::make_unique<Plan>(); plan->B = B; plan->retain = retain; plan->last_used = ++e->plan_clock;
    Builder bd{e, plan.get(), B};
    if (retain) bd.keep = true;
    // op 0: zero the statistics slab (filled in below)
    { Op op{}; op.kind = OP_MEMSET; plan->ops.push_back(op); }
    // a plan that fails to build gives its device memory back (at the BASELINE sizes a half-built plan strands GBs, and
    // pf_engine_memory_bytes would keep counting them)
    auto fail = [&](int rc) { for (void* q : plan->allocs) hipFree(q); e->bytes -= plan->bytes; plan->allocs.clear(); plan->bytes = 0; return rc; };
    if (e->arch == 1) {
        int rc = nx_walk(e, bd, plan.get());
        if (rc != PF_OK) return fail(rc);
    } else {
        int rc = unet_walk(e, bd, plan.get());
        if (rc != PF_OK) return fail(rc);
    }
    if (retain) { int rc = e->arch == 1 ? nx_build_backward(e, plan.get(), bd) : build_backward(e, plan.get(), bd); if (rc != PF_OK) return fail(rc); }
    if (!bd.ok) return fail(PF_ERR_HIP);
    for (auto& kv : e->dev) if (kv.second == nullptr) { e->err = "weight upload failed: " + kv.first; return fail(PF_ERR_HIP); }
    // statistics slab
    void* slab = nullptr;
    if (hipMalloc(&slab, std::max<size_t>(bd.stats_bytes, 256)) != hipSuccess) { e->err = "hipMalloc failed (stats)"; return fail(PF_ERR_HIP); }
    poison(slab, std::max<size_t>(bd.stats_bytes, 256));
    plan->allocs.push_back(slab); plan->bytes += (int64_t)std::max<size_t>(bd.stats_bytes, 256); e->bytes += (int64_t)std::max<size_t>(bd.stats_bytes, 256);
    for (auto& op : plan->ops) fix_stats(op, (double*)slab);
    for (auto& op : plan->bops) fix_stats(op, (double*)slab);
    for (auto& tp : plan->taps) if (tp.t.stats) tp.t.stats = (double*)((char*)slab + ((uintptr_t)tp.t.stats - 1));
    plan->ops[0].ptr = slab; plan->ops[0].bytes = bd.stats_bytes;
    if (!plan->bops.empty()) plan->bops[0].ptr = (char*)slab + ((uintptr_t)plan->bops[0].ptr - 1);
    *out_plan = plan.get();
    e->last_plan = plan.get();
    e->plans[key] = std::move(plan);
    return PF_OK;
}

// --------------------------------------------------------------------------------------
// backward plan: J^T vec with respect to the network input (time embedding path has no input
// gradient).  Walks the forward tape in reverse; every dense step reuses conv_mfma_kernel with
// transposed (and spatially flipped) weight repacks.
// --------------------------------------------------------------------------------------
// OIHW weight, input channels [lo,hi)  ->  adjoint conv weight  W'[ci][co][ky][kx] = W[co][lo+ci][K-1-ky][K-1-kx]
static float* packed_conv_T(pf_engine* e, const std::string& wname, int lo, int hi) {
    const std::string key = wname + "#T" + std::to_string(lo) + ":" + std::to_string(hi);
    if (!e->host.count(key)) {
        const HostTensor& t = W(e, wname);
        const int O = (int)t.shape[0], I = (int)t.shape[1], K = (int)t.shape[2], kk = K * K, C = hi - lo;
        HostTensor tt; tt.shape = {C, O, K, K}; tt.data.resize((size_t)C * O * kk); tt.loaded = true;
        for (int ci = 0; ci < C; ++ci) for (int co = 0; co < O; ++co) for (int tap = 0; tap < kk; ++tap)
            tt.data[((size_t)ci * O + co) * kk + tap] = t.data[((size_t)co * I + lo + ci) * kk + (kk - 1 - tap)];
        e->host[key] = std::move(tt);
    }
    return packed_conv(e, key, 0, (int)e->host.at(key).shape[1]);
}

static const void* packed_conv16_T(pf_engine* e, const std::string& wname, int lo, int hi) {
    packed_conv_T(e, wname, lo, hi);      // creates the transposed host tensor
    const std::string key = wname + "#T" + std::to_string(lo) + ":" + std::to_string(hi);
    return packed_conv16(e, key, 0, (int)e->host.at(key).shape[1]);
}

struct GradEntry { Tensor t; bool has = false; };

struct BwdCtx {
    pf_engine* e; Plan* plan; Builder* bd; int B;
    std::map<float*, GradEntry> grads;
    GradEntry& G(const Tensor& fwd) {
        auto it = grads.find(fwd.p);
        if (it == grads.end()) {
            GradEntry g; g.t.C = fwd.C; g.t.H = fwd.H; g.t.W = fwd.W; g.t.p = bd->acquire((size_t)B * fwd.H * fwd.W * fwd.C);
            it = grads.emplace(fwd.p, g).first;
        }
        return it->second;
    }
    Tensor tmp(int C, int H, int W) { Tensor t; t.C = C; t.H = H; t.W = W; t.p = bd->acquire((size_t)B * H * W * C); return t; }
    float* fvec(size_t n) { return bd->acquire(n); }
    double* dsum(size_t n_doubles) { double* p = (double*)(uintptr_t)(bd->stats_bytes + 1); bd->stats_bytes += n_doubles * sizeof(double); return p; }
    void push(const Op& op) { plan->bops.push_back(op); }
    // dst (=|+=) conv(src; weights)   (raw input, no bias)
    void conv_to(ConvParams p, GradEntry& dst, int stride = 1, int up = 0) {
        p.out = dst.t.p; p.out_cstride = dst.t.C;
        if (dst.has) { p.residual = dst.t.p; p.res_cstride = dst.t.C; }
        dst.has = true;
        Op op{}; op.kind = OP_CONV; op.cp = p; op.stride = stride; op.up = up; op.flops = conv_flops(p);
        plan->bwd_flops += op.flops;
        push(op);
    }
    void conv_plain(const ConvParams& p, int stride = 1, int up = 0) {
        Op op{}; op.kind = OP_CONV; op.cp = p; op.stride = stride; op.up = up; op.flops = conv_flops(p);
        plan->bwd_flops += op.flops;
        push(op);
    }
};

static ConvParams bwd_params(int B, int H, int W, int Hs, int Ws, int Cout) {
    ConvParams p{};
    p.B = B; p.H = H; p.W = W; p.Hs = Hs; p.Ws = Ws; p.Cout = Cout; p.out_scale = 1.0f; p.res_scale = 1.0f; p.gn_eps = 1e-6f;
    return p;
}
static void raw_seg(ConvParams& p, const float* src, int C, int cstride, int taps, const float* w, const void* w16 = nullptr) {
    ConvSeg& s = p.seg[p.nseg++];
    s.src = src; s.C = C; s.cstride = cstride; s.coff = 0; s.xform = 0; s.taps = taps; s.gn_off = 0; s.stats = nullptr;
    s.w = w; s.w_mode = 0; s.w_bs = 0; s.w_cs = 0; s.w_ts = 0; s.w_ns = 0; s.w_ks = 0; s.w16 = w16;
}
static void gen_seg(ConvParams& p, const float* src, int C, int cstride, const float* w, int64_t w_bs, int64_t w_ns, int64_t w_ks) {
    ConvSeg& s = p.seg[p.nseg++];
    s.src = src; s.C = C; s.cstride = cstride; s.coff = 0; s.xform = 0; s.taps = 1; s.gn_off = 0; s.stats = nullptr;
    s.w = w; s.w_mode = 1; s.w_bs = w_bs; s.w_cs = 0; s.w_ts = 0; s.w_ns = w_ns; s.w_ks = w_ks;
}

// GroupNorm(+SiLU) backward over a (possibly concatenated) input, in three steps so that the FIRST stage (dyhat = da*act'(u)*gamma
// and its two per-channel sums) can be fused into the epilogue of the adjoint conv that produces da (ConvParams::gnb_*):
//   gn_bwd_begin   forward coefficients mu / rstd of the GroupNorm, the sum slab
//   gn_bwd_fuse    fills the gnb_* fields of the conv that writes source j's da (split-fp16 kernel, stride 1): no PRE pass for j
//   gn_bwd_finish  PRE pass for the sources that were not fused, group means, second stage
// srcs[j] = forward input tensor j, gtmp[j] = gradient w.r.t. the activated/normalised tensor (overwritten),
// dst[j] = where dx goes, add[j] = optional extra addend (identity-shortcut gradient).
struct GnBwd {
    std::vector<Tensor> srcs; int Ct = 0, cpg = 0, HW = 0; bool silu = false;
    bool retained = false;            // mu / rs are the retained forward's (Plan::gn_ret): not this stage's to recycle
    float *mu = nullptr, *rs = nullptr, *m1 = nullptr, *m2 = nullptr; double* bsum = nullptr;
    const float *gamma = nullptr, *beta = nullptr;
    std::vector<bool> fused;
};

static GnBwd gn_bwd_begin(BwdCtx& c, const std::vector<Tensor>& srcs, const std::string& norm_prefix, bool silu, int groups = 32) {
    pf_engine* e = c.e; const int B = c.B;
    GnBwd g; g.srcs = srcs; g.silu = silu; g.HW = srcs[0].H * srcs[0].W;
    for (auto& t : srcs) g.Ct += t.C;
    g.cpg = g.Ct / groups; g.fused.assign(srcs.size(), false);
    g.m1 = c.fvec((size_t)B * g.Ct); g.m2 = c.fvec((size_t)B * g.Ct);
    g.bsum = c.dsum((size_t)B * g.Ct * 2);
    g.gamma = upload(e, norm_prefix + "weight", W(e, norm_prefix + "weight").data);
    g.beta = upload(e, norm_prefix + "bias", W(e, norm_prefix + "bias").data);
    auto rt = c.plan->gn_ret.find(g.gamma);
    if (rt != c.plan->gn_ret.end()) {      // the forward launch that applied this GroupNorm left its (mean, rstd) behind
        g.mu = rt->second.first; g.rs = rt->second.second; g.retained = true;
        return g;
    }
    g.mu = c.fvec((size_t)B * g.Ct); g.rs = c.fvec((size_t)B * g.Ct);
    Op op{}; op.kind = OP_GN_FWD_COEF; op.P[0] = srcs[0].stats; op.P[1] = srcs.size() > 1 ? srcs[1].stats : nullptr; op.P[2] = g.mu; op.P[3] = g.rs;
    op.I[0] = srcs[0].C; op.I[1] = srcs.size() > 1 ? srcs[1].C : 0; op.I[2] = g.cpg; op.I[3] = g.HW; c.push(op);
    return g;
}

// returns true (and marks source j fused) when the conv described by p can carry the first stage in its epilogue
static bool gn_bwd_fuse(BwdCtx& c, GnBwd& g, size_t j, ConvParams& p, int stride = 1, int up = 0) {
    static const bool enabled = !(getenv("PNPFLOW_HIP_GNB_FUSE") && atoi(getenv("PNPFLOW_HIP_GNB_FUSE")) == 0);
    bool ok16 = c.e->precision != 0 && enabled && stride == 1 && up == 0 && p.residual == nullptr && p.stats_out == nullptr && p.Cout % 4 == 0;
    for (int i = 0; i < p.nseg; ++i) ok16 &= p.seg[i].w_mode == 0 && p.seg[i].w16 != nullptr;
    if (!ok16) return false;
    int coff = 0; for (size_t k = 0; k < j; ++k) coff += g.srcs[k].C;
    p.gnb_x = g.srcs[j].p; p.gnb_xstride = g.srcs[j].C; p.gnb_mu = g.mu; p.gnb_rs = g.rs; p.gnb_gamma = g.gamma; p.gnb_beta = g.beta;
    p.gnb_Ct = g.Ct; p.gnb_coff = coff; p.gnb_silu = g.silu ? 1 : 0; p.gnb_sum = g.bsum;
    g.fused[j] = true;
    return true;
}

static void gn_bwd_finish(BwdCtx& c, GnBwd& g, const std::vector<Tensor>& gtmp, const std::vector<GradEntry*>& dst,
                          const std::vector<const float*>& add, float add_scale = 1.0f) {
    int coff = 0;
    for (size_t j = 0; j < g.srcs.size(); ++j) {
        if (!g.fused[j]) {
            Op op{}; op.kind = OP_GN_BWD_PRE; op.O = gtmp[j].p; op.P[0] = g.srcs[j].p; op.P[1] = g.mu; op.P[2] = g.rs; op.P[3] = g.gamma; op.P[4] = g.beta;
            op.P[6] = g.bsum; op.I[0] = g.HW; op.I[1] = g.srcs[j].C; op.I[2] = coff; op.I[3] = g.Ct; op.I[4] = g.silu ? 1 : 0; c.push(op);
        }
        coff += g.srcs[j].C;
    }
    { Op op{}; op.kind = OP_GN_BWD_COEF; op.P[6] = g.bsum; op.P[1] = g.m1; op.P[2] = g.m2; op.I[0] = g.Ct; op.I[1] = g.cpg; op.I[2] = g.HW; c.push(op); }
    coff = 0;
    for (size_t j = 0; j < g.srcs.size(); ++j) {
        Op op{}; op.kind = OP_GN_BWD_POST; op.P[0] = gtmp[j].p; op.P[1] = g.srcs[j].p; op.P[2] = g.mu; op.P[3] = g.rs; op.P[4] = g.m1; op.P[5] = g.m2;
        op.P[7] = add[j]; op.F = add_scale; op.O = dst[j]->t.p; op.I[0] = g.HW; op.I[1] = g.srcs[j].C; op.I[2] = coff; op.I[3] = g.Ct; op.I[4] = dst[j]->has ? 1 : 0;
        dst[j]->has = true; c.push(op);
        coff += g.srcs[j].C;
    }
    if (!g.retained) { c.bd->recycle(g.mu); c.bd->recycle(g.rs); }
    c.bd->recycle(g.m1); c.bd->recycle(g.m2);
}

static void gn_backward(BwdCtx& c, const std::vector<Tensor>& srcs, const std::vector<Tensor>& gtmp, const std::vector<GradEntry*>& dst,
                        const std::vector<const float*>& add, const std::string& norm_prefix, bool silu) {
    GnBwd g = gn_bwd_begin(c, srcs, norm_prefix, silu);
    gn_bwd_finish(c, g, gtmp, dst, add);
}

// SelfAttention / AttnBlockpp backward (models.py:145-162; layerspp.py:76-94: out = (x + proj(attn(norm(x)))) * s)
static void bwd_attn(BwdCtx& c, const TapeRec& tr, GradEntry& gout) {
    pf_engine* e = c.e; Builder& bd = *c.bd; const int B = c.B;
    const int H = tr.out.H, Wd = tr.out.W;
    const Tensor& x = tr.in0; const int C = x.C, HW = H * Wd;
    const float scale = 1.0f / sqrtf((float)C);
    // d(o) = dout . Wproj
    Tensor d_o = c.tmp(C, H, Wd);
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, C);
      raw_seg(p, gout.t.p, C, C, 1, packed_conv_T(e, tr.pfx + "proj_out.weight", 0, C), packed_conv16_T(e, tr.pfx + "proj_out.weight", 0, C)); p.out = d_o.p; p.out_cstride = C;
      p.out_scale = tr.s; c.conv_plain(p); }
    // dA[i][j] = sum_c d_o[i][c] v[j][c]
    Tensor dA = c.tmp(HW, H, Wd);
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, HW);
      gen_seg(p, d_o.p, C, C, tr.qkv.p + 2 * C, (int64_t)HW * 3 * C, 3 * C, 1); p.out = dA.p; p.out_cstride = HW; c.conv_plain(p); }
    Tensor dqkv = c.tmp(3 * C, H, Wd);
    // dv[j][c] = sum_i A[i][j] d_o[i][c]   (A^T as the pixel-major operand)
    Tensor AT = c.tmp(HW, H, Wd);
    { Op op{}; op.kind = OP_TRANSPOSE; op.P[0] = tr.S.p; op.O = AT.p; op.I[0] = HW; op.I[1] = HW; c.push(op); }
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, C);
      gen_seg(p, AT.p, HW, HW, d_o.p, (int64_t)HW * C, 1, C); p.out = dqkv.p + 2 * C; p.out_cstride = 3 * C; c.conv_plain(p); }
    // dS = scale * A .* (dA - rowsum(dA .* A))   in place
    { Op op{}; op.kind = OP_SOFTMAX_BWD; op.P[0] = tr.S.p; op.O = dA.p; op.sm_rows = (int64_t)B * HW; op.sm_cols = HW; op.F = scale; c.push(op); }
    // dq[i][c] = sum_j dS[i][j] k[j][c]
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, C);
      gen_seg(p, dA.p, HW, HW, tr.qkv.p + C, (int64_t)HW * 3 * C, 1, 3 * C); p.out = dqkv.p; p.out_cstride = 3 * C; c.conv_plain(p); }
    // dk[j][c] = sum_i dS[i][j] q[i][c]
    { Op op{}; op.kind = OP_TRANSPOSE; op.P[0] = dA.p; op.O = AT.p; op.I[0] = HW; op.I[1] = HW; c.push(op); }
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, C);
      gen_seg(p, AT.p, HW, HW, tr.qkv.p, (int64_t)HW * 3 * C, 1, 3 * C); p.out = dqkv.p + C; p.out_cstride = 3 * C; c.conv_plain(p); }
    // d(hn) = dqkv . Wqkv ; GroupNorm backward (no activation) ; + identity path
    Tensor dhn = c.tmp(C, H, Wd);
    GnBwd gna = gn_bwd_begin(c, {x}, tr.pfx + "norm.", false, tr.groups);
    { ConvParams p = bwd_params(B, H, Wd, H, Wd, C);
      raw_seg(p, dqkv.p, 3 * C, 3 * C, 1, packed_conv_T(e, tr.pfx + "qkv.w", 0, C), packed_conv16_T(e, tr.pfx + "qkv.w", 0, C)); p.out = dhn.p; p.out_cstride = C;
      gn_bwd_fuse(c, gna, 0, p);
      c.conv_plain(p); }
    gn_bwd_finish(c, gna, {dhn}, {&c.G(x)}, {gout.t.p}, tr.s);
    for (float* q : {d_o.p, dA.p, dqkv.p, AT.p, dhn.p}) bd.recycle(q);
}

static int build_backward(pf_engine* e, Plan* plan, Builder& bd) {
    const int B = plan->B;
    const pf_unet_cfg& cf = e->cfg;
    const int ch = cf.ch, H0 = cf.input_height;
    BwdCtx c{e, plan, &bd, B};
    // op 0: zero the backward sums (range filled in below)
    { Op op{}; op.kind = OP_MEMSET; plan->bops.push_back(op); }
    plan->vec_scaled = bd.acquire((size_t)B * cf.input_channels * cf.input_height * cf.input_height);
    plan->vjp_scale = bd.acquire(64); plan->vjp_amax = reinterpret_cast<unsigned int*>(bd.acquire(64));
    const size_t bwd_lo = bd.stats_bytes;
    // ---- end: v = conv3x3(silu(gn(h)))  (models.py:492).  d(act) = adjoint conv of vec (image -> ch) --------
    {
        const HostTensor& w = W(e, "end_conv.2.weight");   // [Cimg][ch][3][3]
        const int co_n = cf.output_channels;
        std::vector<float> wp((size_t)9 * co_n * ch), zero(ch, 0.f);
        for (int co = 0; co < co_n; ++co) for (int ci = 0; ci < ch; ++ci) for (int tap = 0; tap < 9; ++tap)
            wp[((size_t)tap * co_n + co) * ch + ci] = w.data[((size_t)co * ch + ci) * 9 + (8 - tap)];
        Tensor da = c.tmp(ch, H0, H0);
        Op op{}; op.kind = OP_BEGIN; op.ep.out = da.p; op.ep.w = upload(e, "end_conv.adj", wp); op.ep.bias = upload(e, "zero.ch", zero);
        op.ep.B = B; op.ep.H = H0; op.ep.W = H0; op.ep.Cimg = co_n; op.ep.C = ch; c.push(op);
        gn_backward(c, {plan->t_last}, {da}, {&c.G(plan->t_last)}, {nullptr}, "end_conv.0.", true);
        bd.recycle(da.p);
    }
    // ---- reverse walk -----------------------------------------------------------------------------------
    for (auto it = plan->tape.rbegin(); it != plan->tape.rend(); ++it) {
        const TapeRec& tr = *it;
        GradEntry& gout = c.G(tr.out);
        if (!gout.has) { e->err = "internal: backward reached a tensor without gradient"; return PF_ERR_INVALID; }
        const int H = tr.out.H, Wd = tr.out.W;
        static const bool upadj_env = !(getenv("PNPFLOW_HIP_UPPHASE_BWD") && atoi(getenv("PNPFLOW_HIP_UPPHASE_BWD")) == 0);      // test-only A/B switch
        if (tr.kind == TP_UP && e->precision != 0 && upadj_env && tr.out.C % 16 == 0 && tr.in0.C % 4 == 0) {
            // out = conv(nearest_up(x)) in its phase form (four 2 x 2 convs of x, see phase_weight): the adjoint is, per output phase (dy, dx), a 2 x 2 conv of the
            // strided view g[2i + dy][2j + dx] of the gradient into dx - 16 multiply-adds per source pixel and channel pair instead of 36 + a sum-pool pass
            // over the fine-resolution tensor.  Source (i, j) is read by phase (dy, dx) at output index i - ty - dy + 1: with u = 1 - ty the window covers rows
            // i + u - dy = patch rows u + (1 - dy) of the 3 x 3 neighbourhood, and tap u carries the forward tap 1 - u.  Two launches of two K-segments each (the
            // second accumulates); the views share the geometry: pixel pitch 2 C_out floats, row pitch two fine rows.
            const int Cy = tr.out.C, Cx = tr.in0.C, Hs = tr.in0.H, Ws = tr.in0.W;
            const std::string pk = phase_weight(e, tr.pfx + "weight");
            const HostTensor& ph = W(e, pk);               // [(phase * Cy + co)][ci][ty * 2 + tx]   (the conv keeps the channel count: Cy == Cx)
            GradEntry& gx = c.G(tr.in0);
            for (int dy = 0; dy < 2; ++dy) {
                ConvParams p = bwd_params(B, Hs, Ws, Hs, Ws, Cx);
                p.src_row_pitch = 2 * Ws;
                for (int dxp = 0; dxp < 2; ++dxp) {
                    const std::string key = pk + "T" + std::to_string(dy) + std::to_string(dxp);
                    if (!e->host.count(key)) {
                        HostTensor tt; tt.shape = {Cx, Cy, 2, 2}; tt.data.resize((size_t)Cx * Cy * 4); tt.loaded = true;
                        for (int ci = 0; ci < Cx; ++ci) for (int co = 0; co < Cy; ++co) for (int u = 0; u < 2; ++u) for (int v = 0; v < 2; ++v)
                            tt.data[((size_t)ci * Cy + co) * 4 + u * 2 + v] = ph.data[(((size_t)(dy * 2 + dxp) * Cy + co) * Cx + ci) * 4 + (1 - u) * 2 + (1 - v)];
                        e->host[key] = std::move(tt);
                    }
                    raw_seg(p, gout.t.p + ((size_t)dy * Wd + dxp) * Cy, Cy, 2 * Cy, 4, nullptr, packed_conv16(e, key, 0, Cy));
                    p.seg[p.nseg - 1].oy = 1 - dy; p.seg[p.nseg - 1].ox = 1 - dxp;
                }
                c.conv_to(p, gx);
            }
        } else if (tr.kind == TP_UP) {
            // out = conv(nearest_up(x)): dU = adjoint conv at the fine resolution, dx = 2x2 sum-pool of dU
            Tensor dU = c.tmp(tr.in0.C, H, Wd);
            ConvParams p = bwd_params(B, H, Wd, H, Wd, tr.in0.C);
            raw_seg(p, gout.t.p, tr.out.C, tr.out.C, 9, packed_conv_T(e, tr.pfx + "weight", 0, tr.in0.C), packed_conv16_T(e, tr.pfx + "weight", 0, tr.in0.C));
            p.out = dU.p; p.out_cstride = dU.C; c.conv_plain(p);
            GradEntry& gx = c.G(tr.in0);
            Op op{}; op.kind = OP_SUMPOOL; op.P[0] = dU.p; op.O = gx.t.p; op.I[0] = tr.in0.H; op.I[1] = tr.in0.W; op.I[2] = tr.in0.C; op.I[3] = gx.has ? 1 : 0;
            gx.has = true; c.push(op);
            bd.recycle(dU.p);
        } else if (tr.kind == TP_DOWN && e->precision != 0 && upadj_env && tr.out.C % 16 == 0 && tr.in0.C % 4 == 0) {
            // out = conv_stride2(x) (Downsample, models.py:50-56): fine pixel (2a + py, 2b + px) is read by tap ky of output row i only when 2i + ky - 1 = 2a + py,
            // i.e. py = 0: (ky, i) = (1, a); py = 1: (0, a + 1) and (2, a) - columns alike.  Per fine phase the adjoint is therefore a conv of the COARSE gradient
            // with 1 / 2 / 2 / 4 taps inside the window rows {a, a + 1} x columns {b, b + 1}, written to every second row and column of dx: 13 multiply-adds per
            // coarse pixel and channel pair (phase (0, 0) as a 1x1 conv, the others as 2 x 2 windows with their unused taps zero) instead of the 36 of a 9-tap
            // conv over the zero-inserted gradient at the fine resolution, three quarters of whose operands are the inserted zeros.
            const int Cy = tr.out.C, Cx = tr.in0.C, Hc = tr.out.H, Wc = tr.out.W;
            const HostTensor& w = W(e, tr.pfx + "weight");       // [Cy][Cx][3][3]
            GradEntry& gx = c.G(tr.in0);
            static const int KOF[2][2] = {{1, -1}, {2, 0}};       // [phase][window index u] -> 3x3 tap index, -1: none
            for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
                const bool one = py == 0 && px == 0;
                const std::string key = tr.pfx + "weight#dT" + std::to_string(py) + std::to_string(px);
                if (!e->host.count(key)) {
                    const int kk = one ? 1 : 4;
                    HostTensor tt; tt.shape = {Cx, Cy, one ? 1 : 2, one ? 1 : 2}; tt.data.assign((size_t)Cx * Cy * kk, 0.f); tt.loaded = true;
                    for (int ci = 0; ci < Cx; ++ci) for (int co = 0; co < Cy; ++co) for (int u = 0; u < (one ? 1 : 2); ++u) for (int v = 0; v < (one ? 1 : 2); ++v) {
                        const int ky = KOF[py][u], kx = KOF[px][v];
                        if (ky >= 0 && kx >= 0) tt.data[((size_t)ci * Cy + co) * kk + u * 2 * (one ? 0 : 1) + v] = w.data[((size_t)co * Cx + ci) * 9 + ky * 3 + kx];
                    }
                    e->host[key] = std::move(tt);
                }
                ConvParams p = bwd_params(B, Hc, Wc, Hc, Wc, Cx);
                raw_seg(p, gout.t.p, Cy, Cy, one ? 1 : 4, nullptr, packed_conv16(e, key, 0, Cy));
                p.seg[0].oy = 1; p.seg[0].ox = 1;
                p.out = gx.t.p + ((size_t)py * tr.in0.W + px) * Cx; p.out_cstride = 2 * Cx; p.dst_row_pitch = 2 * Wc;
                if (gx.has) { p.residual = p.out; p.res_cstride = 2 * Cx; }
                c.conv_plain(p);
            }
            gx.has = true;
        } else if (tr.kind == TP_DOWN) {
            // out = conv_stride2(x): dx = adjoint conv of the zero-inserted gradient
            ConvParams p = bwd_params(B, tr.in0.H, tr.in0.W, H, Wd, tr.in0.C);
            raw_seg(p, gout.t.p, tr.out.C, tr.out.C, 9, packed_conv_T(e, tr.pfx + "weight", 0, tr.in0.C), packed_conv16_T(e, tr.pfx + "weight", 0, tr.in0.C));
            c.conv_to(p, c.G(tr.in0), 1, 2);
        } else if (tr.kind == TP_RES) {
            const ResDesc& r = *tr.r;
            const int cin = tr.in0.C + (tr.has_in1 ? tr.in1.C : 0);
            // conv2 adjoint -> d(act2), GN2+SiLU backward -> d(h1)
            Tensor t1 = c.tmp(r.cout, H, Wd);
            GnBwd gn2 = gn_bwd_begin(c, {tr.h1}, r.prefix + "norm2.", true);
            {
                ConvParams p = bwd_params(B, H, Wd, H, Wd, r.cout);
                raw_seg(p, gout.t.p, r.cout, r.cout, 9, packed_conv_T(e, r.prefix + "conv2.weight", 0, r.cout), packed_conv16_T(e, r.prefix + "conv2.weight", 0, r.cout));
                p.out = t1.p; p.out_cstride = r.cout;
                gn_bwd_fuse(c, gn2, 0, p);
                c.conv_plain(p);
            }
            GradEntry gh1; gh1.t = c.tmp(r.cout, H, Wd);
            gn_bwd_finish(c, gn2, {t1}, {&gh1}, {nullptr});
            bd.recycle(t1.p);
            // conv1 adjoint per source -> d(act1), GN1+SiLU backward over the concatenation
            std::vector<Tensor> srcs{tr.in0}; if (tr.has_in1) srcs.push_back(tr.in1);
            std::vector<Tensor> us; std::vector<GradEntry*> dsts; std::vector<const float*> adds;
            GnBwd gn1 = gn_bwd_begin(c, srcs, r.prefix + "norm1.", true);
            int lo = 0;
            for (size_t sj = 0; sj < srcs.size(); ++sj) {
                const Tensor& sT = srcs[sj];
                Tensor u = c.tmp(sT.C, H, Wd);
                ConvParams p = bwd_params(B, H, Wd, H, Wd, sT.C);
                raw_seg(p, gh1.t.p, r.cout, r.cout, 9, packed_conv_T(e, r.prefix + "conv1.weight", lo, lo + sT.C), packed_conv16_T(e, r.prefix + "conv1.weight", lo, lo + sT.C));
                p.out = u.p; p.out_cstride = sT.C;
                gn_bwd_fuse(c, gn1, sj, p);
                c.conv_plain(p);
                us.push_back(u);
                GradEntry& gd = c.G(sT);
                if (cin != r.cout) {   // 1x1 shortcut adjoint goes straight into the gradient buffer
                    ConvParams q = bwd_params(B, H, Wd, H, Wd, sT.C);
                    raw_seg(q, gout.t.p, r.cout, r.cout, 1, packed_conv_T(e, r.prefix + "shortcut.weight", lo, lo + sT.C), packed_conv16_T(e, r.prefix + "shortcut.weight", lo, lo + sT.C));
                    c.conv_to(q, gd);
                    adds.push_back(nullptr);
                } else {
                    adds.push_back(gout.t.p);   // identity shortcut
                }
                dsts.push_back(&gd);
                lo += sT.C;
            }
            gn_bwd_finish(c, gn1, us, dsts, adds);
            for (auto& u : us) bd.recycle(u.p);
            bd.recycle(gh1.t.p);
        } else {   // TP_ATTN  (models.py:145-162)
            bwd_attn(c, tr, gout);
        }
    }
    // ---- begin: h0 = conv3x3(x) (models.py:451): g = adjoint conv (ch -> image), NCHW out ------------------
    {
        GradEntry& g0 = c.G(plan->t_begin);
        if (!g0.has) { e->err = "internal: no gradient at begin_conv output"; return PF_ERR_INVALID; }
        const HostTensor& w = W(e, "begin_conv.weight");   // [ch][Cimg][3][3]
        const int ci_n = cf.input_channels;
        std::vector<float> wp((size_t)9 * ci_n * ch), zero3(4, 0.f);
        for (int n = 0; n < ch; ++n) for (int ci = 0; ci < ci_n; ++ci) for (int tap = 0; tap < 9; ++tap)
            wp[((size_t)tap * ci_n + ci) * ch + n] = w.data[((size_t)n * ci_n + ci) * 9 + (8 - tap)];
        Op op{}; op.kind = OP_END; op.ep.in = g0.t.p; op.ep.w = upload(e, "begin_conv.adj", wp); op.ep.bias = upload(e, "zero.img", zero3);
        op.ep.B = B; op.ep.H = H0; op.ep.W = H0; op.ep.Cimg = ci_n; op.ep.C = ch; op.ep.stats = nullptr; op.ep.gn_cpg = 1; op.ep.gn_eps = 1e-6f;
        c.push(op);
    }
    plan->bops[0].ptr = (void*)(uintptr_t)(bwd_lo + 1);          // placeholder, resolved against the slab below
    plan->bops[0].bytes = bd.stats_bytes - bwd_lo;
    return PF_OK;
}

// zero fill as a KERNEL: inside the captured per-step graphs a hipMemsetAsync becomes a memset node; the OT-ODE graph (three of them
// per Euler step) produced non-finite statistics in ~1 of 3 runs of 270 steps while the same launches issued eagerly never did
// (profiles/r02_ab_variants_same_box.txt) - with kernels for the fills every node of the graph is a kernel node on one queue
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* p, size_t n16, unsigned char* tail, size_t ntail) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0) for (size_t i = threadIdx.x; i < ntail; i += 256) tail[i] = 0;
}
static hipError_t zero_fill(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (((uintptr_t)ptr & 15) != 0) return hipMemsetAsync(ptr, 0, bytes, s);      // (allocations are 256-B aligned: not taken)
    const size_t n16 = bytes / 16, ntail = bytes % 16;
    const unsigned g = (unsigned)std::min<size_t>((n16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(g), dim3(256), 0, s, (uint4*)ptr, n16, (unsigned char*)ptr + n16 * 16, ntail);
    return hipGetLastError();
}

#include "engine_ncsnpp_bwd.inc"

static hipError_t dispatch_conv(pf_engine* e, const Op& op, hipStream_t s);
static int run_backward(pf_engine* e, Plan* plan, const float* vec, float* g, hipStream_t s) {
    const int B = plan->B;
    // J^T is linear in vec: normalise vec by a power of two (exact) so that the split-fp16 adjoint convs stay in
    // range for any input magnitude, and undo the scale on the result
    const int64_t nimg = (int64_t)B * e->cfg.input_channels * e->cfg.input_height * e->cfg.input_height;
    { hipError_t r = launch_vjp_normalise(vec, plan->vec_scaled, nimg, plan->vjp_amax, plan->vjp_scale, s);
      if (r != hipSuccess) { e->err = std::string("vjp normalise: ") + hipGetErrorString(r); return PF_ERR_HIP; } }
    vec = plan->vec_scaled;
    for (auto& op : plan->bops) {
        hipError_t r = hipSuccess;
        switch (op.kind) {
            case OP_MEMSET: if (op.bytes) r = zero_fill(op.ptr, op.bytes, s); break;
            case OP_BEGIN: { EdgeConvParams ep = op.ep; ep.in = vec; r = launch_begin_conv(ep, s); break; }
            case OP_END: { EdgeConvParams ep = op.ep; ep.out = g; r = launch_end_conv(ep, s); break; }
            case OP_CONV: r = dispatch_conv(e, op, s); break;
            case OP_PREP: r = launch_prep_split(op.pp, s); break;
            case OP_GN_FWD_COEF:
                r = launch_gn_fwd_coeffs((const double*)op.P[0], op.I[0], (const double*)op.P[1], op.I[1], op.I[2], op.I[3], 1e-6f, (float*)op.P[2],
                                         (float*)op.P[3], B, s); break;
            case OP_GN_BWD_PRE:
                r = launch_gn_bwd_pre((float*)op.O, (const float*)op.P[0], (const float*)op.P[1], (const float*)op.P[2], (const float*)op.P[3],
                                      (const float*)op.P[4], (double*)op.P[6], B, op.I[0], op.I[1], op.I[2], op.I[3], op.I[4], s); break;
            case OP_GN_BWD_COEF:
                r = launch_gn_bwd_coeffs((const double*)op.P[6], op.I[0], op.I[1], op.I[2], (float*)op.P[1], (float*)op.P[2], B, s); break;
            case OP_GN_BWD_POST:
                r = launch_gn_bwd_post((const float*)op.P[0], (const float*)op.P[1], (const float*)op.P[2], (const float*)op.P[3], (const float*)op.P[4],
                                       (const float*)op.P[5], (const float*)op.P[7], (float*)op.O, B, op.I[0], op.I[1], op.I[2], op.I[3], op.I[4], s, op.F); break;
            case OP_TRANSPOSE: r = launch_transpose((const float*)op.P[0], (float*)op.O, B, op.I[0], op.I[1], s); break;
            case OP_SOFTMAX_BWD: r = launch_softmax_bwd((const float*)op.P[0], (float*)op.O, op.sm_rows, op.sm_cols, op.F, s); break;
            case OP_SUMPOOL: r = launch_sumpool2((const float*)op.P[0], (float*)op.O, B, op.I[0], op.I[1], op.I[2], op.I[3], s); break;
            case OP_NX_IMG_IN: r = launch_img_to_nhwc32(vec, (float*)op.O, B, op.I[0], op.I[1], op.I[2], s, (const float*)op.P[0]); break;
            case OP_NX_FIR: r = launch_fir_nhwc(op.fp, s); break;
            case OP_NX_IMG_OUT: r = launch_nhwc32_to_img((const float*)op.P[0], g, (const float*)op.P[1], 1.0f, op.P[1] != nullptr ? 1 : 0, B, op.I[0], op.I[1], op.I[2], s); break;
            default: break;
        }
        if (r != hipSuccess) { e->err = std::string("backward launch failed: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    }
    { hipError_t r = launch_scale_inplace(g, nimg, plan->vjp_scale + 1, s);
      if (r != hipSuccess) { e->err = std::string("vjp rescale: ") + hipGetErrorString(r); return PF_ERR_HIP; } }
    return PF_OK;
}

static hipError_t dispatch_conv(pf_engine* e, const Op& op, hipStream_t s) {
    if (op.use_pp == 2 && e->precision != 0) return launch_conv_sp(op.ppp, s, e->precision == 2 ? 1 : 3);
    if (op.use_pp == 1 && e->precision != 0) return launch_conv_pp(op.ppp, s, e->precision == 2 ? 1 : 3);
    if (op.dma) return launch_conv_dma(op.cp, op.up, s, e->precision == 2 ? 1 : 3);
    if (e->precision != 0) {
        bool ok16 = true;
        for (int i = 0; i < op.cp.nseg; ++i) ok16 &= op.cp.seg[i].w_mode == 0 && op.cp.seg[i].w16 != nullptr;
        if (ok16) return launch_conv16(op.cp, op.stride, op.up, s, e->precision == 2 ? 1 : 3);
    }
    return launch_conv(op.cp, op.stride, op.up, s);
}

static int run_plan(pf_engine* e, Plan* plan, const float* x, const float* t, float* v, hipStream_t s, float t_scale = 1.0f) {
    bool prep_open = false;
    for (auto& op : plan->ops) {
        hipError_t r = hipSuccess;
        switch (op.kind) {
            case OP_MEMSET: r = zero_fill(op.ptr, op.bytes, s); break;
            case OP_TEMB: { TembParams tp = op.tp; tp.t = t; r = launch_temb(tp, s); break; }
            case OP_BEGIN: { EdgeConvParams ep = op.ep; ep.in = x; r = launch_begin_conv(ep, s); break; }
            case OP_PREP:
                // profiling: the prep pass is part of its conv's cost - the event pair of the following OP_CONV opens here
                if (e->profile) {
                    if (e->ev_used == e->ev_pool.size()) {
                        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); e->ev_pool.emplace_back(a, b);
                    }
                    hipEventRecord(e->ev_pool[e->ev_used].first, s);
                    prep_open = true;
                }
                r = launch_prep_split(op.pp, s);
                break;
            case OP_CONV:
                if (e->profile) {
                    if (e->ev_used == e->ev_pool.size()) {
                        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); e->ev_pool.emplace_back(a, b);
                    }
                    auto& ev = e->ev_pool[e->ev_used++];
                    if (e->ev_ops.size() < e->ev_used) e->ev_ops.resize(e->ev_used);
                    e->ev_ops[e->ev_used - 1] = &op;
                    if (!prep_open) hipEventRecord(ev.first, s);
                    prep_open = false;
                    r = dispatch_conv(e, op, s);
                    hipEventRecord(ev.second, s);
                    e->prof_flops += (double)op.flops;
                } else {
                    r = dispatch_conv(e, op, s);
                }
                break;
            case OP_SOFTMAX: r = launch_softmax_rows(op.sm, op.sm_rows, op.sm_cols, s); break;
            case OP_ATTN: r = launch_attn_fused(op.ap, s); break;
            case OP_STATS: r = launch_partial_stats((const float*)op.P[0], (double*)op.O, plan->B, op.I[0], op.I[1], s); break;
            case OP_END: { EdgeConvParams ep = op.ep; ep.out = v; r = launch_end_conv(ep, s); break; }
            case OP_GN_COEF: r = launch_gn_coef(op.gp, plan->B, s); break;
            case OP_NX_TEMB: { NxTembParams tp = op.ntp; tp.t = t; tp.t_scale = t_scale; r = launch_nx_temb(tp, s); break; }
            case OP_NX_IMG_IN: r = launch_img_to_nhwc32(x, (float*)op.O, plan->B, op.I[0], op.I[1], op.I[2], s); break;
            case OP_NX_FIR: r = launch_fir_nhwc(op.fp, s); break;
            case OP_NX_IMG_OUT: r = launch_nhwc32_to_img((const float*)op.P[0], v, t, t_scale, op.I[3], plan->B, op.I[0], op.I[1], op.I[2], s, (float*)op.O); break;
            default: break;
        }
        if (r != hipSuccess) { e->err = std::string("kernel launch failed: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    }
    return PF_OK;
}

// --------------------------------------------------------------------------------------
// PnP-Flow outer loop helpers
// --------------------------------------------------------------------------------------
__global__ void prep_iter_kernel(const int* iter, const float* t_all, const float* coef_all, float* t_cur, float* coef_cur, int B) {
    const int it = *iter;
    for (int b = threadIdx.x; b < B; b += blockDim.x) { t_cur[b] = t_all[it]; coef_cur[b] = coef_all[it]; }
}
__global__ void bump_iter_kernel(int* iter) { if (threadIdx.x == 0) *iter += 1; }


static DegView to_view(const pf_degradation* d) {
    DegView v{}; v.kind = d->kind; v.half = d->half_size_mask; v.sf = d->sf; v.ntaps = d->ntaps; v.mask = d->mask; v.taps = d->taps;
    return v;
}

// --------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------
extern "C" {

int pf_abi_version(void) { return PF_ABI_VERSION; }

const char* pf_last_error(const pf_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int pf_engine_create(int device_id, const pf_unet_cfg* cfg, pf_engine** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return PF_ERR_INVALID; }
    if (cfg->num_levels < 1 || cfg->num_levels > 8 || cfg->num_attn_resolutions < 0 || cfg->num_attn_resolutions > 8 ||
        cfg->ch != 32 || cfg->input_channels < 1 || cfg->input_channels > 3 || cfg->output_channels < 1 || cfg->output_channels > 3 ||
        cfg->num_res_blocks < 1 || cfg->input_height % (1 << (cfg->num_levels - 1)) != 0) {
        g_create_err = "unsupported UNet configuration (need ch=32, 1..3 image channels, height divisible by 2^(levels-1))";
        return PF_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0) { g_create_err = std::string("no HIP device: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "bad device id"; return PF_ERR_INVALID; }
    if ((r = hipSetDevice(device_id)) != hipSuccess) { g_create_err = hipGetErrorString(r); return PF_ERR_HIP; }
    auto* e = new pf_engine();
    e->device = device_id; e->cfg = *cfg;
    int rc = build_arch(e);
    if (rc != PF_OK) { g_create_err = e->err; delete e; return rc; }
    *out = e;
    return PF_OK;
}

int pf_ncsnpp_create(int device_id, const pf_ncsnpp_cfg* cfg, pf_engine** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return PF_ERR_INVALID; }
    bool ok = cfg->num_levels >= 1 && cfg->num_levels <= 8 && cfg->num_attn_resolutions >= 0 && cfg->num_attn_resolutions <= 8 &&
              cfg->nf >= 32 && cfg->nf <= 128 && cfg->nf % 32 == 0 && cfg->num_channels >= 1 && cfg->num_channels <= 3 && cfg->num_res_blocks >= 1 &&
              cfg->fir_taps >= 2 && cfg->fir_taps <= 8 && cfg->fir_taps % 2 == 0 && cfg->centered == 1 && cfg->image_size > 0 &&
              cfg->image_size % (1 << (cfg->num_levels - 1)) == 0;
    for (int i = 0; ok && i < cfg->num_levels; ++i) ok = cfg->ch_mult[i] >= 1 && cfg->nf * cfg->ch_mult[i] <= 256;   // GroupNorm of a cat input parks <= 1024 channels; FIR lanes <= 512
    if (!ok) {
        g_create_err = "unsupported NCSN++ configuration (need nf in {32,64,96,128}, nf*ch_mult <= 256, 1..3 image channels, centered data, "
                       "an even-length FIR kernel, image_size divisible by 2^(levels-1))";
        return PF_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0) { g_create_err = std::string("no HIP device: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "bad device id"; return PF_ERR_INVALID; }
    if ((r = hipSetDevice(device_id)) != hipSuccess) { g_create_err = hipGetErrorString(r); return PF_ERR_HIP; }
    auto* e = new pf_engine();
    e->device = device_id; e->arch = 1; e->ncfg = *cfg;
    e->cfg.input_channels = e->cfg.output_channels = cfg->num_channels; e->cfg.input_height = cfg->image_size;      // what the solver loops read
    int rc = nx_build_arch(e);
    if (rc != PF_OK) { g_create_err = e->err; delete e; return rc; }
    *out = e;
    return PF_OK;
}

static void drop_graph(pf_engine* e);
static void drop_ode_graph(pf_engine* e);
int pf_engine_set_solver_time_scale(pf_engine* e, float scale) {
    if (!e || !(scale > 0.f)) return PF_ERR_INVALID;
    if (scale != e->solver_time_scale) { drop_graph(e); drop_ode_graph(e); e->solver_time_scale = scale; }   // the captured graphs bake the scale in
    return PF_OK;
}

// reads (and clears) the numeric-health flags of every plan; the caller has synchronised the stream(s) that ran them
static int check_flags(pf_engine* e) {
    bool bad = false; std::string where;
    for (auto& kv : e->plans) {
        Plan* pl = kv.second.get();
        if (!pl->flags) continue;
        unsigned int f[2] = {0, 0};
        HIPCHK(e, hipMemcpy(f, pl->flags, sizeof f, hipMemcpyDeviceToHost));
        if (f[0]) {
            bad = true; HIPCHK(e, hipMemset(pl->flags, 0, sizeof f)); HIPCHK(e, hipDeviceSynchronize());
            char buf[160] = "";
            const int id = (int)f[1];                 // the op that follows the flagging finalisation is the consuming conv
            if (id >= 1 && id < (int)pl->ops.size() && pl->ops[id].kind == OP_CONV)
                snprintf(buf, sizeof buf, " (first seen by launch #%d of the B=%d plan: %dx%d, K segments %d, Cout %d)", id, pl->B, pl->ops[id].cp.H,
                         pl->ops[id].cp.W, pl->ops[id].cp.nseg, pl->ops[id].cp.Cout);
            where = buf;
        }
    }
    if (bad) {
        e->err = "non-finite activation statistics: an activation overflowed or was NaN inside the U-Net (inputs / weights out of range)" + where;
        return PF_ERR_NUMERIC;
    }
    return PF_OK;
}

static void free_ode(pf_engine* e);

static void drop_graph(pf_engine* e) {
    if (e->gexec) hipGraphExecDestroy(e->gexec);
    if (e->graph) hipGraphDestroy(e->graph);
    e->gexec = nullptr; e->graph = nullptr; e->gkey = pf_engine::GraphKey{};
}

static void free_solver(pf_engine* e) {
    drop_graph(e);          // its nodes point into the solver buffers
    SolverBufs& b = e->sb;
    for (void* p : {(void*)b.x, (void*)b.z, (void*)b.zt, (void*)b.v, (void*)b.scratch, (void*)b.y, (void*)b.rng, (void*)b.t_all,
                    (void*)b.coef_all, (void*)b.t_cur, (void*)b.coef_cur, (void*)b.iter})
        if (p) hipFree(p);
    e->bytes -= b.bytes;
    b = SolverBufs{};
}

void pf_engine_destroy(pf_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    drop_graph(e);
    free_ode(e);
    for (auto& kv : e->plans) for (void* p : kv.second->allocs) hipFree(p);
    for (void* p : e->weight_allocs) hipFree(p);
    for (auto& ev : e->ev_pool) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    free_solver(e);
    if (e->work_stream) hipStreamDestroy(e->work_stream);
    delete e;
}

int pf_engine_num_weights(const pf_engine* e) { return e ? (int)e->expected.size() : 0; }
const char* pf_engine_weight_name(const pf_engine* e, int i) {
    if (!e || i < 0 || i >= (int)e->expected.size()) return nullptr;
    return e->expected[i].first.c_str();
}

int pf_engine_weight_shape(const pf_engine* e, int i, int64_t shape[4]) {
    if (!e || !shape || i < 0 || i >= (int)e->expected.size()) return PF_ERR_INVALID;
    const auto& s = e->expected[i].second;
    for (size_t k = 0; k < s.size() && k < 4; ++k) shape[k] = s[k];
    return (int)s.size();
}

int pf_engine_load_weight(pf_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    if (!e || !name || !host_data || !shape) return PF_ERR_INVALID;
    if (e->finalized) { e->err = "weights already finalized"; return PF_ERR_STATE; }
    for (auto& ex : e->expected) {
        if (ex.first != name) continue;
        if ((int)ex.second.size() != ndim) { e->err = std::string("rank mismatch for ") + name; return PF_ERR_WEIGHTS; }
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) {
            if (ex.second[i] != shape[i]) { e->err = std::string("shape mismatch for ") + name; return PF_ERR_WEIGHTS; }
            n *= (size_t)shape[i];
        }
        HostTensor& t = e->host[name];
        t.shape.assign(shape, shape + ndim); t.data.assign(host_data, host_data + n); t.loaded = true;
        return PF_OK;
    }
    e->err = std::string("unexpected weight: ") + name;
    return PF_ERR_WEIGHTS;
}

int pf_engine_finalize_weights(pf_engine* e) {
    if (!e) return PF_ERR_INVALID;
    for (auto& ex : e->expected) {
        auto it = e->host.find(ex.first);
        if (it == e->host.end() || !it->second.loaded) { e->err = "missing weight: " + ex.first; return PF_ERR_WEIGHTS; }
    }
    USE_DEVICE(e);
    e->finalized = true;
    return PF_OK;
}

int pf_engine_set_precision(pf_engine* e, int mode) {
    if (!e) return PF_ERR_INVALID;
    if (mode < 0 || mode > 2) { e->err = "precision mode must be 0 (fp32 MFMA), 1 (split-fp16 MFMA, fp32-equivalent) or 2 (single fp16 MFMA)"; return PF_ERR_INVALID; }
    if (mode != e->precision) { drop_graph(e); drop_ode_graph(e); }       // captured graphs bake the kernel choice in
    e->precision = mode;
    return PF_OK;
}

int pf_unet_forward(pf_engine* e, const float* x, const float* t, float* v, int B, void* stream) {
    if (!e || !x || !t || !v || B <= 0) return PF_ERR_INVALID;
    if (!e->finalized) { e->err = "weights not finalized"; return PF_ERR_STATE; }
    USE_DEVICE(e);
    Plan* plan = nullptr;
    int rc = build_plan(e, B, false, &plan);
    if (rc != PF_OK) return rc;
    return run_plan(e, plan, x, t, v, (hipStream_t)stream);
}

int pf_unet_forward_retain(pf_engine* e, const float* x, const float* t, float* v, int B, void* stream) {
    if (!e || !x || !t || !v || B <= 0) return PF_ERR_INVALID;
    if (!e->finalized) { e->err = "weights not finalized"; return PF_ERR_STATE; }
    USE_DEVICE(e);
    Plan* plan = nullptr;
    int rc = build_plan(e, B, true, &plan);
    if (rc != PF_OK) return rc;
    e->retained_B = B; e->retained_plan = plan;
    return run_plan(e, plan, x, t, v, (hipStream_t)stream);
}

int pf_unet_backward(pf_engine* e, const float* vec, float* g, int B, void* stream) {
    if (!e || !vec || !g || B <= 0) return PF_ERR_INVALID;
    if (e->retained_B != B || !e->retained_plan) { e->err = "pf_unet_backward: no retained forward with this batch size"; return PF_ERR_STATE; }
    USE_DEVICE(e);
    Plan* plan = nullptr;
    int rc = build_plan(e, B, true, &plan);
    if (rc != PF_OK) return rc;
    if (plan != e->retained_plan) {      // e.g. pf_engine_set_precision between the retained forward and the backward
        e->err = "pf_unet_backward: the retained forward ran under a different precision mode / plan";
        return PF_ERR_STATE;
    }
    return run_backward(e, plan, vec, g, (hipStream_t)stream);
}

int pf_unet_vjp(pf_engine* e, const float* x, const float* t, const float* vec, float* v, float* g, int B, void* stream) {
    int rc = pf_unet_forward_retain(e, x, t, v, B, stream);
    if (rc != PF_OK) return rc;
    return pf_unet_backward(e, vec, g, B, stream);
}

int pf_ot_ode_vec(const pf_degradation* d, const float* x, const float* vt, const float* y, const float* one_minus_t, const float* rt2,
                  float sigma2, float* vec, int B, int C, int H, int W, float* scratch, void* stream) {
    if (!d || !x || !vt || !y || !one_minus_t || !rt2 || !vec) return PF_ERR_INVALID;
    if (d->kind == PF_DEG_GAUSSIAN_BLUR) {
        if (!scratch) return PF_ERR_INVALID;
        LAUNCHCHK(launch_ot_ode_vec_blur(to_view(d), x, vt, y, one_minus_t, rt2, sigma2, vec, B, C, H, W, scratch, (hipStream_t)stream));
        return PF_OK;
    }
    LAUNCHCHK(launch_ot_ode_vec(to_view(d), x, vt, y, one_minus_t, rt2, sigma2, vec, B, C, H, W, (hipStream_t)stream));
    return PF_OK;
}

int pf_ot_ode_update(float* x, const float* vt, const float* vec, const float* g, const float* one_minus_t, const float* coef, float delta,
                     int B, int n_per_image, void* stream) {
    if (!x || !vt || !vec || !g || !one_minus_t || !coef) return PF_ERR_INVALID;
    LAUNCHCHK(launch_ot_ode_update(x, vt, vec, g, one_minus_t, coef, delta, B, n_per_image, (hipStream_t)stream));
    return PF_OK;
}

int pf_engine_num_taps(const pf_engine* e) {
    if (!e || !e->last_plan) return 0;
    return (int)e->last_plan->taps.size();
}
const char* pf_engine_tap_name(const pf_engine* e, int i) {
    if (!e || !e->last_plan) return nullptr;
    auto& taps = e->last_plan->taps;
    return (i >= 0 && i < (int)taps.size()) ? taps[i].name.c_str() : nullptr;
}
int pf_engine_read_tap(pf_engine* e, int i, float* host_out, int64_t capacity, int32_t dims[3], void* stream) {
    if (!e || !e->last_plan || !host_out || !dims) return PF_ERR_INVALID;
    Plan* plan = e->last_plan;
    if (i < 0 || i >= (int)plan->taps.size()) return PF_ERR_INVALID;
    const Tensor& t = plan->taps[i].t;
    const size_t n = (size_t)plan->B * t.C * t.H * t.W;
    if ((int64_t)n > capacity) { e->err = "tap buffer too small"; return PF_ERR_INVALID; }
    std::vector<float> nhwc(n);
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(e, hipMemcpy(nhwc.data(), t.p, n * sizeof(float), hipMemcpyDeviceToHost));
    const size_t HW = (size_t)t.H * t.W;
    for (int b = 0; b < plan->B; ++b)
        for (size_t p = 0; p < HW; ++p)
            for (int c = 0; c < t.C; ++c) host_out[((size_t)b * t.C + c) * HW + p] = nhwc[((size_t)b * HW + p) * t.C + c];
    dims[0] = t.C; dims[1] = t.H; dims[2] = t.W;
    return PF_OK;
}


int pf_degradation_H(const pf_degradation* d, const float* x, float* y, int B, int C, int H, int W, float* scratch, void* stream) {
    if (!d || !x || !y) return PF_ERR_INVALID;
    LAUNCHCHK(launch_deg_H(to_view(d), x, y, B, C, H, W, scratch, (hipStream_t)stream));
    return PF_OK;
}
int pf_degradation_H_adj(const pf_degradation* d, const float* y, float* x, int B, int C, int H, int W, float* scratch, void* stream) {
    if (!d || !x || !y) return PF_ERR_INVALID;
    LAUNCHCHK(launch_deg_Hadj(to_view(d), y, x, B, C, H, W, scratch, (hipStream_t)stream));
    return PF_OK;
}
int pf_grad_step(const pf_degradation* d, const float* x, const float* y, const float* coef, float* z, int B, int C, int H, int W,
                 float* scratch, void* stream) {
    if (!d || !x || !y || !coef || !z) return PF_ERR_INVALID;
    LAUNCHCHK(launch_grad_step(to_view(d), x, y, coef, z, B, C, H, W, scratch, 0, (hipStream_t)stream));
    return PF_OK;
}
int pf_grad_step_laplace(const pf_degradation* d, const float* x, const float* y, const float* coef, float* z, int B, int C, int H, int W,
                         float* scratch, void* stream) {
    if (!d || !x || !y || !coef || !z) return PF_ERR_INVALID;
    LAUNCHCHK(launch_grad_step(to_view(d), x, y, coef, z, B, C, H, W, scratch, 1, (hipStream_t)stream));
    return PF_OK;
}
int pf_interpolate(const float* z, const float* t, const float* noise, uint64_t seed, uint64_t stream_id, float* z_tilde, int B,
                   int n_per_image, void* stream) {
    if (!z || !t || !z_tilde) return PF_ERR_INVALID;
    LAUNCHCHK(launch_interpolate(z, t, noise, seed, stream_id, z_tilde, B, n_per_image, (hipStream_t)stream));
    return PF_OK;
}
int pf_denoise_accumulate(float* acc, const float* z_tilde, const float* v, const float* t, int mode, float num_samples, int B,
                          int n_per_image, void* stream) {
    if (!acc || !z_tilde || !v || !t) return PF_ERR_INVALID;
    LAUNCHCHK(launch_denoise_accum(acc, z_tilde, v, t, mode, num_samples, B, n_per_image, (hipStream_t)stream));
    return PF_OK;
}
int pf_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
    if (!out || n < 0) return PF_ERR_INVALID;
    LAUNCHCHK(launch_fill_normal(out, n, seed, stream_id, 0, (hipStream_t)stream));
    return PF_OK;
}
int pf_fill_normal_at(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, void* stream) {
    if (!out || n < 0) return PF_ERR_INVALID;
    LAUNCHCHK(launch_fill_normal(out, n, seed, stream_id, elem_offset, (hipStream_t)stream));
    return PF_OK;
}
int pf_attention_core(const float* qkv, float* out, int B, int T, int C, void* stream) {
    if (!qkv || !out || B <= 0 || !attn_fused_supported(T, C)) return PF_ERR_INVALID;
    AttnParams ap{qkv, out, B, T, C, 1.0f / sqrtf((float)C)};
    LAUNCHCHK(launch_attn_fused(ap, (hipStream_t)stream));
    return PF_OK;
}

int pf_psnr(const float* rec, const float* clean, float* out, int B, int n_per_image, void* stream) {
    if (!rec || !clean || !out) return PF_ERR_INVALID;
    LAUNCHCHK(launch_psnr(rec, clean, out, B, n_per_image, (hipStream_t)stream));
    return PF_OK;
}

int pf_upfirdn2d(const float* in, const float* kernel, float* out, int planes, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                 int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
    if (!in || !kernel || !out) return PF_ERR_INVALID;
    LAUNCHCHK(launch_upfirdn2d(in, kernel, out, planes, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, (hipStream_t)stream));
    return PF_OK;
}
int pf_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t n, int step_b, int size_b, int act, int grad,
                      float alpha, float scale, void* stream) {
    if (!x || !out) return PF_ERR_INVALID;
    LAUNCHCHK(launch_fused_bias_act(x, bias, ref, out, n, step_b, size_b, act, grad, alpha, scale, (hipStream_t)stream));
    return PF_OK;
}

int pf_ssim(const float* rec, const float* clean, double* out, int B, int C, int H, int W, void* stream) {
    if (!rec || !clean || !out) return PF_ERR_INVALID;
    LAUNCHCHK(launch_ssim(rec, clean, out, B, C, H, W, (hipStream_t)stream));
    return PF_OK;
}

static int ensure_solver(pf_engine* e, int B, size_t n, size_t ny, int steps, int ns) {
    SolverBufs& b = e->sb;
    if (b.B == B && b.n == n && b.ny == ny && b.steps >= steps && b.ns >= ns) return PF_OK;
    free_solver(e);
    const size_t tot = (size_t)B * n;
    HIPCHK(e, hipMalloc(&b.x, tot * 4)); HIPCHK(e, hipMalloc(&b.z, tot * 4)); HIPCHK(e, hipMalloc(&b.zt, ns * tot * 4));
    HIPCHK(e, hipMalloc(&b.v, ns * tot * 4)); HIPCHK(e, hipMalloc(&b.scratch, 2 * tot * 4));
    HIPCHK(e, hipMalloc(&b.t_all, (size_t)steps * 4)); HIPCHK(e, hipMalloc(&b.coef_all, (size_t)steps * 4));
    HIPCHK(e, hipMalloc(&b.t_cur, (size_t)ns * B * 4)); HIPCHK(e, hipMalloc(&b.coef_cur, (size_t)ns * B * 4));
    HIPCHK(e, hipMalloc(&b.iter, 64));
    HIPCHK(e, hipMalloc(&b.y, (size_t)B * ny * 4)); HIPCHK(e, hipMalloc(&b.rng, 64));
    if (poison_enabled()) { poison(b.x, tot * 4, 2); poison(b.z, tot * 4, 2); poison(b.zt, ns * tot * 4, 2); poison(b.v, ns * tot * 4, 2); poison(b.scratch, 2 * tot * 4, 2); poison(b.y, (size_t)B * ny * 4, 2);
                            poison(b.t_cur, (size_t)ns * B * 4, 2); poison(b.coef_cur, (size_t)ns * B * 4, 2); }
    b.B = B; b.n = n; b.ny = ny; b.steps = steps; b.ns = ns;
    b.bytes = (int64_t)((1 + 1 + 2 * (size_t)ns + 2) * tot * 4 + (size_t)B * ny * 4 + 2 * (size_t)steps * 4 + 2 * (size_t)ns * B * 4 + 128);
    e->bytes += b.bytes;
    return PF_OK;
}

static int enqueue_iteration(pf_engine* e, Plan* plan, const DegView& dv, const pf_pnp_params* prm, const float* y, int B, int C,
                             int H, hipStream_t s) {
    SolverBufs& b = e->sb;
    const int n = C * H * H;
    const int nsb = prm->batch_samples ? prm->num_samples : 1;
    hipLaunchKernelGGL(prep_iter_kernel, dim3(1), dim3(64), 0, s, (const int*)b.iter, (const float*)b.t_all, (const float*)b.coef_all,
                       b.t_cur, b.coef_cur, nsb * B);
    hipError_t r = launch_grad_step(dv, b.x, y, b.coef_cur, b.z, B, C, H, H, b.scratch, prm->noise_model == 1 ? 1 : 0, s);
    if (r != hipSuccess) { e->err = std::string("grad_step: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    if (prm->batch_samples) {
        // the num_samples velocity evaluations of one outer iteration are independent given z: run them as ONE
        // U-Net pass over num_samples*B images (`plan` was built for that batch); the average keeps the
        // reference's summation order (pnp_flow.py:114-121)
        const size_t tot = (size_t)B * n;
        for (int smp = 0; smp < prm->num_samples; ++smp) {
            r = launch_interp_iter(b.z, b.t_cur, prm->noise, b.rng, b.iter, prm->num_samples, smp, b.zt + smp * tot, B, n, s);
            if (r != hipSuccess) { e->err = std::string("interpolate: ") + hipGetErrorString(r); return PF_ERR_HIP; }
        }
        int rc = run_plan(e, plan, b.zt, b.t_cur, b.v, s, e->solver_time_scale);
        if (rc != PF_OK) return rc;
        for (int smp = 0; smp < prm->num_samples; ++smp) {
            const int mode = (smp == 0 ? 1 : 0) | (smp == prm->num_samples - 1 ? 2 : 0);
            r = launch_denoise_accum(b.x, b.zt + smp * tot, b.v + smp * tot, b.t_cur, mode, (float)prm->num_samples, B, n, s);
            if (r != hipSuccess) { e->err = std::string("denoise_accum: ") + hipGetErrorString(r); return PF_ERR_HIP; }
        }
    } else
    for (int smp = 0; smp < prm->num_samples; ++smp) {
        r = launch_interp_iter(b.z, b.t_cur, prm->noise, b.rng, b.iter, prm->num_samples, smp, b.zt, B, n, s);
        if (r != hipSuccess) { e->err = std::string("interpolate: ") + hipGetErrorString(r); return PF_ERR_HIP; }
        int rc = run_plan(e, plan, b.zt, b.t_cur, b.v, s, e->solver_time_scale);
        if (rc != PF_OK) return rc;
        const int mode = (smp == 0 ? 1 : 0) | (smp == prm->num_samples - 1 ? 2 : 0);
        r = launch_denoise_accum(b.x, b.zt, b.v, b.t_cur, mode, (float)prm->num_samples, B, n, s);
        if (r != hipSuccess) { e->err = std::string("denoise_accum: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    }
    hipLaunchKernelGGL(bump_iter_kernel, dim3(1), dim3(64), 0, s, b.iter);
    r = hipGetLastError();
    if (r != hipSuccess) { e->err = std::string("iteration: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    return PF_OK;
}

int pf_pnp_flow_restore(pf_engine* e, const pf_degradation* d, const pf_pnp_params* prm, const float* y, float* x_out, int B,
                        void* stream, pf_iter_callback iter_cb, void* user) {
    if (!e || !d || !prm || !y || !x_out || B <= 0 || prm->steps <= 0 || prm->num_samples <= 0 || !prm->host_t || !prm->host_coef)
        return PF_ERR_INVALID;
    if (!e->finalized) { e->err = "weights not finalized"; return PF_ERR_STATE; }
    USE_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (prm->use_graph && s == nullptr) {
        // the legacy NULL stream cannot be captured: run on an engine-owned stream, ordered after everything already
        // enqueued on the NULL stream (the function synchronises before returning).  The stream is a BLOCKING one
        // (hipStreamDefault): work a callback puts on the NULL stream and the engine's next launches stay ordered by the
        // legacy-stream rule (round 2: on a non-blocking stream, metric kernels launched from the callbacks on the NULL
        // stream and cached-graph replays produced NaNs on the second batch; the Python solvers also hand over a real stream)
        if (!e->work_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->work_stream, hipStreamDefault));
        HIPCHK(e, hipStreamSynchronize(nullptr));
        s = e->work_stream;
    }
    const int C = e->cfg.input_channels, H = e->cfg.input_height;
    if (e->cfg.output_channels != C) { e->err = "restoration needs output_channels == input_channels"; return PF_ERR_INVALID; }
    const size_t n = (size_t)C * H * H;
    const int Hy = (d->kind == PF_DEG_SUPERRESOLUTION || d->kind == PF_DEG_SR_FILTERED) ? H / std::max(1, d->sf) : H;
    const size_t ny = (size_t)C * Hy * Hy;
    int rc = ensure_solver(e, B, n, ny, prm->steps, prm->batch_samples ? prm->num_samples : 1);
    if (rc != PF_OK) return rc;
    SolverBufs& b = e->sb;
    Plan* plan = nullptr;
    if ((rc = build_plan(e, prm->batch_samples ? B * prm->num_samples : B, false, &plan)) != PF_OK) return rc;
    const DegView dv = to_view(d);
    const unsigned long long rng_host[3] = {prm->seed, prm->stream_base, prm->elem_offset};
    HIPCHK(e, hipMemcpyAsync(b.t_all, prm->host_t, (size_t)prm->steps * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.coef_all, prm->host_coef, (size_t)prm->steps * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.rng, rng_host, sizeof rng_host, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.y, y, (size_t)B * ny * 4, hipMemcpyDeviceToDevice, s));     // the graph's nodes read the engine's own copy
    HIPCHK(e, hipMemsetAsync(b.iter, 0, 64, s));
    // x0 = H_adj(ones_like(y))   (pnp_flow.py:93)
    HIPCHK(e, launch_fill(b.zt, (int64_t)B * ny, 1.0f, s));
    HIPCHK(e, launch_deg_Hadj(dv, b.zt, b.x, B, C, H, H, b.scratch, s));
    HIPCHK(e, hipStreamSynchronize(s));   // host_t/host_coef/rng_host may go away after return; also orders the memcpys

    // One outer iteration = one hipGraph (gradient step, interpolation, U-Net pass, average): iteration 0 runs eagerly (all
    // lazy initialisation done), the graph is captured once and kept while the captured arguments stay the same (every
    // per-batch / per-shard quantity - t, lr_t/sigma^2, noise streams, y - is read from engine-owned device buffers).
    const pf_engine::GraphKey key{plan, dv.kind, dv.half, dv.sf, dv.ntaps, dv.mask, dv.taps, prm->noise, prm->num_samples,
                                  prm->batch_samples, prm->noise_model, B};
    if (e->gexec && memcmp(&key, &e->gkey, sizeof key) != 0) drop_graph(e);
    const bool can_graph = prm->use_graph && !e->profile;
    for (int it = 0; it < prm->steps; ++it) {
        if (can_graph && (it >= 1 || e->gexec)) {
            if (!e->gexec) {
                HIPCHK(e, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                rc = enqueue_iteration(e, plan, dv, prm, b.y, B, C, H, s);
                hipGraph_t g = nullptr;
                hipError_t ce = hipStreamEndCapture(s, &g);
                if (rc != PF_OK) { if (g) hipGraphDestroy(g); return rc; }
                if (ce != hipSuccess) { e->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(ce); return PF_ERR_HIP; }
                e->graph = g;
                hipError_t ie = hipGraphInstantiate(&e->gexec, e->graph, nullptr, nullptr, 0);
                if (ie != hipSuccess) { drop_graph(e); e->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(ie); return PF_ERR_HIP; }
                memset(&e->gkey, 0, sizeof e->gkey); e->gkey = key;
            }
            HIPCHK(e, hipGraphLaunch(e->gexec, s));
        } else {
            if ((rc = enqueue_iteration(e, plan, dv, prm, b.y, B, C, H, s)) != PF_OK) return rc;
        }
        // the host is involved only on the iterations the caller asked for (the reference touches it on its logging
        // iterations only, pnp_flow.py:128-139)
        if (iter_cb && (!prm->host_cb_mask || prm->host_cb_mask[it])) {
            HIPCHK(e, hipMemcpyAsync(x_out, b.x, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
            HIPCHK(e, hipStreamSynchronize(s));
            iter_cb(it, user);
        }
    }
    HIPCHK(e, hipMemcpyAsync(x_out, b.x, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));
    return check_flags(e);
}

int64_t pf_engine_memory_bytes(const pf_engine* e) { return e ? e->bytes : 0; }

int pf_engine_check_numerics(pf_engine* e, void* stream) {
    if (!e) return PF_ERR_INVALID;
    USE_DEVICE(e);
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    return check_flags(e);
}


// ---- OT-ODE restoration loop on the device (pnpflow/methods/ot_ode.py:49-52, 63-147) ---------------------------------------
__global__ void ode_prep_kernel(const int* iter, const float* tab, int steps, float* cur, int B) {
    const int it = *iter;
    for (int i = threadIdx.x; i < 4 * B; i += blockDim.x) cur[i] = tab[(i / B) * steps + it];
}

static void drop_ode_graph(pf_engine* e) {
    if (e->ogexec) hipGraphExecDestroy(e->ogexec);
    if (e->ograph) hipGraphDestroy(e->ograph);
    e->ogexec = nullptr; e->ograph = nullptr; e->okey = pf_engine::OdeKey{};
}

static void free_ode(pf_engine* e) {
    drop_ode_graph(e);
    auto& b = e->ob;
    for (void* p : {(void*)b.x, (void*)b.vt, (void*)b.vec, (void*)b.g, (void*)b.y, (void*)b.scratch, (void*)b.tab, (void*)b.cur, (void*)b.iter})
        if (p) hipFree(p);
    e->bytes -= b.bytes;
    b = pf_engine::OdeBufs{};
}

static int ensure_ode(pf_engine* e, int B, size_t n, size_t ny, int steps, bool blur, int H) {
    auto& b = e->ob;
    if (b.B == B && b.n == n && b.ny == ny && b.steps >= steps && b.blur == blur) return PF_OK;
    free_ode(e);
    const size_t tot = (size_t)B * n;
    HIPCHK(e, hipMalloc(&b.x, tot * 4)); HIPCHK(e, hipMalloc(&b.vt, tot * 4)); HIPCHK(e, hipMalloc(&b.vec, tot * 4)); HIPCHK(e, hipMalloc(&b.g, tot * 4));
    HIPCHK(e, hipMalloc(&b.y, (size_t)B * ny * 4));
    const size_t scr = blur ? 4 * tot + 2 * (size_t)H : 0;
    if (scr) HIPCHK(e, hipMalloc(&b.scratch, scr * 4));
    HIPCHK(e, hipMalloc(&b.tab, (size_t)4 * steps * 4)); HIPCHK(e, hipMalloc(&b.cur, (size_t)4 * B * 4)); HIPCHK(e, hipMalloc(&b.iter, 64));
    if (poison_enabled()) { for (float* q : {b.x, b.vt, b.vec, b.g}) poison(q, tot * 4, 4); poison(b.y, (size_t)B * ny * 4, 4); poison(b.scratch, scr * 4, 4); poison(b.tab, (size_t)4 * steps * 4, 4); poison(b.cur, (size_t)4 * B * 4, 4); }
    b.B = B; b.n = n; b.ny = ny; b.steps = steps; b.blur = blur;
    b.bytes = (int64_t)(4 * tot * 4 + (size_t)B * ny * 4 + scr * 4 + (size_t)4 * steps * 4 + (size_t)4 * B * 4 + 64);
    e->bytes += b.bytes;
    return PF_OK;
}

// one Euler step (ot_ode.py:67-147): every per-iteration scalar is read on the device from the schedule tables through the
// iteration counter, so one captured graph serves every iteration
static int enqueue_ode_step(pf_engine* e, Plan* plan, const DegView& dv, const pf_ot_ode_params* prm, int B, int C, int H, hipStream_t s) {
    auto& b = e->ob;
    const int n = C * H * H;
    hipLaunchKernelGGL(ode_prep_kernel, dim3(1), dim3(256), 0, s, (const int*)b.iter, (const float*)b.tab, b.steps, b.cur, B);
    const float* t_cur = b.cur; const float* omt = b.cur + B; const float* rt2 = b.cur + 2 * B; const float* coef = b.cur + 3 * B;
    int rc = run_plan(e, plan, b.x, t_cur, b.vt, s, e->solver_time_scale);                                        // v_t = v_theta(x, t), activations retained
    if (rc != PF_OK) return rc;
    hipError_t r = dv.kind == DEG_BLUR
        ? launch_ot_ode_vec_blur(dv, b.x, b.vt, b.y, omt, rt2, prm->sigma2, b.vec, B, C, H, H, b.scratch, s)
        : launch_ot_ode_vec(dv, b.x, b.vt, b.y, omt, rt2, prm->sigma2, b.vec, B, C, H, H, s);
    if (r != hipSuccess) { e->err = std::string("ot_ode vec: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    if ((rc = run_backward(e, plan, b.vec, b.g, s)) != PF_OK) return rc;                     // g = J^T vec
    r = launch_ot_ode_update(b.x, b.vt, b.vec, b.g, omt, coef, prm->delta, B, n, s);
    if (r != hipSuccess) { e->err = std::string("ot_ode update: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    hipLaunchKernelGGL(bump_iter_kernel, dim3(1), dim3(64), 0, s, b.iter);
    r = hipGetLastError();
    if (r != hipSuccess) { e->err = std::string("ot_ode step: ") + hipGetErrorString(r); return PF_ERR_HIP; }
    return PF_OK;
}

int pf_ot_ode_restore(pf_engine* e, const pf_degradation* d, const pf_ot_ode_params* prm, const float* y, float* x_inout, int B,
                      void* stream, pf_iter_callback iter_cb, void* user) {
    if (!e || !d || !prm || !y || !x_inout || B <= 0 || prm->steps <= 0 || prm->first < 0 || prm->first > prm->steps || !prm->host_t ||
        !prm->host_one_minus_t || !prm->host_rt2 || !prm->host_coef)
        return PF_ERR_INVALID;
    if (!e->finalized) { e->err = "weights not finalized"; return PF_ERR_STATE; }
    USE_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (prm->use_graph && s == nullptr) {
        if (!e->work_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->work_stream, hipStreamDefault));
        HIPCHK(e, hipStreamSynchronize(nullptr));
        s = e->work_stream;
    }
    const int C = e->cfg.input_channels, H = e->cfg.input_height;
    if (e->cfg.output_channels != C) { e->err = "restoration needs output_channels == input_channels"; return PF_ERR_INVALID; }
    if (d->kind == PF_DEG_SR_FILTERED) { e->err = "ot_ode: no closed-form solve for the filtered superresolution operator"; return PF_ERR_INVALID; }
    const size_t n = (size_t)C * H * H;
    const int Hy = d->kind == PF_DEG_SUPERRESOLUTION ? H / std::max(1, d->sf) : H;
    const size_t ny = (size_t)C * Hy * Hy;
    int rc = ensure_ode(e, B, n, ny, prm->steps, d->kind == PF_DEG_GAUSSIAN_BLUR, H);
    if (rc != PF_OK) return rc;
    auto& b = e->ob;
    Plan* plan = nullptr;
    if ((rc = build_plan(e, B, true, &plan)) != PF_OK) return rc;
    e->retained_B = B; e->retained_plan = plan;
    const DegView dv = to_view(d);
    std::vector<float> tab((size_t)4 * b.steps, 0.f);
    for (int i = 0; i < prm->steps; ++i) {
        tab[i] = prm->host_t[i]; tab[(size_t)b.steps + i] = prm->host_one_minus_t[i];
        tab[(size_t)2 * b.steps + i] = prm->host_rt2[i]; tab[(size_t)3 * b.steps + i] = prm->host_coef[i];
    }
    const int first = prm->first;
    HIPCHK(e, hipMemcpyAsync(b.tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.iter, &first, sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.y, y, (size_t)B * ny * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipMemcpyAsync(b.x, x_inout, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));      // the host tables may go away after return

    static_assert(sizeof(pf_engine::OdeKey) == 3 * sizeof(void*) + 8 * sizeof(int), "OdeKey is compared with memcmp: it must have no padding bytes");
    const pf_engine::OdeKey key{plan, dv.kind, dv.half, dv.sf, dv.ntaps, dv.mask, dv.taps, B, prm->sigma2, prm->delta, 0};
    if (e->ogexec && memcmp(&key, &e->okey, sizeof key) != 0) drop_ode_graph(e);
    const bool can_graph = prm->use_graph && !e->profile;
    for (int it = first; it < prm->steps; ++it) {
        if (can_graph && (it > first || e->ogexec)) {
            if (!e->ogexec) {
                HIPCHK(e, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                rc = enqueue_ode_step(e, plan, dv, prm, B, C, H, s);
                hipGraph_t g = nullptr;
                hipError_t ce = hipStreamEndCapture(s, &g);
                if (rc != PF_OK) { if (g) hipGraphDestroy(g); return rc; }
                if (ce != hipSuccess) { e->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(ce); return PF_ERR_HIP; }
                e->ograph = g;
                hipError_t ie = hipGraphInstantiate(&e->ogexec, e->ograph, nullptr, nullptr, 0);
                if (ie != hipSuccess) { drop_ode_graph(e); e->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(ie); return PF_ERR_HIP; }
                memset(&e->okey, 0, sizeof e->okey); e->okey = key;
            }
            HIPCHK(e, hipGraphLaunch(e->ogexec, s));
        } else {
            if ((rc = enqueue_ode_step(e, plan, dv, prm, B, C, H, s)) != PF_OK) return rc;
        }
        if (iter_cb && (!prm->host_cb_mask || prm->host_cb_mask[it])) {
            HIPCHK(e, hipMemcpyAsync(x_inout, b.x, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
            HIPCHK(e, hipStreamSynchronize(s));
            iter_cb(it, user);
        }
    }
    HIPCHK(e, hipMemcpyAsync(x_inout, b.x, (size_t)B * n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));
    return check_flags(e);
}

int pf_engine_profile(pf_engine* e, int enable) {
    if (!e) return PF_ERR_INVALID;
    e->profile = enable != 0;
    e->ev_used = 0; e->prof_launches = 0; e->prof_ms = 0.0; e->prof_flops = 0.0;
    return PF_OK;
}

int pf_engine_profile_read(pf_engine* e, int64_t* launches, double* ms_conv_gemm, double* flops_conv_gemm) {
    if (!e) return PF_ERR_INVALID;
    HIPCHK(e, hipDeviceSynchronize());
    FILE* dump = getenv("PNPFLOW_HIP_PROFILE_CSV") ? fopen(getenv("PNPFLOW_HIP_PROFILE_CSV"), "w") : nullptr;
    if (dump) fprintf(dump, "idx,H,W,Cout,K,nseg,taps0,stride,up,gflop,us,tflops,dma,alg_mb\n");
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->ev_pool[i].first, e->ev_pool[i].second) == hipSuccess) { e->prof_ms += ms; e->prof_launches += 1; }
        if (dump && i < e->ev_ops.size() && e->ev_ops[i]) {
            const Op& op = *e->ev_ops[i]; size_t K = 0;
            for (int j = 0; j < op.cp.nseg; ++j) K += (size_t)op.cp.seg[j].taps * op.cp.seg[j].C;
            // algorithmic HBM bytes of the launch: every operand tensor read once, the result written once, the residual read once
            double bytes = (double)op.cp.B * op.cp.H * op.cp.W * op.cp.Cout * 4.0 * (op.cp.residual ? 2.0 : 1.0);
            for (int j = 0; j < op.cp.nseg; ++j) bytes += (double)op.cp.B * op.cp.Hs * op.cp.Ws * op.cp.seg[j].C * 4.0;
            fprintf(dump, "%zu,%d,%d,%d,%zu,%d,%d,%d,%d,%.4f,%.2f,%.2f,%d,%.3f", i, op.cp.H, op.cp.W, op.cp.Cout, K, op.cp.nseg, op.cp.seg[0].taps,
                    op.stride, op.up, op.flops / 1e9, ms * 1e3, op.flops / (ms * 1e-3) / 1e12, op.use_pp == 2 ? 5 : op.use_pp ? 2 : op.dma, bytes / 1e6);
            fprintf(dump, "\n");
        }
    }
    if (dump) fclose(dump);
    e->ev_used = 0;
    if (launches) *launches = e->prof_launches;
    if (ms_conv_gemm) *ms_conv_gemm = e->prof_ms;
    if (flops_conv_gemm) *flops_conv_gemm = e->prof_flops;
    return PF_OK;
}

}  // extern "C"
