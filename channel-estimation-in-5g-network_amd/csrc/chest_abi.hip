// Host side of the C ABI declared in include/chest.h: context, tap tables, pattern storage,
// argument checking, and the thin extern "C" wrappers around the kernel launchers.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "chest_internal.h"

namespace chest {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// 3GPP TS 36.101 Annex B.2 delay profiles as the reference tabulates them
// (src/channel_simulator.py:41-54): delays in ns, relative powers in dB.
struct ProfileTable { int n; double delay_ns[9]; double power_db[9]; };
static const ProfileTable kProfiles[CHEST_NUM_PROFILES] = {
    {7, {0, 30, 70, 90, 110, 190, 410}, {0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8}},
    {9, {0, 30, 150, 310, 370, 710, 1090, 1730, 2510}, {0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9}},
    {9, {0, 50, 120, 200, 230, 500, 1600, 2300, 5000}, {-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0}},
};

// Tap table of ONE channel_type code from (delay in seconds, relative power in dB) per path, exactly the arithmetic of
// ChannelModel.__init__ (src/channel_simulator.py:72-82): the frame-kernel form (distinct delays after the overwrite quirk, twiddle
// steps) and the Gram matrix of its tap twiddles over the used bins.
static int build_profile(chest_ctx* ctx, int ty, int n, const double* delays_s, const double* powers_db) {
    const chest_config& c = ctx->cfg;
    FrameParams& fp = ctx->fp;
    if (n < 1 || n > CHEST_MAX_PATHS) { set_error("a delay profile has 1..%d paths (got %d)", CHEST_MAX_PATHS, n); return CHEST_EINVAL; }
    const double fs = (double)c.fft_size * c.subcarrier_spacing;          // (:136, :379)
    double lin[CHEST_MAX_PATHS], tot = 0.0;
    int32_t dsamp[CHEST_MAX_PATHS];
    for (int l = 0; l < n; ++l) {
        lin[l] = pow(10.0, powers_db[l] / 10.0); tot += lin[l];                                       // (:75)
        // np.round(delays * fs).astype(int): round-half-even on float64 (:81)
        const double d = nearbyint(delays_s[l] * fs);
        if (!(d >= 0.0) || d > 1.0e6 || !(lin[l] >= 0.0) || !isfinite(lin[l])) { set_error("profile %d: path %d has delay %g s / power %g dB", ty, l, delays_s[l], powers_db[l]); return CHEST_EINVAL; }
        dsamp[l] = (int32_t)d;
    }
    if (!(tot > 0.0)) { set_error("profile %d: the path powers sum to 0", ty); return CHEST_EINVAL; }
    TapProfile tp;
    memset(&tp, 0, sizeof(tp));
    tp.npaths = n;
    for (int l = 0; l < n; ++l) {
        // later path with the same delay overwrites the earlier one (:125, quirk Q1)
        bool overwritten = false;
        for (int m = l + 1; m < n; ++m)
            if (dsamp[m] == dsamp[l]) overwritten = true;
        if (dsamp[l] > tp.max_delay) tp.max_delay = dsamp[l];
        if (overwritten) continue;
        if (tp.ntaps >= KT) { set_error("profile %d has more than %d distinct delays", ty, KT); return CHEST_EINVAL; }
        tp.delay[tp.ntaps] = dsamp[l];
        tp.path[tp.ntaps] = l;
        tp.amp[tp.ntaps] = (float)(sqrt(lin[l] / tot) / sqrt(2.0 * CHEST_NUM_OSC));                   // (:78, :104, :124)
        const double sa = -2.0 * M_PI * (double)dsamp[l] / (double)fp.N;
        tp.step[tp.ntaps][0] = (float)cos(sa); tp.step[tp.ntaps][1] = (float)sin(sa);
        ++tp.ntaps;
    }
    // accepted: commit
    ctx->num_paths[ty] = n;
    for (int l = 0; l < CHEST_MAX_PATHS; ++l) {
        ctx->powers_linear[ty][l] = l < n ? lin[l] / tot : 0.0;
        ctx->delay_samples[ty][l] = l < n ? dsamp[l] : 0;
    }
    fp.prof[ty] = tp;
    // Gram matrix Q[d][d'] = sum_k W_d[k] conj(W_d'[k]), W_d[k] = exp(-2 pi j kappa_k delay_d / N), float64 on the
    // host: sum_k |sum_d a_d W_d[k]|^2 = sum_{d,d'} a_d conj(a_d') Q[d][d'] gives frame powers without a grid pass.
    for (int d1 = 0; d1 < KT; ++d1)
        for (int d2 = 0; d2 < KT; ++d2) {
            double re = 0.0, im = 0.0;
            if (d1 < tp.ntaps && d2 < tp.ntaps)
                for (int k = 0; k < fp.nsc; ++k) {
                    const int kappa = (k < fp.half) ? (k - fp.half) : (k - fp.half + 1);
                    const double a = -2.0 * M_PI * (double)kappa * (double)(tp.delay[d1] - tp.delay[d2]) / (double)fp.N;
                    re += cos(a); im += sin(a);
                }
            ctx->gram_host[((size_t)ty * KT + d1) * KT + d2] = make_float2((float)re, (float)im);
        }
    return CHEST_OK;
}

static int build_tables(chest_ctx* ctx) {
    const chest_config& c = ctx->cfg;
    FrameParams& fp = ctx->fp;
    memset(&fp, 0, sizeof(fp));
    fp.N = c.fft_size; fp.cp = c.cp_length; fp.S = c.num_symbols; fp.U = c.useful_subcarriers;
    fp.half = c.useful_subcarriers / 2;
    // used = arange(dc - U//2, dc + U//2) minus dc  (src/channel_simulator.py:141-148)
    fp.nsc = 2 * fp.half - 1;
    fp.T = c.num_tx; fp.R = c.num_rx; fp.G = fp.S * fp.nsc;
    const double fs = (double)c.fft_size * c.subcarrier_spacing;          // (:136, :379)
    for (int s = 0; s < fp.S; ++s) fp.ts[s] = (float)((double)((int64_t)s * (c.fft_size + c.cp_length)) / fs);
    ctx->gram_host.assign((size_t)CHEST_NUM_PROFILES * KT * KT, make_float2(0.f, 0.f));
    for (int ty = 0; ty < CHEST_NUM_PROFILES; ++ty) {
        const ProfileTable& pt = kProfiles[ty];
        double ds[9];
        for (int l = 0; l < pt.n; ++l) ds[l] = pt.delay_ns[l] * 1e-9;     // np.array([...]) * 1e-9 (:42-53)
        const int rc = build_profile(ctx, ty, pt.n, ds, pt.power_db);
        if (rc != CHEST_OK) return rc;
    }
    return CHEST_OK;
}

static int upload_pattern_table_entry(chest_ctx* ctx, int slot) {
    if (slot >= ctx->pattern_cap) {
        int cap = ctx->pattern_cap ? ctx->pattern_cap : 16;
        while (cap <= slot) cap *= 2;
        PatternDev* nd = nullptr;
        CHEST_HIP(hipMalloc((void**)&nd, (size_t)cap * sizeof(PatternDev)));
        CHEST_HIP(hipMemset(nd, 0, (size_t)cap * sizeof(PatternDev)));
        std::vector<PatternDev> host(ctx->patterns.size());
        for (size_t i = 0; i < host.size(); ++i) host[i] = ctx->patterns[i].dev;
        if (!host.empty()) CHEST_HIP(hipMemcpy(nd, host.data(), host.size() * sizeof(PatternDev), hipMemcpyHostToDevice));
        CHEST_HIP(hipDeviceSynchronize());
        if (ctx->d_patterns) CHEST_HIP(hipFree(ctx->d_patterns));
        ctx->d_patterns = nd;
        ctx->pattern_cap = cap;
    }
    CHEST_HIP(hipMemcpy(ctx->d_patterns + slot, &ctx->patterns[slot].dev, sizeof(PatternDev), hipMemcpyHostToDevice));
    return CHEST_OK;
}

}  // namespace chest

using namespace chest;

extern "C" {

const char* chest_last_error(void) { return g_err; }
int chest_abi_version(void) { return CHEST_ABI_VERSION; }

int chest_create(const chest_config* cfg, chest_ctx** out) {
    if (cfg == nullptr || out == nullptr) { set_error("chest_create: null argument"); return CHEST_EINVAL; }
    *out = nullptr;
    if (cfg->fft_size < 8 || cfg->fft_size > 8192 || cfg->cp_length < 0 || cfg->num_symbols < 1 ||
        cfg->num_symbols > CHEST_MAX_SYMBOLS || cfg->useful_subcarriers < 4 ||
        cfg->useful_subcarriers > cfg->fft_size || cfg->num_tx < 1 || cfg->num_tx > 8 || cfg->num_rx < 1 ||
        cfg->num_rx > 8 || !(cfg->subcarrier_spacing > 0)) {
        set_error("chest_create: unsupported configuration (fft %d cp %d S %d U %d T %d R %d)", cfg->fft_size,
                  cfg->cp_length, cfg->num_symbols, cfg->useful_subcarriers, cfg->num_tx, cfg->num_rx);
        return CHEST_EINVAL;
    }
    if ((int64_t)cfg->num_symbols * (2 * (cfg->useful_subcarriers / 2) - 1) > 65535) {
        set_error("chest_create: S*nsc must be <= 65535 (16-bit resource-element slots)");
        return CHEST_EINVAL;
    }
    int ndev = 0;
    CHEST_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("chest_create: device %d of %d", cfg->device, ndev); return CHEST_EINVAL; }
    chest::DeviceGuard guard_(cfg->device);                      // the caller's current device is restored on return
    if (!guard_.ok) { set_error("hipSetDevice(%d) failed", cfg->device); return CHEST_EHIP; }
    chest_ctx* ctx = new (std::nothrow) chest_ctx();
    if (!ctx) { set_error("out of host memory"); return CHEST_ENOMEM; }
    ctx->cfg = *cfg;
    int rc = build_tables(ctx);
    if (rc != CHEST_OK) { delete ctx; return rc; }
    if (hipMalloc((void**)&ctx->d_status, sizeof(int32_t)) != hipSuccess ||
        hipMemset(ctx->d_status, 0, sizeof(int32_t)) != hipSuccess) {
        set_error("hipMalloc(status) failed");
        delete ctx;
        return CHEST_EHIP;
    }
    if (hipMalloc((void**)&ctx->d_gram, ctx->gram_host.size() * sizeof(float2)) != hipSuccess ||
        hipMemcpy(ctx->d_gram, ctx->gram_host.data(), ctx->gram_host.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("gram upload failed");
        chest_destroy(ctx);
        return CHEST_EHIP;
    }
    if (hipMalloc((void**)&ctx->d_fp, sizeof(FrameParams)) != hipSuccess ||
        hipMemcpy(ctx->d_fp, &ctx->fp, sizeof(FrameParams), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("parameter block upload failed");
        chest_destroy(ctx);
        return CHEST_EHIP;
    }
    const int N = cfg->fft_size;
    if ((N & (N - 1)) == 0) {                                    // FFT twiddles exp(-2 pi j n/N) from float64
        std::vector<float2> tw(N);
        for (int n = 0; n < N; ++n) {
            const double a = -2.0 * M_PI * (double)n / (double)N;
            tw[n] = make_float2((float)cos(a), (float)sin(a));
        }
        if (hipMalloc((void**)&ctx->d_twiddle, N * sizeof(float2)) != hipSuccess ||
            hipMemcpy(ctx->d_twiddle, tw.data(), N * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("twiddle upload failed");
            chest_destroy(ctx);
            return CHEST_EHIP;
        }
    }
    *out = ctx;
    return CHEST_OK;
}

int chest_create_grid(int32_t num_symbols, int32_t num_subcarriers, int32_t num_tx, int32_t num_rx, int32_t device, chest_ctx** out) {
    if (out == nullptr) { set_error("chest_create_grid: null argument"); return CHEST_EINVAL; }
    *out = nullptr;
    if (num_symbols < 1 || num_symbols > CHEST_MAX_SYMBOLS || num_subcarriers < 1 || num_tx < 1 || num_tx > 8 || num_rx < 1 || num_rx > 8 ||
        (int64_t)num_symbols * num_subcarriers > 65535) {
        set_error("chest_create_grid: unsupported grid (S %d nsc %d T %d R %d; S <= %d, S*nsc <= 65535, 1..8 antennas)", num_symbols,
                  num_subcarriers, num_tx, num_rx, CHEST_MAX_SYMBOLS);
        return CHEST_EINVAL;
    }
    int ndev = 0;
    CHEST_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("chest_create_grid: device %d of %d", device, ndev); return CHEST_EINVAL; }
    chest::DeviceGuard guard_(device);
    if (!guard_.ok) { set_error("hipSetDevice(%d) failed", device); return CHEST_EHIP; }
    chest_ctx* ctx = new (std::nothrow) chest_ctx();
    if (!ctx) { set_error("out of host memory"); return CHEST_ENOMEM; }
    memset(&ctx->cfg, 0, sizeof(ctx->cfg));                      // fft_size == 0 marks the context as grid-only
    ctx->cfg.num_symbols = num_symbols; ctx->cfg.num_tx = num_tx; ctx->cfg.num_rx = num_rx; ctx->cfg.device = device;
    FrameParams& fp = ctx->fp;
    memset(&fp, 0, sizeof(fp));
    fp.S = num_symbols; fp.nsc = num_subcarriers; fp.T = num_tx; fp.R = num_rx; fp.G = num_symbols * num_subcarriers;
    if (hipMalloc((void**)&ctx->d_status, sizeof(int32_t)) != hipSuccess || hipMemset(ctx->d_status, 0, sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_fp, sizeof(FrameParams)) != hipSuccess ||
        hipMemcpy(ctx->d_fp, &ctx->fp, sizeof(FrameParams), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("chest_create_grid: device allocation failed");
        chest_destroy(ctx);
        return CHEST_EHIP;
    }
    *out = ctx;
    return CHEST_OK;
}

int chest_set_profile(chest_ctx* ctx, int32_t type, int32_t num_paths, const double* delays_s, const double* powers_db) {
    if (!ctx || !delays_s || !powers_db || (unsigned)type >= CHEST_NUM_PROFILES) { set_error("chest_set_profile: bad argument (channel_type 0..%d)", CHEST_NUM_PROFILES - 1); return CHEST_EINVAL; }
    if (ctx->cfg.fft_size == 0) { set_error("chest_set_profile: a grid-only context (chest_create_grid) has no channel model"); return CHEST_ESTATE; }
    CHEST_GUARD(ctx);
    CHEST_HIP(hipDeviceSynchronize());                           // launches in flight read the tables being replaced
    const int rc = build_profile(ctx, type, num_paths, delays_s, powers_db);
    if (rc != CHEST_OK) return rc;
    CHEST_HIP(hipMemcpy(ctx->d_fp, &ctx->fp, sizeof(FrameParams), hipMemcpyHostToDevice));
    CHEST_HIP(hipMemcpy(ctx->d_gram, ctx->gram_host.data(), ctx->gram_host.size() * sizeof(float2), hipMemcpyHostToDevice));
    return CHEST_OK;
}

void chest_destroy(chest_ctx* ctx) {
    if (!ctx) return;
    chest::DeviceGuard guard_(ctx->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& p : ctx->patterns)
        if (p.block) (void)hipFree(p.block);
    for (auto& pl : ctx->pools) { if (pl.block) (void)hipFree(pl.block); if (pl.d_meta) (void)hipFree(pl.d_meta); }
    if (ctx->d_patterns) (void)hipFree(ctx->d_patterns);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
    if (ctx->d_gram) (void)hipFree(ctx->d_gram);
    if (ctx->d_fp) (void)hipFree(ctx->d_fp);
    for (auto& sp : ctx->prof_spans) { (void)hipEventDestroy((hipEvent_t)sp.a); (void)hipEventDestroy((hipEvent_t)sp.b); }
    delete ctx;
}

#define CHEST_NEEDS_OFDM(ctx, who)                                                                                          \
    if ((ctx)->cfg.fft_size == 0) { set_error("%s: a grid-only context (chest_create_grid) has no OFDM numerology / channel model", who); return CHEST_ESTATE; }

int chest_get_used_indices(chest_ctx* ctx, int32_t* num_sc, int64_t* out) {
    if (!ctx) { set_error("null context"); return CHEST_EINVAL; }
    if (num_sc) *num_sc = ctx->fp.nsc;
    if (out) {
        CHEST_NEEDS_OFDM(ctx, "chest_get_used_indices");
        const int dc = ctx->fp.N / 2, half = ctx->fp.half;
        int n = 0;
        for (int i = dc - half; i < dc + half; ++i)
            if (i != dc) out[n++] = i;
    }
    return CHEST_OK;
}

int chest_get_tap_table(chest_ctx* ctx, int32_t type, int32_t* num_paths, int32_t* delay_samples, double* powers_linear) {
    if (!ctx || (unsigned)type >= CHEST_NUM_PROFILES) { set_error("unknown channel_type %d", type); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_get_tap_table");
    if (num_paths) *num_paths = ctx->num_paths[type];
    for (int l = 0; l < ctx->num_paths[type]; ++l) {
        if (delay_samples) delay_samples[l] = ctx->delay_samples[type][l];
        if (powers_linear) powers_linear[l] = ctx->powers_linear[type][l];
    }
    return CHEST_OK;
}

// shared body of chest_pattern_set / chest_pattern_set_triangulation
static int pattern_set_impl(chest_ctx* ctx, int32_t slot, const int32_t* pilot_idx, int32_t P, const uint16_t* plan_idx,
                            const float* plan_w, const int32_t* simplices, int32_t ns) {
    const int G = ctx->fp.G;
    if (P < 1 || P > G) { set_error("chest_pattern_set: num_pilots %d out of range (1..%d)", P, G); return CHEST_EINVAL; }
    for (int i = 0; i < P; ++i) {
        if (pilot_idx[i] < 0 || pilot_idx[i] >= G || (i > 0 && pilot_idx[i] <= pilot_idx[i - 1])) {
            set_error("chest_pattern_set: pilot_idx must be sorted, unique and inside the %d-element grid", G);
            return CHEST_EINVAL;
        }
    }
    if (plan_idx)
        for (size_t i = 0; i < (size_t)3 * G; ++i)
            if (plan_idx[i] >= P) { set_error("chest_pattern_set: plan index %u >= num_pilots %d", plan_idx[i], P); return CHEST_EINVAL; }
    if (simplices)
        for (size_t i = 0; i < (size_t)3 * ns; ++i)
            if (simplices[i] < 0 || simplices[i] >= P) {
                set_error("chest_pattern_set_triangulation: simplex vertex %d outside 0..%d", simplices[i], P - 1);
                return CHEST_EINVAL;
            }
    const bool with_plan = plan_idx != nullptr || simplices != nullptr;
    CHEST_GUARD(ctx);
    // pilot bit table (chest_internal.h PatternDev::bits): i-th pilot <-> i-th True of the row-major mask (insert_pilots,
    // src/channel_simulator.py:249-252); data symbols follow in row-major order -- both ordinals follow from the ranks stored here.
    const int W = pattern_bit_words(G);
    std::vector<uint2> bits((size_t)W, make_uint2(0u, 0u));
    for (int i = 0; i < P; ++i) bits[pilot_idx[i] >> 5].x |= 1u << (pilot_idx[i] & 31);
    {
        uint32_t before = 0;
        for (int w = 0; w < W; ++w) { bits[w].y = before; before += (uint32_t)__builtin_popcount(bits[w].x); }
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    struct Rec { float w0, w1; uint32_t i01, i2v; };
    std::vector<Rec> recs;
    if (plan_idx) {
        recs.resize(G);
        for (int g = 0; g < G; ++g) {
            const float w0 = plan_w[g], w1 = plan_w[(size_t)G + g], w2 = plan_w[2 * (size_t)G + g];
            const uint32_t i0 = plan_idx[g], i1 = plan_idx[(size_t)G + g], i2 = plan_idx[2 * (size_t)G + g];
            const uint32_t valid = !(w0 == 0.0f && w1 == 0.0f && w2 == 0.0f);     // NaN weights stay "valid"
            recs[g] = Rec{w0, w1, i0 | (i1 << 16), i2 | (valid << 16)};
        }
    }
    // block layout: pilot bit table uint2[W] | pilot_idx[P] | simplices[ns][3] + their packed form (triangulation path) | plan[G] records (last, so the
    // triangulation path uploads everything before it in one copy and the device fills the plan)
    const size_t o_slot = 0;
    const size_t o_pil = up16(o_slot + (size_t)W * sizeof(uint2));
    const size_t o_simp = up16(o_pil + (size_t)P * 4);
    const size_t o_pack = up16(o_simp + (simplices ? (size_t)ns * 12 : 0));            // packed simplices, written by the device
    const size_t o_plan = (o_pack + (simplices ? (size_t)ns * 16 : 0) + 255) & ~(size_t)255;
    const size_t total = up16(o_plan + (with_plan ? (size_t)G * sizeof(Rec) : 0));
    const size_t upload = plan_idx ? total : o_pack;
    unsigned char* blk = nullptr;
    CHEST_HIP(hipMalloc((void**)&blk, total));
    std::vector<unsigned char> host(upload, 0);
    memcpy(host.data() + o_slot, bits.data(), (size_t)W * sizeof(uint2));
    memcpy(host.data() + o_pil, pilot_idx, (size_t)P * 4);
    if (simplices) memcpy(host.data() + o_simp, simplices, (size_t)ns * 12);
    if (plan_idx) memcpy(host.data() + o_plan, recs.data(), (size_t)G * sizeof(Rec));
    if (hipMemcpy(blk, host.data(), upload, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(blk);
        set_error("pattern upload failed");
        return CHEST_EHIP;
    }
    if (simplices) {
        // point location + barycentric weights on the device (plan_kernels.hip); synchronous like the uploads around it
        int rc = launch_plan_locate(ctx, nullptr, (const int32_t*)(blk + o_pil), (const int32_t*)(blk + o_simp), ns, blk + o_pack,
                                    blk + o_plan);
        if (rc == CHEST_OK && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("plan construction failed"); rc = CHEST_EHIP; }
        if (rc != CHEST_OK) { (void)hipFree(blk); return rc; }
    }
    if ((size_t)slot >= ctx->patterns.size()) ctx->patterns.resize(slot + 1);
    PatternHost& ph = ctx->patterns[slot];
    if (ph.block) { CHEST_HIP(hipDeviceSynchronize()); CHEST_HIP(hipFree(ph.block)); }
    ph.block = blk;
    ph.dev = PatternDev{};                                       // (a slot that held a compact plan must not keep its `pad`)
    ph.dev.bits = (const uint2*)(blk + o_slot);
    ph.dev.plan = with_plan ? (const void*)(blk + o_plan) : nullptr;
    ph.dev.pilot_idx = (const int32_t*)(blk + o_pil);
    ph.dev.P = P;
    // table entries between the old size and `slot` stay zero (bits == nullptr -> "not set")
    return upload_pattern_table_entry(ctx, slot);
}

int chest_pattern_set(chest_ctx* ctx, int32_t slot, const int32_t* pilot_idx, int32_t P, const uint16_t* plan_idx,
                      const float* plan_w) {
    if (!ctx || slot < 0 || slot > (1 << 20) || pilot_idx == nullptr) { set_error("chest_pattern_set: bad argument"); return CHEST_EINVAL; }
    if ((plan_idx == nullptr) != (plan_w == nullptr)) { set_error("plan_idx/plan_w must be given together"); return CHEST_EINVAL; }
    return pattern_set_impl(ctx, slot, pilot_idx, P, plan_idx, plan_w, nullptr, 0);
}

int chest_pattern_set_triangulation(chest_ctx* ctx, int32_t slot, const int32_t* pilot_idx, int32_t P, const int32_t* simplices,
                                    int32_t num_simplices) {
    if (!ctx || slot < 0 || slot > (1 << 20) || pilot_idx == nullptr || simplices == nullptr || num_simplices < 1 ||
        num_simplices > (1 << 22)) {
        set_error("chest_pattern_set_triangulation: bad argument");
        return CHEST_EINVAL;
    }
    return pattern_set_impl(ctx, slot, pilot_idx, P, nullptr, nullptr, simplices, num_simplices);
}

int chest_pattern_get_plan(chest_ctx* ctx, int32_t slot, uint16_t* plan_idx, float* plan_w) {
    if (!ctx || slot < 0 || (size_t)slot >= ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    if (ctx->patterns[slot].dev.plan == nullptr) { set_error("pattern slot %d has no interpolation plan", slot); return CHEST_ESTATE; }
    if (!plan_idx || !plan_w) { set_error("chest_pattern_get_plan: null output"); return CHEST_EINVAL; }
    const int G = ctx->fp.G;
    struct Rec { float w0, w1; uint32_t i01, i2v; };
    std::vector<Rec> recs(G);
    CHEST_GUARD(ctx);
    CHEST_HIP(hipDeviceSynchronize());
    const chest::PatternDev& pd = ctx->patterns[slot].dev;
    if (pd.pad != 0) {
        // compact plan: expand the triangle records at every element exactly as the frame kernel does (chest_device.h: plan_rec)
        const int ntab = pd.pad / 16 < 2 * 4096 ? pd.pad / 16 : 2 * 4096;     // the record table ends before the per-element ids
        std::vector<uint32_t> tab((size_t)ntab * 4);
        std::vector<uint16_t> tri(G);
        CHEST_HIP(hipMemcpy(tab.data(), pd.plan, tab.size() * 4, hipMemcpyDeviceToHost));
        CHEST_HIP(hipMemcpy(tri.data(), (const unsigned char*)pd.plan + pd.pad, (size_t)G * 2, hipMemcpyDeviceToHost));
        const int nsc = ctx->fp.nsc;
        for (int g = 0; g < G; ++g) {
            const uint32_t* d = &tab[(size_t)(tri[g] < ntab ? tri[g] : 0) * 4];
            const int s = g / nsc, k = g - s * nsc;
            const int pa = (int)(d[1] >> 16), pb = (int)(d[2] & 0xFFFFu), pc = (int)(d[2] >> 16);
            const int ax = pa >> 11, ay = pa & 0x7FF, bx = pb >> 11, by = pb & 0x7FF, cx = pc >> 11, cy = pc & 0x7FF;
            const int o0 = (cx - bx) * (k - by) - (cy - by) * (s - bx), o1 = (ax - cx) * (k - cy) - (ay - cy) * (s - cx);
            float rden;
            memcpy(&rden, &d[3], 4);
            recs[g] = Rec{(float)o0 * rden, (float)o1 * rden, d[0], (d[1] & 0xFFFFu) | (pa != 0xFFFF ? 1u << 16 : 0u)};
        }
    } else
    CHEST_HIP(hipMemcpy(recs.data(), pd.plan, (size_t)G * sizeof(Rec), hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g) {
        const Rec& r = recs[g];
        const bool valid = (r.i2v >> 16) != 0;
        plan_idx[g] = (uint16_t)(r.i01 & 0xFFFFu); plan_idx[(size_t)G + g] = (uint16_t)(r.i01 >> 16);
        plan_idx[2 * (size_t)G + g] = (uint16_t)(r.i2v & 0xFFFFu);
        plan_w[g] = r.w0; plan_w[(size_t)G + g] = r.w1;
        plan_w[2 * (size_t)G + g] = valid ? 1.0f - r.w0 - r.w1 : 0.0f;      // as the kernels evaluate it
    }
    return CHEST_OK;
}

int chest_pattern_num_pilots(chest_ctx* ctx, int32_t slot) {
    if (!ctx || slot < 0 || (size_t)slot >= ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    return ctx->patterns[slot].dev.P;
}

int chest_get_status(chest_ctx* ctx, int32_t* status) {
    if (!ctx || !status) { set_error("null argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    CHEST_HIP(hipMemcpy(status, ctx->d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (*status != 0) CHEST_HIP(hipMemset(ctx->d_status, 0, sizeof(int32_t)));
    return CHEST_OK;
}

int64_t chest_frames_workspace_bytes(chest_ctx* ctx, int32_t num_frames) {
    if (!ctx || num_frames < 0) return 0;
    return (int64_t)frames_workspace_bytes(ctx, num_frames);
}

int chest_profile_begin(chest_ctx* ctx, int32_t capacity) {
    if (!ctx || capacity < 0) { set_error("chest_profile_begin: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    for (auto& sp : ctx->prof_spans) { (void)hipEventDestroy((hipEvent_t)sp.a); (void)hipEventDestroy((hipEvent_t)sp.b); }
    ctx->prof_spans.clear();
    ctx->prof_used = 0;
    for (int i = 0; i < capacity * 8; ++i) {
        hipEvent_t a = nullptr, b = nullptr;
        CHEST_HIP(hipEventCreate(&a));
        CHEST_HIP(hipEventCreate(&b));
        ctx->prof_spans.push_back({0, a, b});
    }
    return CHEST_OK;
}

int chest_profile_read_kernel(chest_ctx* ctx, int32_t kernel_id, float* ms, int32_t* count) {
    if (!ctx || !ms || !count) { set_error("chest_profile_read_kernel: null argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    double t = 0.0;
    int n = 0;
    for (int i = 0; i < ctx->prof_used; ++i) {
        const auto& sp = ctx->prof_spans[i];
        if (sp.id != kernel_id) continue;
        CHEST_HIP(hipEventSynchronize((hipEvent_t)sp.b));
        float a = 0.f;
        CHEST_HIP(hipEventElapsedTime(&a, (hipEvent_t)sp.a, (hipEvent_t)sp.b));
        t += a; ++n;
    }
    *count = n;
    *ms = n ? (float)(t / n) : 0.f;
    return CHEST_OK;
}

int chest_profile_read(chest_ctx* ctx, float* taps_ms, float* frame_ms, int32_t* count) {
    if (!ctx || !taps_ms || !frame_ms || !count) { set_error("chest_profile_read: null argument"); return CHEST_EINVAL; }
    int32_t n1 = 0;
    int rc = chest_profile_read_kernel(ctx, CHEST_K_TAPS, taps_ms, &n1);
    if (rc == CHEST_OK) rc = chest_profile_read_kernel(ctx, CHEST_K_FRAME, frame_ms, count);
    ctx->prof_used = 0;
    return rc;
}

int chest_tune(chest_ctx* ctx, const char* key, int32_t value) {
    if (!ctx || !key) { set_error("chest_tune: null argument"); return CHEST_EINVAL; }
    if (strcmp(key, "frame_rx_group") == 0) { ctx->tune_rx_group = value; return CHEST_OK; }
    if (strcmp(key, "frame_two_slabs") == 0) { ctx->tune_two_slabs = value < 0 ? -1 : 0; return CHEST_OK; }
    if (strcmp(key, "gemm_f32") == 0) { ctx->tune_gemm_f32 = value ? 1 : 0; return CHEST_OK; }
    if (strcmp(key, "interp_mode") == 0) { ctx->tune_interp_mode = (value == 2 || value == 3) ? value : 0; return CHEST_OK; }
    if (strcmp(key, "interp_touch") == 0) { ctx->tune_interp_touch = value; return CHEST_OK; }
    if (strcmp(key, "interp_xcd") == 0) { ctx->tune_interp_xcd = value < 0 ? -1 : 0; return CHEST_OK; }
    if (strcmp(key, "frame_store_extra") == 0) { ctx->tune_store_extra = (value >= -1 && value <= 4) ? value : 0; return CHEST_OK; }
    if (strcmp(key, "frame_store_waves") == 0) { ctx->tune_store_waves = (value >= 0 && value <= 16) ? value : 0; return CHEST_OK; }
    if (strcmp(key, "frame_min_waves") == 0) { ctx->tune_min_waves = value; return CHEST_OK; }
    if (strcmp(key, "taps_threads") == 0) { ctx->tune_taps_threads = (value == 64 || value == 128 || value == 256) ? value : 0; return CHEST_OK; }
    if (strcmp(key, "pattern_extra_lds") == 0) { ctx->tune_pattern_extra_lds = value < 0 ? 0 : value; return CHEST_OK; }
    if (strcmp(key, "pattern_flip_cap") == 0) { ctx->tune_pattern_flip_cap = value < 0 ? 0 : value; return CHEST_OK; }
    if (strcmp(key, "pattern_stop_after") == 0) { ctx->tune_pattern_stop_after = value; return CHEST_OK; }
    if (strcmp(key, "pattern_solo_at") == 0) { ctx->tune_pattern_solo_at = value < 0 ? 0 : value; return CHEST_OK; }
    set_error("chest_tune: unknown key '%s'", key);
    return CHEST_EINVAL;
}

int chest_frames_run(chest_ctx* ctx, void* stream, int32_t B, const chest_frame_desc* descs, uint64_t seed,
                     uint64_t frame_offset, const chest_injected* inj, const chest_frame_outputs* out,
                     void* workspace, int64_t workspace_bytes) {
    if (!ctx || !out || (B > 0 && !descs)) { set_error("chest_frames_run: null argument"); return CHEST_EINVAL; }
    if (B < 0) { set_error("chest_frames_run: negative frame count"); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_frames_run");
    if (ctx->patterns.empty()) { set_error("chest_frames_run: no pilot pattern uploaded"); return CHEST_ESTATE; }
    if (out->pilot_symbols != nullptr) {
        int Pmax = 0;
        for (const auto& p : ctx->patterns) Pmax = p.dev.P > Pmax ? p.dev.P : Pmax;
        if (out->pilot_stride < Pmax) { set_error("pilot_stride %d < largest pattern %d", out->pilot_stride, Pmax); return CHEST_EINVAL; }
    }
    if (out->h_ls_pilots != nullptr) {
        int Pmax = 0;
        for (const auto& p : ctx->patterns) Pmax = p.dev.P > Pmax ? p.dev.P : Pmax;
        if (out->h_ls_stride < Pmax) { set_error("h_ls_stride %d < largest pattern %d", out->h_ls_stride, Pmax); return CHEST_EINVAL; }
    }
    CHEST_GUARD(ctx);
    return launch_frames(ctx, (hipStream_t)stream, B, descs, seed, frame_offset, inj, out, workspace, workspace_bytes);
}

int chest_estimate_batch(chest_ctx* ctx, void* stream, int32_t B, const float* rx4d, const float* tx_pilots,
                         int32_t pilot_stride, int32_t slot, int32_t mmse, float snr_db, float* H_est) {
    if (!ctx || (B > 0 && (!rx4d || !tx_pilots || !H_est))) { set_error("chest_estimate_batch: null argument"); return CHEST_EINVAL; }
    if (B < 0 || pilot_stride < 0) { set_error("chest_estimate_batch: negative size"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_estimate(ctx, (hipStream_t)stream, B, rx4d, tx_pilots, pilot_stride, slot, mmse, snr_db, H_est);
}

int chest_at_pilots(chest_ctx* ctx, void* stream, int32_t n, const float* rx, const float* tx, int32_t mmse,
                    float snr_db, float* out) {
    if (!ctx || n < 0 || (n > 0 && (!rx || !tx || !out))) { set_error("chest_at_pilots: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_at_pilots(ctx, (hipStream_t)stream, n, rx, tx, mmse, snr_db, out);
}

int chest_error_sums(chest_ctx* ctx, void* stream, int64_t n, const float* a, const float* b, double* out) {
    if (!ctx || n < 0 || !out || (n > 0 && (!a || !b))) { set_error("chest_error_sums: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_error_sums(ctx, (hipStream_t)stream, n, a, b, out);
}

int chest_apply_channel(chest_ctx* ctx, void* stream, int32_t B, const float* tx, const float* H, const float* snr_db,
                        const float* n_re, const float* n_im, uint64_t seed, uint64_t frame_offset, float* rx) {
    if (!ctx || B < 0 || (B > 0 && (!tx || !H || !snr_db || !rx))) { set_error("chest_apply_channel: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_apply_channel(ctx, (hipStream_t)stream, B, tx, H, snr_db, n_re, n_im, seed, frame_offset, rx);
}

int chest_time_varying_channel(chest_ctx* ctx, void* stream, int32_t type, float fd, int64_t ns, int32_t T, int32_t R,
                               const float* ua, const float* up, uint64_t seed, uint64_t frame, float* out) {
    if (!ctx || ns < 0 || T < 0 || R < 0 || (ns > 0 && !out)) { set_error("chest_time_varying_channel: bad argument"); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_time_varying_channel");
    CHEST_GUARD(ctx);
    return launch_tv_channel(ctx, (hipStream_t)stream, type, fd, ns, T, R, ua, up, seed, frame, out);
}

int chest_ofdm_modulate(chest_ctx* ctx, void* stream, int32_t n, const float* sym, float* out) {
    if (!ctx) { set_error("chest_ofdm_modulate: null context"); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_ofdm_modulate");
    if (n < 0 || (n > 0 && (!sym || !out))) { set_error("chest_ofdm_modulate: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_ofdm(ctx, (hipStream_t)stream, n, sym, out, true);
}

int chest_ofdm_demodulate(chest_ctx* ctx, void* stream, int32_t n, const float* tin, float* out) {
    if (!ctx) { set_error("chest_ofdm_demodulate: null context"); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_ofdm_demodulate");
    if (n < 0 || (n > 0 && (!tin || !out))) { set_error("chest_ofdm_demodulate: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_ofdm(ctx, (hipStream_t)stream, n, tin, out, false);
}

int chest_equalize(chest_ctx* ctx, void* stream, int32_t B, const float* rx, const float* H, int32_t method, float* out) {
    if (!ctx || B < 0 || (B > 0 && (!rx || !H || !out))) { set_error("chest_equalize: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_equalize(ctx, (hipStream_t)stream, B, rx, H, method, out);
}

int chest_tdl_apply(chest_ctx* ctx, void* stream, int32_t type, int64_t ns, int32_t T, int32_t R, const float* x, const float* h,
                    float* y) {
    if (!ctx || ns < 0 || T < 0 || R < 0 || (ns > 0 && (!x || !h || !y))) { set_error("chest_tdl_apply: bad argument"); return CHEST_EINVAL; }
    CHEST_NEEDS_OFDM(ctx, "chest_tdl_apply");
    CHEST_GUARD(ctx);
    return launch_tdl(ctx, (hipStream_t)stream, type, ns, T, R, x, h, y);
}

int chest_pack_ml_inputs(chest_ctx* ctx, void* stream, int32_t B, const float* rx, const float* H_ls, const float* H_true,
                         const uint8_t* mask, int32_t normalize, float* inputs, float* targets) {
    if (!ctx || B < 0 || (B > 0 && (!rx || !H_ls || !H_true || !mask || !inputs || !targets))) {
        set_error("chest_pack_ml_inputs: bad argument");
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    return launch_pack_ml(ctx, (hipStream_t)stream, B, rx, H_ls, H_true, mask, normalize, inputs, targets);
}

int64_t chest_ml_stats_workspace_bytes(void) { return ml_stats_workspace_bytes(); }

int chest_ml_dataset_stats(chest_ctx* ctx, void* stream, int32_t B, const float* rx, const float* H_ls, const float* H_true,
                           double* stats, void* workspace, int64_t workspace_bytes) {
    if (!ctx || B < 1 || !rx || !H_ls || !H_true || !stats || !workspace) { set_error("chest_ml_dataset_stats: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_ml_stats(ctx, (hipStream_t)stream, B, rx, H_ls, H_true, stats, workspace, workspace_bytes);
}

int chest_pack_ml_training(chest_ctx* ctx, void* stream, int32_t B, const float* rx, const float* H_ls, const float* H_true,
                           const void* mask, int32_t mask_is_f32, int32_t mode, const double* stats, float* inputs, float* targets,
                           float* mask_out) {
    if (!ctx || B < 0 || mode < 0 || mode > 2 || (mode != 0 && !stats) ||
        (B > 0 && (!rx || !H_ls || !H_true || !mask || !inputs || !targets))) {
        set_error("chest_pack_ml_training: bad argument (mode 0 raw / 1 dataset mean+std / 2 per-tensor |x| std; stats needed for 1, 2)");
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    return launch_pack_train(ctx, (hipStream_t)stream, B, rx, H_ls, H_true, mask, mask_is_f32, mode, stats, inputs, targets, mask_out);
}

int chest_qam_constellation(int32_t M, double* points, int32_t* gray_map) { return qam_constellation_host(M, points, gray_map); }

int chest_qam_modulate(chest_ctx* ctx, void* stream, int32_t M, int64_t n, const uint8_t* bits, int32_t wide, void* symbols) {
    if (!ctx || n < 0 || (n > 0 && (!bits || !symbols)) || ((uintptr_t)bits & 3)) {
        set_error("chest_qam_modulate: bad argument (bits must be 4-byte aligned)");
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    return launch_qam_mod(ctx, (hipStream_t)stream, M, n, bits, wide, symbols);
}

int chest_qam_demodulate(chest_ctx* ctx, void* stream, int32_t M, int64_t n, int32_t wide, const void* symbols, uint8_t* bits) {
    if (!ctx || n < 0 || (n > 0 && (!bits || !symbols)) || ((uintptr_t)bits & 3)) {
        set_error("chest_qam_demodulate: bad argument (bits must be 4-byte aligned)");
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    return launch_qam_demod(ctx, (hipStream_t)stream, M, n, wide, symbols, bits);
}

int chest_bit_errors(chest_ctx* ctx, void* stream, int64_t n, const uint8_t* a, const uint8_t* b, uint64_t* count) {
    if (!ctx || n < 0 || !count || (n > 0 && (!a || !b))) { set_error("chest_bit_errors: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    return launch_bit_errors(ctx, (hipStream_t)stream, n, a, b, count);
}

}  // extern "C"
