// Fused per-frame pipeline for gfx950.
//
//   reference path replaced (one iteration of src/dataset_generator.py:112-125):
//     simulate_transmission          src/channel_simulator.py:348-421
//       Jakes taps at 14 symbol starts   :84-127 sampled at :301-302 (quirks Q1, Q2, Q3)
//       sparse DFT onto used bins        :300-309
//       pilot/data grid, y = Hx, AWGN    :394-404, :326-345 (Q5, Q6)
//     LSEstimator.estimate           src/baseline_estimators.py:83-117 (Q9)
//     MMSEEstimator.estimate         src/baseline_estimators.py:155-196, :232-270 (Q10, diagonal statistics)
//     evaluate_estimator sums        src/baseline_estimators.py:326-337
//
// Two launches per batch:
//   taps_kernel  (K1)  one workgroup per frame: oscillators -> tap gains c[s][r][t][d] at the symbol starts,
//                      their tx-sums, and -- through the Gram matrix of the tap twiddles over the used bins --
//                      the frame's realised signal power mean|y|^2 WITHOUT touching the 14x599 grid.  Writes a
//                      ~7 KB "frame state" record (workspace).
//   frame_kernel (K2)  one workgroup per frame, WAVE <-> OFDM symbol, lane <-> aligned subcarrier pairs.  A wave
//                      produces the rows of its symbol one after the other, so each wave writes long sequential
//                      runs (the write schedule MI355X's HBM likes: ~95 % of a flat fill, vs ~65 % when waves
//                      interleave at 1 KiB -- tools/micro/store_wave_runs.hip).  Tap gains come in through SCALAR
//                      loads (SGPR operands of the FMAs), tap twiddles from an LDS table built once per frame;
//                      everything else lives in registers and the kernel only WRITES to HBM, 16 B per lane.
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

// ================================================================================================
// K1: per-frame tap gains and analytic powers
// ================================================================================================
__global__ void __launch_bounds__(256)
taps_kernel(const FrameParams* __restrict__ fpp, const chest_frame_desc* __restrict__ descs, int npatterns,
            const float2* __restrict__ gram, uint64_t seed, uint64_t frame_offset,
            const float* __restrict__ inj_angle, const float* __restrict__ inj_phase, int inj_paths,
            float* __restrict__ state, int* __restrict__ status, uint8_t* __restrict__ frame_status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const FrameParams& fp = *fpp;                                // device-resident constant block (uniform scalar loads)
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int S = fp.S, T = fp.T, R = fp.R;
    const chest_frame_desc fd = descs[b];
    float* rec = state + (size_t)b * frame_state_floats(S, R, T);
    if ((unsigned)fd.channel_type >= CHEST_NUM_PROFILES || (unsigned)fd.pattern_slot >= (unsigned)npatterns) {
        if (tid == 0) { atomicExch(status, 1); rec[ST_VALID] = 0.f; if (frame_status) frame_status[b] = 1; }
        return;
    }
    if (tid == 0 && frame_status) frame_status[b] = 0;           // K2 overwrites it if the pattern turns out unusable
    const TapProfile& prof = fp.prof[fd.channel_type];
    const int nt = prof.ntaps;
    float2* osc = (float2*)lds_raw;                              // [nt*T*R*20]
    float2* csum_l = osc + (size_t)KT * T * R * CHEST_NUM_OSC;   // [S*R*KTS]
    float* red = (float*)(csum_l + (size_t)S * R * KTS);         // [32]
    float2* c_g = (float2*)(rec + ST_SCAL);                      // [S*R*T][KTS]
    float2* csum_g = c_g + (size_t)S * R * T * KTS;              // [S*R][KTS]
    const uint64_t frame = frame_offset + (uint64_t)b;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};

    // ---- oscillators (src/channel_simulator.py:109-110, :119): Doppler shift [Hz] and phase [turns] ----
    const int nosc = nt * T * R * CHEST_NUM_OSC;
    for (int i = tid; i < nosc; i += nthr) {
        const int n = i % CHEST_NUM_OSC;
        int q = i / CHEST_NUM_OSC;
        const int r = q % R; q /= R;
        const int t = q % T;
        const int d = q / T;
        const int src = ((prof.path[d] * T + t) * R + r) * CHEST_NUM_OSC + n;
        float ua, up;
        if (inj_angle != nullptr) {
            const size_t base = (size_t)b * inj_paths * T * R * CHEST_NUM_OSC;
            ua = inj_angle[base + src];
            up = inj_phase[base + src];
        } else {
            const u32x4 w = draw(key, STREAM_OSC, (uint32_t)src);
            ua = u01(w.x);
            up = u01(w.y);
        }
        osc[i] = make_float2(fd.doppler_hz * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(ua)), up);
    }
    __syncthreads();

    // ---- tap gains at the symbol starts (:118-125 evaluated at :301-302) -> global record -------------
    const int nc = S * R * T * nt;
    for (int j = tid; j < nc; j += nthr) {
        const int d = j % nt;
        int q = j / nt;                                          // (s*R + r)*T + t
        const int t = q % T;
        const int sr = q / T;
        const int r = sr % R;
        const int s = sr / R;
        const float ts = fp.ts[s];
        const float2* o = osc + ((d * T + t) * R + r) * CHEST_NUM_OSC;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll 4
        for (int n = 0; n < CHEST_NUM_OSC; ++n) {
            const float2 ab = o[n];
            const float2 e = cis_turns(fmaf(ab.x, ts, ab.y));
            acc.x += e.x; acc.y += e.y;
        }
        const float a = prof.amp[d];
        c_g[(size_t)q * KTS + d] = make_float2(a * acc.x, a * acc.y);
    }
    // zero the padding taps so that readers may always load KTS entries
    for (int j = tid; j < S * R * T * (KTS - nt); j += nthr) {
        const int q = j / (KTS - nt), d = nt + j % (KTS - nt);
        c_g[(size_t)q * KTS + d] = make_float2(0.f, 0.f);
    }
    __syncthreads();                                             // workgroup-scope visibility of c_g

    // ---- tx-summed taps (every TX sends the same grid, quirk Q5) ----------------------------------------
    for (int j = tid; j < S * R * KTS; j += nthr) {
        const int d = j % KTS, sr = j / KTS;
        float2 acc = make_float2(0.f, 0.f);
        for (int t = 0; t < T; ++t) { const float2 v = c_g[((size_t)sr * T + t) * KTS + d]; acc.x += v.x; acc.y += v.y; }
        csum_l[j] = acc;
        csum_g[j] = acc;
    }
    __syncthreads();

    // ---- analytic powers: sum_k |sum_d a_d W_d[k]|^2 = sum_{d,d'} a_d conj(a_d') Q[d][d'] ----------------
    const float2* Q = gram + (size_t)fd.channel_type * KT * KT;
    float p_y = 0.f, p_h = 0.f;
    for (int j = tid; j < S * R * nt * nt; j += nthr) {
        const int d2 = j % nt, d1 = (j / nt) % nt, sr = j / (nt * nt);
        const float2 a = csum_l[sr * KTS + d1], bq = csum_l[sr * KTS + d2], q = Q[d1 * KT + d2];
        const float2 ab = make_float2(a.x * bq.x + a.y * bq.y, a.y * bq.x - a.x * bq.y);   // a * conj(b)
        p_y += ab.x * q.x - ab.y * q.y;
    }
    for (int j = tid; j < S * R * T * nt * nt; j += nthr) {
        const int d2 = j % nt, d1 = (j / nt) % nt, q0 = j / (nt * nt);
        const float2 a = c_g[(size_t)q0 * KTS + d1], bq = c_g[(size_t)q0 * KTS + d2], q = Q[d1 * KT + d2];
        const float2 ab = make_float2(a.x * bq.x + a.y * bq.y, a.y * bq.x - a.x * bq.y);
        p_h += ab.x * q.x - ab.y * q.y;
    }
    p_y = block_sum(p_y, red);
    p_h = block_sum(p_h, red);
    if (tid == 0) {
        // signal power = mean|y|^2 with |x| = 1 (:337); noise_std = sqrt(P / snr / 2) (:338-340)
        const double snr_lin = pow(10.0, (double)fd.snr_db / 10.0);
        const double sp = (double)p_y / ((double)S * R * fp.nsc);
        rec[ST_NOISE_STD] = (float)sqrt(sp / snr_lin * 0.5);
        rec[ST_SIGMA2] = (float)(1.0 / snr_lin);                 // MMSE sigma^2 (src/baseline_estimators.py:174-175)
        rec[ST_SIGNAL_POWER] = (float)sp;
        rec[ST_SUM_H2] = p_h;                                    // sum |H|^2 over (s,r,t,k)
        rec[ST_SUM_VAR] = p_h - p_y / (float)T;                  // sum_t |H_t - mean_t H|^2 summed over (s,r,k)
        rec[ST_VALID] = 1.f;
        rec[6] = 0.f; rec[7] = 0.f;
    }
}

// ================================================================================================
// K2: the streaming frame kernel
// ================================================================================================
// Lane <-> aligned subcarrier pair (k0 even, k1 = k0 + 1).  A wave sweeps its symbol in chunks of 126 subcarriers
// (128 lanes' worth; lane 63 recomputes the pair lane 0 owns in the next chunk and only feeds the shuffle).
// Every output row has nsc (odd) complex64 elements, so rows alternate between 16-B aligned and 8-B shifted
// starts ("parity").  Aligned rows are stored as float4 (k0,k1); shifted rows as float4 (k1, k0+2) with k0+2
// fetched from lane+1: every store is 16-B aligned, and consecutive stores of a wave are consecutive in memory.
struct RowGeom {
    unsigned k0;      // first subcarrier of the lane (even)
    unsigned nsc;
    bool storer;      // lane < 63
    bool first;       // chunk 0, lane 0 (owns the leading single element of shifted rows)
};

__device__ __forceinline__ float2 from_next_lane(float2 v) {
    return make_float2(__shfl_down(v.x, 1, 64), __shfl_down(v.y, 1, 64));
}

// rowp: wave-uniform pointer to the row start (SGPR base + 32-bit lane offset addressing);
// par: (absolute element offset of the row start) & 1
// Output tensors are written once and never read back by the kernel: nontemporal (`nt`) stores.  With the real
// synthesis between a wave's store bursts the default policy loses ~10 % of the write rate, nt does not
// (tools/micro/store_flavour.hip: 5.36 vs 5.98 TB/s on this schedule, flat fill 6.08).
#ifndef CHEST_STORE_NT
#define CHEST_STORE_NT 1
#endif
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(float2* p, float2 a, float2 b) {
#if CHEST_STORE_NT
    const v4f_t t = {a.x, a.y, b.x, b.y};
    __builtin_nontemporal_store(t, (v4f_t*)p);
#else
    *(float4*)p = make_float4(a.x, a.y, b.x, b.y);
#endif
}
__device__ __forceinline__ void st8(float2* p, float2 a) {
#if CHEST_STORE_NT
    const v2f_t t = {a.x, a.y};
    __builtin_nontemporal_store(t, (v2f_t*)p);
#else
    *p = a;
#endif
}

__device__ __forceinline__ void store_pair(float2* __restrict__ rowp, int par, const RowGeom& g,
                                           float2 v0, float2 v1, float2 n0) {
    if (par == 0) {
        if (g.storer) {
            if (g.k0 + 1 < g.nsc) st16(rowp + g.k0, v0, v1);
            else if (g.k0 < g.nsc) st8(rowp + g.k0, v0);
        }
    } else {
        if (g.storer) {
            if (g.k0 + 2 < g.nsc) st16(rowp + 1 + g.k0, v1, n0);
            else if (g.k0 + 1 < g.nsc) st8(rowp + 1 + g.k0, v1);
        }
        if (g.first) st8(rowp, v0);
    }
}

struct FrameArgs {
    const FrameParams* fp;
    const PatternDev* pat;
    const TapProfile* prof;
    const float* rec;
    const float2* c_g;
    const float2* csum_g;
    const chest_injected* inj;
    const chest_frame_outputs* out;
    float2* tw;        // LDS [KT][tw_stride] twiddle table exp(-2 pi j kappa_k delay_d / N)
    float2* hp;        // LDS [Pmax*R] LS estimates at the pilots
    float* red;
    float* alpha;
    float2* psl;       // LDS [Pmax] pilot symbols in pilot order, written out linearly after phase W
    int tw_stride;     // float2 elements per tap row (even)
    RngKey key;
    int b;
    int par_tx, par_rx, par_h;     // parity of the frame's first element in each array
};

template <int NT>
__device__ __forceinline__ void frame_body(const FrameArgs& A, int nch) {
    const FrameParams& fp = *A.fp;
    const PatternDev& pat = *A.pat;
    const chest_injected& inj = *A.inj;
    const chest_frame_outputs& out = *A.out;
    const int S = fp.S, G = fp.G, P = pat.P, b = A.b, T = fp.T, R = fp.R;
    const unsigned nsc = (unsigned)fp.nsc;
    const int tid = threadIdx.x, lane = tid & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform -> SGPR indices, scalar loads
    const unsigned npair = (nsc + 1) >> 1;
    const float noise_std = A.rec[ST_NOISE_STD];
    float2* __restrict__ hp = A.hp;
    float2* __restrict__ txb = out.tx ? (float2*)out.tx + (size_t)b * S * T * nsc : nullptr;
    float2* __restrict__ rxb = out.rx ? (float2*)out.rx + (size_t)b * S * R * nsc : nullptr;
    float2* __restrict__ Hb = out.H ? (float2*)out.H + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ Lb = out.H_ls ? (float2*)out.H_ls + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ Mb = out.H_mmse ? (float2*)out.H_mmse + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ psb = out.pilot_symbols ? (float2*)out.pilot_symbols + (size_t)b * out.pilot_stride : nullptr;
    const float* __restrict__ usym = inj.u_sym ? inj.u_sym + (size_t)b * G : nullptr;
    const float* __restrict__ nre = inj.n_re ? inj.n_re + (size_t)b * S * R * nsc : nullptr;
    const float* __restrict__ nim = inj.n_im ? inj.n_im + (size_t)b * S * R * nsc : nullptr;
    const uint16_t* __restrict__ slots = pat.slot;
    const float4* __restrict__ tw4 = (const float4*)A.tw;
    const int tws4 = A.tw_stride >> 1;

    // ---- twiddle table exp(-2 pi j kappa_k delay_d / N), all taps x all (padded) subcarriers -------------
    {
        const float invN = 1.0f / (float)fp.N;
        const int half = fp.half, tws = A.tw_stride;
        for (int i = tid; i < NT * tws; i += blockDim.x) {
            const int d = i / tws, k = i - d * tws;
            const int kappa = (k < half) ? (k - half) : (k - half + 1);
            const float2 e = cis_turns((float)(kappa * A.prof->delay[d]) * invN);   // |kappa*delay| < 2^24: exact
            A.tw[i] = make_float2(e.x, -e.y);
        }
    }
    __syncthreads();

    // ---- phase W: wave <-> symbol; per 126-subcarrier chunk: tx rows, then per rx antenna the H rows and the rx row.
    //      The wave's stores walk a handful of short sequential streams inside its own symbol slice.
    // Work unit = (symbol, chunk); each wave takes a contiguous range of the S * nch units, i.e. whole symbols plus at most
    // two partial ones, so any number of waves per workgroup is evenly loaded (8 waves: 8.75 units each for S = 14, nch = 5).
    const int units = S * nch;
    const int u_lo = (int)((long long)wave * units / nwave), u_hi = (int)((long long)(wave + 1) * units / nwave);
    {
        uint32_t sl0 = 0, sl1 = 0;
        if (u_lo < u_hi) {                                                       // slots of the first unit
            const int s_ = u_lo / nch, j_ = u_lo - s_ * nch;
            const unsigned k_ = 2u * lane + 126u * (unsigned)j_;
            const uint16_t* __restrict__ sr_ = slots + (unsigned)s_ * nsc;
            sl0 = sr_[k_ < nsc ? k_ : nsc - 1]; sl1 = sr_[k_ + 1 < nsc ? k_ + 1 : nsc - 1];
        }
        for (int u = u_lo; u < u_hi; ++u) {
            const int s = u / nch, j = u - s * nch;
            const unsigned k0 = 2u * lane + 126u * (unsigned)j;
            const unsigned kc0 = k0 < nsc ? k0 : nsc - 1, kc1 = k0 + 1 < nsc ? k0 + 1 : nsc - 1;
            const RowGeom g{k0, nsc, lane < 63, j == 0 && lane == 0};
            const unsigned s0 = sl0, s1 = sl1;
            if (u + 1 < u_hi) {                                                  // prefetched one unit ahead
                const int sn = (u + 1) / nch, jn = (u + 1) - sn * nch;
                const unsigned kn = 2u * lane + 126u * (unsigned)jn;
                const uint16_t* __restrict__ srn = slots + (unsigned)sn * nsc;
                sl0 = srn[kn < nsc ? kn : nsc - 1]; sl1 = srn[kn + 1 < nsc ? kn + 1 : nsc - 1];
            }
            float u0, u1;
            if (usym != nullptr) {
                u0 = usym[s0]; u1 = usym[s1];
            } else {
                const u32x4 w = draw(A.key, STREAM_SYM, (uint32_t)s * npair + (k0 >> 1));
                u0 = u01(w.x); u1 = u01(w.y);
            }
            const float2 x0 = cis_unit(__builtin_amdgcn_fractf(u0));             // exp(j 2pi u) (:395,:399)
            const float2 x1 = cis_unit(__builtin_amdgcn_fractf(u1));
            if (txb != nullptr) {
                const float2 xn = from_next_lane(x0);
                for (int t = 0; t < T; ++t) {
                    const unsigned row = (unsigned)(s * T + t) * nsc;
                    store_pair(txb + row, (A.par_tx + row) & 1, g, x0, x1, xn);
                }
            }
            const bool p0 = lane < 63 && k0 < nsc && s0 < (unsigned)P, p1 = lane < 63 && k0 + 1 < nsc && s1 < (unsigned)P;
            if (psb != nullptr) {                 // staged in LDS: 838 scattered 8-B global stores per frame cost more than their bytes
                if (A.psl != nullptr) { if (p0) A.psl[s0] = x0; if (p1) A.psl[s1] = x1; }
                else { if (p0) psb[s0] = x0; if (p1) psb[s1] = x1; }
            }
            float4 w[NT];                                                        // twiddles of the pair, all taps
#pragma unroll
            for (int d = 0; d < NT; ++d) w[d] = tw4[d * tws4 + j * 63 + lane];
            for (int r = 0; r < R; ++r) {
                // sum_t H_t is always accumulated from the per-tx rows, whether or not H is stored, so that rx / H_ls /
                // H_mmse are bit-identical for every choice of requested outputs.
                float2 hs0 = make_float2(0.f, 0.f), hs1 = make_float2(0.f, 0.f);
                for (int t = 0; t < T; ++t) {
                    const float2* __restrict__ coef = A.c_g + ((s * R + r) * T + t) * KTS;
                    float2 h0 = make_float2(0.f, 0.f), h1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int d = 0; d < NT; ++d) {
                        const float2 c = coef[d];
                        cfma(h0, c, make_float2(w[d].x, w[d].y));
                        cfma(h1, c, make_float2(w[d].z, w[d].w));
                    }
                    if (Hb != nullptr) {
                        const unsigned row = (unsigned)((s * R + r) * T + t) * nsc;
                        store_pair(Hb + row, (A.par_h + row) & 1, g, h0, h1, from_next_lane(h0));
                    }
                    hs0.x += h0.x; hs0.y += h0.y; hs1.x += h1.x; hs1.y += h1.y;
                }
                const unsigned row = (unsigned)(s * R + r) * nsc;
                const float2 y0 = cmul(hs0, x0), y1 = cmul(hs1, x1);             // y = sum_t H x (:330-334)
                float2 n0v, n1v;
                if (nre != nullptr) {
                    n0v = make_float2((nre + row)[kc0], (nim + row)[kc0]);
                    n1v = make_float2((nre + row)[kc1], (nim + row)[kc1]);
                } else {
                    const u32x4 z = draw(A.key, STREAM_NOISE, (uint32_t)(s * R + r) * npair + (k0 >> 1));
                    n0v = gauss_pair(z.x, z.y);
                    n1v = gauss_pair(z.z, z.w);
                }
                const float2 r0 = make_float2(fmaf(noise_std, n0v.x, y0.x), fmaf(noise_std, n0v.y, y0.y));
                const float2 r1 = make_float2(fmaf(noise_std, n1v.x, y1.x), fmaf(noise_std, n1v.y, y1.y));
                if (rxb != nullptr) store_pair(rxb + row, (A.par_rx + row) & 1, g, r0, r1, from_next_lane(r0));
                if (p0) hp[s0 * R + r] = pilot_divide(r0, x0);                   // (src/baseline_estimators.py:110)
                if (p1) hp[s1 * R + r] = pilot_divide(r1, x1);
            }
        }
    }
    const bool want_metrics = out.metrics != nullptr;
    const bool want_est = (Lb != nullptr) || (Mb != nullptr) || want_metrics;
    if (!want_est && out.h_ls_pilots == nullptr && psb == nullptr) return;
    __syncthreads();
    if (psb != nullptr && A.psl != nullptr) {         // pilot symbols: one linear write, 16 B per lane where the row is aligned
        if ((((uintptr_t)psb) & 15) == 0) {
            for (int i = tid * 2; i + 1 < P; i += blockDim.x * 2) st16(psb + i, A.psl[i], A.psl[i + 1]);
            if ((P & 1) && tid == 0) st8(psb + P - 1, A.psl[P - 1]);
        } else {
            for (int i = tid; i < P; i += blockDim.x) st8(psb + i, A.psl[i]);
        }
    }
    if (out.h_ls_pilots != nullptr) {                // LS estimates at the pilots, [R][stride] per frame, zero padded
        float2* __restrict__ hl = (float2*)out.h_ls_pilots + (size_t)b * R * out.h_ls_stride;
        for (int i = tid; i < R * out.h_ls_stride; i += blockDim.x) {
            const int r = i / out.h_ls_stride, p = i - r * out.h_ls_stride;
            hl[i] = p < P ? hp[p * R + r] : make_float2(0.f, 0.f);
        }
    }
    if (!want_est) return;

    // ---- MMSE shrinkage per rx: p = mean|h_ls|^2, alpha = p/(p+sigma^2) (:177-190, scalar form) ---
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
        for (int i = tid; i < P; i += blockDim.x) a += cabs2(hp[i * R + r]);
        a = block_sum(a, A.red);
        if (tid == 0) {
            const double p = (double)a / (double)P;
            A.alpha[r] = (float)(p / (p + (double)A.rec[ST_SIGMA2]));
        }
    }
    __syncthreads();

    // ---- phase E: plan-driven interpolation (griddata linear/nearest as a 3-point gather), ----------
    //      tx-broadcast stores (quirk Q9), optional error sums.  Again wave <-> symbol, chunk by chunk.
    const PlanRec* __restrict__ plan = (const PlanRec*)pat.plan;
    const float invT = 1.0f / (float)T;
    float m_ls = 0.f, m_mm = 0.f;
    {
        PlanRec q0{0.f, 0.f, 0u, 0u}, q1{0.f, 0.f, 0u, 0u};
        if (u_lo < u_hi) {
            const int s_ = u_lo / nch, j_ = u_lo - s_ * nch;
            const unsigned k_ = 2u * lane + 126u * (unsigned)j_;
            const PlanRec* __restrict__ pr_ = plan + (unsigned)s_ * nsc;
            q0 = pr_[k_ < nsc ? k_ : nsc - 1]; q1 = pr_[k_ + 1 < nsc ? k_ + 1 : nsc - 1];
        }
        for (int u = u_lo; u < u_hi; ++u) {
            const int s = u / nch, j = u - s * nch;
            const unsigned k0 = 2u * lane + 126u * (unsigned)j;
            const RowGeom g{k0, nsc, lane < 63, j == 0 && lane == 0};
            const PlanRec c0 = q0, c1 = q1;
            if (u + 1 < u_hi) {                                                  // prefetched one unit ahead
                const int sn = (u + 1) / nch, jn = (u + 1) - sn * nch;
                const unsigned kn = 2u * lane + 126u * (unsigned)jn;
                const PlanRec* __restrict__ prn = plan + (unsigned)sn * nsc;
                q0 = prn[kn < nsc ? kn : nsc - 1]; q1 = prn[kn + 1 < nsc ? kn + 1 : nsc - 1];
            }
            for (int r = 0; r < R; ++r) {
                const float2 e0 = interp(c0, hp, R, r), e1 = interp(c1, hp, R, r);
                const float2 en = from_next_lane(e0);
                const float al = A.alpha[r];
                const float2 f0 = make_float2(al * e0.x, al * e0.y), f1 = make_float2(al * e1.x, al * e1.y);
                const float2 fn = make_float2(al * en.x, al * en.y);
                for (int t = 0; t < T; ++t) {
                    const unsigned row = (unsigned)((s * R + r) * T + t) * nsc;
                    const int par = (A.par_h + row) & 1;
                    if (Lb != nullptr) store_pair(Lb + row, par, g, e0, e1, en);
                    if (Mb != nullptr) store_pair(Mb + row, par, g, f0, f1, fn);
                }
                if (want_metrics) {
                    // sum_t |H_t - v|^2 = sum_t |H_t - Hbar|^2 + T |Hbar - v|^2 ; the first term is
                    // estimate-independent and comes analytically from K1 (ST_SUM_VAR).
                    const float2* __restrict__ coef = A.csum_g + (s * R + r) * KTS;
                    float2 hb0 = make_float2(0.f, 0.f), hb1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int d = 0; d < NT; ++d) {
                        const float2 c = coef[d];
                        const float4 w = tw4[d * tws4 + j * 63 + lane];
                        cfma(hb0, c, make_float2(w.x, w.y));
                        cfma(hb1, c, make_float2(w.z, w.w));
                    }
                    if (lane < 63 && k0 < nsc) {
                        const float2 m = make_float2(hb0.x * invT, hb0.y * invT);
                        m_ls += cabs2(make_float2(m.x - e0.x, m.y - e0.y));
                        m_mm += cabs2(make_float2(m.x - f0.x, m.y - f0.y));
                    }
                    if (lane < 63 && k0 + 1 < nsc) {
                        const float2 m = make_float2(hb1.x * invT, hb1.y * invT);
                        m_ls += cabs2(make_float2(m.x - e1.x, m.y - e1.y));
                        m_mm += cabs2(make_float2(m.x - f1.x, m.y - f1.y));
                    }
                }
            }
        }
    }
    if (want_metrics) {
        m_ls = block_sum(m_ls, A.red);
        m_mm = block_sum(m_mm, A.red);
        if (tid == 0) {
            float* m = out.metrics + (size_t)b * CHEST_NUM_METRICS;
            m[CHEST_M_SUM_H2] = A.rec[ST_SUM_H2];
            m[CHEST_M_SUM_ERR_LS] = A.rec[ST_SUM_VAR] + (float)T * m_ls;
            m[CHEST_M_SUM_ERR_MMSE] = A.rec[ST_SUM_VAR] + (float)T * m_mm;
            m[CHEST_M_SIGNAL_POWER] = A.rec[ST_SIGNAL_POWER]; m[CHEST_M_NOISE_STD] = noise_std;
            m[CHEST_M_ALPHA0] = A.alpha[0]; m[CHEST_M_COUNT] = (float)(S * R * T) * (float)fp.nsc;
            m[CHEST_M_RESERVED] = 0.f;
        }
    }
}

__host__ __device__ inline int frame_tw_stride(int nch) { return nch * 126 + 2; }     // float2 per tap row (even)

__global__ void __launch_bounds__(1024, 1)
frame_kernel(const FrameParams* __restrict__ fpp, const chest_frame_desc* __restrict__ descs,
             const PatternDev* __restrict__ patterns, int Pmax, uint64_t seed, uint64_t frame_offset,
             const chest_injected inj, const chest_frame_outputs out, const float* __restrict__ state,
             int nch, int xcd_remap, int* __restrict__ status, int stage_pilots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const FrameParams& fp = *fpp;
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    if (xcd_remap) {
        // Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share one).  Give each XCD a
        // contiguous frame range: its L2 then holds one pattern group's tables and its writes are one sequential region.
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);      // bijective for any grid size
    }
    const int S = fp.S, T = fp.T, R = fp.R, nsc = fp.nsc, G = fp.G;
    const float* rec = state + (size_t)b * frame_state_floats(S, R, T);
    const chest_frame_desc fd = descs[b];
    if (rec[ST_VALID] == 0.f) return;                            // K1 flagged the descriptor (uniform exit)
    const PatternDev pat = patterns[fd.pattern_slot];
    if (pat.slot == nullptr) { if (tid == 0) atomicExch(status, 1); return; }
    const bool want_est = out.H_ls != nullptr || out.H_mmse != nullptr || out.metrics != nullptr;
    if (want_est && pat.plan == nullptr) { if (tid == 0) atomicExch(status, 2); return; }
    const int P = pat.P;

    // ---- pilot mask: 16 resource elements per thread and one 16-B store, after peeling the row to 16-B alignment
    //      (G is rarely a multiple of 16, so most rows start misaligned) ---------------------------------------------
    if (out.pilot_mask != nullptr) {
        uint8_t* mrow = out.pilot_mask + (size_t)b * G;
        const int lead = min(G, (int)((16 - ((uintptr_t)mrow & 15)) & 15));
        const int nvec = (G - lead) >> 4, tail0 = lead + (nvec << 4);
        if (tid < lead) mrow[tid] = pat.slot[tid] < P ? 1 : 0;
        for (int v = tid; v < nvec; v += blockDim.x) {
            const uint16_t* sp = pat.slot + lead + (v << 4);
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[q] |= (sp[4 * q + j] < P ? 1u : 0u) << (8 * j);
            }
            *(uint4*)(mrow + lead + (v << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (tid < G - tail0) mrow[tail0 + tid] = pat.slot[tail0 + tid] < P ? 1 : 0;
    }

    FrameArgs A;
    A.fp = &fp; A.pat = &pat; A.prof = &fp.prof[fd.channel_type]; A.rec = rec;
    A.c_g = (const float2*)(rec + ST_SCAL);
    A.csum_g = A.c_g + (size_t)S * R * T * KTS;
    A.inj = &inj; A.out = &out;
    A.tw_stride = frame_tw_stride(nch);
    A.tw = (float2*)lds_raw;                                     // [KT][tw_stride]
    A.hp = A.tw + (size_t)KT * A.tw_stride;                      // [Pmax*R]
    A.red = (float*)(A.hp + (size_t)Pmax * R);                   // [32]
    A.alpha = A.red + 32;                                        // [8]
    A.psl = stage_pilots ? (float2*)(A.alpha + 8) : nullptr;     // [Pmax], or pilot symbols go straight to global memory
    const uint64_t frame = frame_offset + (uint64_t)b;
    A.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};
    A.b = b;
    A.par_tx = (int)(((size_t)b * S * T * nsc) & 1);
    A.par_rx = (int)(((size_t)b * S * R * nsc) & 1);
    A.par_h = (int)(((size_t)b * S * R * T * nsc) & 1);
    switch (A.prof->ntaps) {                                     // workgroup-uniform dispatch on the profile
        case 5: frame_body<5>(A, nch); break;                    // EPA at 15.36 MHz
        case 8: frame_body<8>(A, nch); break;                    // EVA
        default: frame_body<KT>(A, nch); break;                  // ETU / anything else (taps padded with zeros)
    }
}

size_t frames_workspace_bytes(const chest_ctx* ctx, int64_t B) {
    return (size_t)B * frame_state_floats(ctx->fp.S, ctx->fp.R, ctx->fp.T) * sizeof(float);
}

static size_t k2_lds_bytes(int nch, int Pmax, int R, bool stage_pilots) {
    return ((size_t)KT * frame_tw_stride(nch) + (size_t)Pmax * R + (stage_pilots ? (size_t)Pmax : 0)) * sizeof(float2) +
           40 * sizeof(float);                                   // tw | hp | red, alpha | psl
}

static int launch_k2(chest_ctx* ctx, hipStream_t st, int32_t B, int nch, const chest_frame_desc* descs, int Pmax,
                     uint64_t seed, uint64_t frame_offset, const chest_injected& inj, const chest_frame_outputs& out,
                     const float* workspace) {
    // Pilot symbols are staged in LDS unless that costs a resident workgroup (80 KiB each for two per CU) or does not fit.
    const size_t with = k2_lds_bytes(nch, Pmax, ctx->fp.R, true), without = k2_lds_bytes(nch, Pmax, ctx->fp.R, false);
    const bool stage = with <= 160 * 1024 && !(with > 80 * 1024 && without <= 80 * 1024);
    const size_t lds2 = stage ? with : without;
    if (lds2 > 160 * 1024) { set_error("frame kernel needs %zu B of LDS (> 160 KiB): too many antennas or pilots", lds2); return CHEST_EINVAL; }
    CHEST_LDS_OPT_IN(ctx, LDS_FRAME_ROWS, frame_kernel, lds2);
    hipLaunchKernelGGL(frame_kernel, dim3(B), dim3(ctx->block_threads), lds2, st, ctx->d_fp, descs, ctx->d_patterns,
                       Pmax, seed, frame_offset, inj, out, workspace, nch, 1, ctx->d_status, stage ? 1 : 0);
    return CHEST_OK;
}

int launch_frames(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, uint64_t seed,
                  uint64_t frame_offset, const chest_injected* inj, const chest_frame_outputs* out,
                  void* workspace, int64_t workspace_bytes) {
    if (B <= 0) return CHEST_OK;
    const FrameParams& fp = ctx->fp;
    if (workspace == nullptr || (size_t)workspace_bytes < frames_workspace_bytes(ctx, B)) {
        set_error("chest_frames_run: workspace of %lld B is smaller than the %zu B needed for %d frames",
                  (long long)workspace_bytes, frames_workspace_bytes(ctx, B), B);
        return CHEST_EINVAL;
    }
    const void* ptrs[] = {out->tx, out->rx, out->H, out->H_ls, out->H_mmse, out->pilot_symbols, workspace};
    for (const void* q : ptrs)
        if (((uintptr_t)q & 15) != 0) { set_error("chest_frames_run: output/workspace pointers must be 16-byte aligned"); return CHEST_EINVAL; }
    int Pmax = 1;
    for (const auto& p : ctx->patterns) Pmax = p.dev.P > Pmax ? p.dev.P : Pmax;
    const size_t lds1 = ((size_t)KT * fp.T * fp.R * CHEST_NUM_OSC + (size_t)fp.S * fp.R * KTS) * sizeof(float2) + 32 * sizeof(float);
    if (lds1 > 160 * 1024) { set_error("taps kernel needs %zu B of LDS (> 160 KiB): too many antennas", lds1); return CHEST_EINVAL; }
    CHEST_LDS_OPT_IN(ctx, LDS_TAPS, taps_kernel, lds1);
    chest_injected inj_v{};
    if (inj != nullptr) inj_v = *inj;
    if ((inj_v.u_angle == nullptr) != (inj_v.u_phase == nullptr) || (inj_v.n_re == nullptr) != (inj_v.n_im == nullptr)) {
        set_error("injected u_angle/u_phase and n_re/n_im must be given in pairs");
        return CHEST_EINVAL;
    }
    // optional per-kernel timing (chest_profile_begin): events on the launch stream around each of the two kernels
    const int sp1 = prof_open(ctx, st, CHEST_K_TAPS);
    hipLaunchKernelGGL(taps_kernel, dim3(B), dim3(256), lds1, st, ctx->d_fp, descs, (int)ctx->patterns.size(), ctx->d_gram, seed,
                       frame_offset, inj_v.u_angle, inj_v.u_phase, inj_v.inj_paths, (float*)workspace, ctx->d_status, out->frame_status);
    prof_close(ctx, st, sp1);
    const int sp2 = prof_open(ctx, st, CHEST_K_FRAME);
    int rc;
    if (ctx->tune_legacy_rows) {
        if (out->pilot_mask_f32 != nullptr) { set_error("pilot_mask_f32 is not available on the legacy row-store path"); return CHEST_EINVAL; }
        rc = launch_k2(ctx, st, B, (fp.nsc + 125) / 126, descs, Pmax, seed, frame_offset, inj_v, *out, (const float*)workspace);
    } else {
        rc = launch_k2_slab(ctx, st, B, descs, Pmax, seed, frame_offset, inj_v, *out, (const float*)workspace);
    }
    if (rc != CHEST_OK) return rc;
    prof_close(ctx, st, sp2);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
