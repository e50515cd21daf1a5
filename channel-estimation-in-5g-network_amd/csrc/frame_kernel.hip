// Fused per-frame pipeline for gfx950: K1 (this file) and the launcher; K2 is frame_slab_kernel.hip.
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
//   taps_kernel  (K1)  one workgroup per frame: oscillators -> tap gains c[s][d][r][t] at the symbol starts,
//                      their tx-sums, and -- through the Gram matrix of the tap twiddles over the used bins --
//                      the frame's realised signal power mean|y|^2 WITHOUT touching the 14x599 grid.  Writes a
//                      ~7 KB "frame state" record (workspace), laid out tap-major inside a symbol so that K2 streams
//                      one tap's gains for all antenna pairs with a single scalar load.
//   frame kernel (K2)  one workgroup per frame, thread <-> subcarrier pair; assembles every output span in an LDS slab
//                      and writes it as line-aligned contiguous bursts (frame_slab_kernel.hip).
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
    float2* Ql = (float2*)(red + 32);                            // [KT*KT] Gram matrix of this frame's profile
    float2* c_g = (float2*)(rec + ST_SCAL);                      // [S][KTS][R][T]   (state_c_index)
    float2* csum_g = c_g + (size_t)S * R * T * KTS;              // [S][KTS][R]      (state_csum_index)
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
        c_g[state_c_index(s, d, r, t, R, T)] = make_float2(a * acc.x, a * acc.y);
    }
    // zero the padding taps so that readers may always load KTS entries
    for (int j = tid; j < S * R * T * (KTS - nt); j += nthr) {
        const int q = j / (KTS - nt), d = nt + j % (KTS - nt);
        c_g[state_c_index(q / (R * T), d, (q / T) % R, q % T, R, T)] = make_float2(0.f, 0.f);
    }
    __syncthreads();                                             // workgroup-scope visibility of c_g

    // ---- tx-summed taps (every TX sends the same grid, quirk Q5) ----------------------------------------
    for (int j = tid; j < KT * KT; j += nthr) Ql[j] = gram[(size_t)fd.channel_type * KT * KT + j];
    for (int j = tid; j < S * R * KTS; j += nthr) {
        const int d = j % KTS, sr = j / KTS;
        float2 acc = make_float2(0.f, 0.f);
        for (int t = 0; t < T; ++t) { const float2 v = c_g[state_c_index(sr / R, d, sr % R, t, R, T)]; acc.x += v.x; acc.y += v.y; }
        csum_l[j] = acc;
        csum_g[state_csum_index(sr / R, d, sr % R, R)] = acc;
    }
    __syncthreads();

    // ---- analytic powers: sum_k |sum_d a_d W_d[k]|^2 = sum_{d,d'} a_d conj(a_d') Q[d][d'] ----------------
    // One item per gain a_d1 (memory order of the record: coalesced, one run-time division per item); the inner sum
    // over d2 walks the same (s, r[, t]) column of the record at stride R*T.
    float p_y = 0.f, p_h = 0.f;
    for (int j = tid; j < S * R * KTS; j += nthr) {
        const int d1 = j % KTS, sr = j / KTS;
        if (d1 >= nt) continue;
        const float2 a = csum_l[j];
        float2 acc = make_float2(0.f, 0.f);                      // sum_d2 conj(b_d2) Q[d1][d2]
        for (int d2 = 0; d2 < nt; ++d2) {
            const float2 bq = csum_l[sr * KTS + d2], q = Ql[d1 * KT + d2];
            acc.x += bq.x * q.x + bq.y * q.y; acc.y += bq.x * q.y - bq.y * q.x;
        }
        p_y += a.x * acc.x - a.y * acc.y;
    }
    const int RT = R * T;
    for (int j = tid; j < S * KTS * RT; j += nthr) {
        const int d1 = (j / RT) % KTS;
        if (d1 >= nt) continue;
        const float2 a = c_g[j];
        const float2* __restrict__ col = c_g + (j - d1 * RT);    // tap 0 of this (s, r, t)
        float2 acc = make_float2(0.f, 0.f);
        for (int d2 = 0; d2 < nt; ++d2) {
            const float2 bq = col[d2 * RT], q = Ql[d1 * KT + d2];
            acc.x += bq.x * q.x + bq.y * q.y; acc.y += bq.x * q.y - bq.y * q.x;
        }
        p_h += a.x * acc.x - a.y * acc.y;
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

size_t frames_workspace_bytes(const chest_ctx* ctx, int64_t B) {
    return (size_t)B * frame_state_floats(ctx->fp.S, ctx->fp.R, ctx->fp.T) * sizeof(float);
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
    bool any_compact = false, any_full = false;                  // interpolation-plan formats present in the context
    for (const auto& p : ctx->patterns) {
        Pmax = p.dev.P > Pmax ? p.dev.P : Pmax;
        if (p.dev.plan != nullptr) { if (p.dev.pad != 0) any_compact = true; else any_full = true; }
    }
    const bool want_est = out->H_ls != nullptr || out->H_mmse != nullptr || out->metrics != nullptr;
    if (want_est && any_compact && any_full) {
        set_error("chest_frames_run: the context holds compact and full interpolation plans; a launch reads one format");
        return CHEST_ESTATE;
    }
    const size_t lds1 = ((size_t)KT * fp.T * fp.R * CHEST_NUM_OSC + (size_t)fp.S * fp.R * KTS + (size_t)KT * KT) * sizeof(float2) + 32 * sizeof(float);
    if (lds1 > 160 * 1024) { set_error("taps kernel needs %zu B of LDS (> 160 KiB): too many antennas", lds1); return CHEST_EINVAL; }
    CHEST_LDS_OPT_IN(ctx, LDS_TAPS, taps_kernel, lds1);
    chest_injected inj_v{};
    if (inj != nullptr) inj_v = *inj;
    if ((inj_v.u_angle == nullptr) != (inj_v.u_phase == nullptr) || (inj_v.n_re == nullptr) != (inj_v.n_im == nullptr)) {
        set_error("injected u_angle/u_phase and n_re/n_im must be given in pairs");
        return CHEST_EINVAL;
    }
    // optional per-kernel timing (chest_profile_begin): events on the launch stream around each of the two kernels
    // workgroup size by the work a frame holds (T * R links x taps x 20 oscillators, 14 x T * R x taps gains): a SISO frame gives a
    // 256-thread workgroup a fraction of one pass -- one wave per frame: 0.041 against 0.051 ms per 10 000 frames; 2x2: 64 / 128 / 256
    // threads 0.110 / 0.087 / 0.087 ms, 4x4: 0.415 / 0.289 / 0.208 (tools/ab_taps_threads.py; chest_tune "taps_threads")
    const int taps_threads = ctx->tune_taps_threads > 0 ? ctx->tune_taps_threads : (fp.T * fp.R <= 1 ? 64 : 256);
    const int sp1 = prof_open(ctx, st, CHEST_K_TAPS);
    hipLaunchKernelGGL(taps_kernel, dim3(B), dim3(taps_threads), lds1, st, ctx->d_fp, descs, (int)ctx->patterns.size(), ctx->d_gram, seed,
                       frame_offset, inj_v.u_angle, inj_v.u_phase, inj_v.inj_paths, (float*)workspace, ctx->d_status, out->frame_status);
    prof_close(ctx, st, sp1);
    const int sp2 = prof_open(ctx, st, CHEST_K_FRAME);
    const int rc = launch_k2_slab(ctx, st, B, descs, Pmax, seed, frame_offset, inj_v, *out, (const float*)workspace, any_compact);
    if (rc != CHEST_OK) return rc;
    prof_close(ctx, st, sp2);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
