// Stand-alone kernels for the rest of the simulator's public surface (src/channel_simulator.py):
//   MIMOChannel.apply_channel                 :313-345
//   ChannelModel.generate_time_varying_channel :84-127   (dense, arbitrary num_samples)
//   OFDMSystem.modulate / demodulate           :150-203  (LDS radix-2 FFT, CP insert/remove)
// and equalize_channel (src/baseline_estimators.py:273-312).
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

// ---------------------------------------------------------------------------------------------
// apply_channel: one workgroup per frame; pass 1 = frame power, pass 2 = y + noise.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
apply_channel_kernel(int S, int R, int T, int nsc, const float2* __restrict__ tx, const float2* __restrict__ H,
                     const float* __restrict__ snr_db, const float* __restrict__ n_re,
                     const float* __restrict__ n_im, uint64_t seed, uint64_t frame_offset, float2* __restrict__ rx) {
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int M = S * R * nsc;
    const float2* txb = tx + (size_t)b * S * T * nsc;
    const float2* Hb = H + (size_t)b * S * R * T * nsc;
    auto y_of = [&](int m) {
        const int k = m % nsc;
        const int sr = m / nsc;
        const int s = sr / R;
        float2 acc = make_float2(0.f, 0.f);
        for (int t = 0; t < T; ++t) cfma(acc, Hb[((size_t)sr * T + t) * nsc + k], txb[((size_t)s * T + t) * nsc + k]);
        return acc;
    };
    float pw = 0.f;
    for (int m = tid; m < M; m += nthr) pw += cabs2(y_of(m));
    pw = block_sum(pw, red);
    const float noise_std = sqrtf(pw / (float)M / (float)pow(10.0, (double)snr_db[b] / 10.0) * 0.5f);
    const uint64_t frame = frame_offset + (uint64_t)b;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};
    const int R2 = (R + 1) >> 1;
    for (int m = tid; m < M; m += nthr) {
        const float2 y = y_of(m);
        float2 nv;
        if (n_re != nullptr) {
            nv = make_float2(n_re[(size_t)b * M + m], n_im[(size_t)b * M + m]);
        } else {                                          // same keying as the frame pipeline
            const int k = m % nsc, sr = m / nsc, s = sr / R, r = sr - s * R;
            const u32x4 w = draw(key, STREAM_NOISE, (uint32_t)((s * R2 + (r >> 1)) * nsc + k));
            nv = (r & 1) ? gauss_pair(w.z, w.w) : gauss_pair(w.x, w.y);
        }
        rx[(size_t)b * M + m] = make_float2(fmaf(noise_std, nv.x, y.x), fmaf(noise_std, nv.y, y.y));
    }
}

int launch_apply_channel(chest_ctx* ctx, hipStream_t st, int32_t B, const float* tx, const float* H,
                         const float* snr_db, const float* n_re, const float* n_im, uint64_t seed,
                         uint64_t frame_offset, float* rx) {
    if (B <= 0) return CHEST_OK;
    if ((n_re == nullptr) != (n_im == nullptr)) { set_error("n_re/n_im must be given together"); return CHEST_EINVAL; }
    const auto& f = ctx->fp;
    hipLaunchKernelGGL(apply_channel_kernel, dim3(B), dim3(512), 0, st, f.S, f.R, f.T, f.nsc, (const float2*)tx,
                       (const float2*)H, snr_db, n_re, n_im, seed, frame_offset, (float2*)rx);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// ---------------------------------------------------------------------------------------------
// dense time-varying channel: out[n][r][t][delay] for every sample n.  grid.y = (tap, tx, rx).
// Phase in float64 turns: num_samples is unbounded, so fd*t is not small.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
tv_channel_kernel(const TapProfile prof, float fd, double inv_fs, int64_t ns, int T, int R,
                  const float* __restrict__ ua, const float* __restrict__ up, uint64_t seed, uint64_t frame,
                  float2* __restrict__ out) {
    __shared__ double oa[CHEST_NUM_OSC];
    __shared__ double ob[CHEST_NUM_OSC];
    const int combo = blockIdx.y;                       // (d*T + t)*R + r
    const int r = combo % R, t = (combo / R) % T, d = combo / (R * T);
    const int path = prof.path[d];
    if (threadIdx.x < CHEST_NUM_OSC) {
        const int src = ((path * T + t) * R + r) * CHEST_NUM_OSC + threadIdx.x;
        float a, p;
        if (ua != nullptr) { a = ua[src]; p = up[src]; }
        else {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};
            const u32x4 w = draw(key, STREAM_OSC, (uint32_t)src);
            a = u01(w.x); p = u01(w.y);
        }
        oa[threadIdx.x] = (double)(fd * cospif(2.0f * a)) * inv_fs;     // turns per sample
        ob[threadIdx.x] = (double)p;
    }
    __syncthreads();
    const int D1 = prof.max_delay + 1;
    const float amp = prof.amp[d];
    const int delay = prof.delay[d];
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < ns; n += (int64_t)gridDim.x * blockDim.x) {
        float ax = 0.f, ay = 0.f;
#pragma unroll 4
        for (int o = 0; o < CHEST_NUM_OSC; ++o) {
            const double ph = fma(oa[o], (double)n, ob[o]);
            const float f = (float)(ph - floor(ph));
            float s, c;
            sincospif(2.0f * f, &s, &c);
            ax += c; ay += s;
        }
        out[(((size_t)n * R + r) * T + t) * D1 + delay] = make_float2(amp * ax, amp * ay);
    }
}

int launch_tv_channel(chest_ctx* ctx, hipStream_t st, int32_t type, float fd, int64_t ns, int32_t T, int32_t R,
                      const float* ua, const float* up, uint64_t seed, uint64_t frame, float* out) {
    if ((unsigned)type >= CHEST_NUM_PROFILES) { set_error("unknown channel_type %d", type); return CHEST_EINVAL; }
    if (ns <= 0 || T <= 0 || R <= 0) return CHEST_OK;
    if ((ua == nullptr) != (up == nullptr)) { set_error("u_angle/u_phase must be given together"); return CHEST_EINVAL; }
    const TapProfile& prof = ctx->fp.prof[type];
    const size_t bytes = (size_t)ns * R * T * (prof.max_delay + 1) * sizeof(float2);
    CHEST_HIP(hipMemsetAsync(out, 0, bytes, st));
    const double fs = (double)ctx->cfg.fft_size * ctx->cfg.subcarrier_spacing;
    int64_t bx = (ns + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(tv_channel_kernel, dim3((unsigned)bx, (unsigned)(prof.ntaps * T * R)), dim3(256), 0, st, prof, fd,
                       1.0 / fs, ns, T, R, ua, up, seed, frame, (float2*)out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// ---------------------------------------------------------------------------------------------
// OFDM modulate / demodulate: one workgroup per OFDM symbol, radix-2 DIT FFT in LDS with a
// host-built float64->float32 twiddle table.  CP = verbatim copy of the last cp samples.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int natural_bin(int k, int half, int N) {     // used subcarrier k -> FFT bin
    const int kappa = (k < half) ? (k - half) : (k - half + 1);
    return kappa < 0 ? kappa + N : kappa;
}

template <bool kInverse>
__device__ __forceinline__ void fft_lds(float2* buf, const float2* __restrict__ tw, int N, int logN) {
    // input in bit-reversed order in buf, output natural order
    for (int stage = 0; stage < logN; ++stage) {
        const int hsz = 1 << stage;
        for (int j = threadIdx.x; j < (N >> 1); j += blockDim.x) {
            const int pos = j & (hsz - 1);
            const int i0 = ((j >> stage) << (stage + 1)) + pos;
            const int i1 = i0 + hsz;
            float2 w = tw[pos << (logN - stage - 1)];
            if (kInverse) w.y = -w.y;
            const float2 a = buf[i0], bv = cmul(buf[i1], w);
            buf[i0] = make_float2(a.x + bv.x, a.y + bv.y);
            buf[i1] = make_float2(a.x - bv.x, a.y - bv.y);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
ofdm_modulate_kernel(int N, int logN, int cp, int nsc, int half, const float2* __restrict__ tw,
                     const float2* __restrict__ sym, float2* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2* buf = (float2*)lds_raw;
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[i] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int k = threadIdx.x; k < nsc; k += blockDim.x) {
        const int bin = natural_bin(k, half, N);
        buf[__brev((unsigned)bin) >> (32 - logN)] = sym[(size_t)row * nsc + k];
    }
    __syncthreads();
    fft_lds<true>(buf, tw, N, logN);
    const float scale = 1.0f / sqrtf((float)N);          // ifft (1/N) * sqrt(N)  (:170)
    float2* o = out + (size_t)row * (N + cp);
    for (int i = threadIdx.x; i < N + cp; i += blockDim.x) {
        const float2 v = buf[i < cp ? (N - cp + i) : (i - cp)];          // CP prefix (:173-176)
        o[i] = make_float2(v.x * scale, v.y * scale);
    }
}

__global__ void __launch_bounds__(256)
ofdm_demodulate_kernel(int N, int logN, int cp, int nsc, int half, const float2* __restrict__ tw,
                       const float2* __restrict__ tin, float2* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2* buf = (float2*)lds_raw;
    const int row = blockIdx.x;
    const float2* in = tin + (size_t)row * (N + cp) + cp;                // drop CP (:195)
    for (int i = threadIdx.x; i < N; i += blockDim.x) buf[__brev((unsigned)i) >> (32 - logN)] = in[i];
    __syncthreads();
    fft_lds<false>(buf, tw, N, logN);
    const float scale = 1.0f / sqrtf((float)N);                           // (:198)
    for (int k = threadIdx.x; k < nsc; k += blockDim.x) {
        const float2 v = buf[natural_bin(k, half, N)];
        out[(size_t)row * nsc + k] = make_float2(v.x * scale, v.y * scale);
    }
}

int launch_ofdm(chest_ctx* ctx, hipStream_t st, int32_t n, const float* in, float* out, bool modulate) {
    if (n <= 0) return CHEST_OK;
    const auto& f = ctx->fp;
    int logN = 0;
    while ((1 << logN) < f.N) ++logN;
    if ((1 << logN) != f.N || f.N > 8192) { set_error("OFDM (de)modulation needs a power-of-two fft_size <= 8192"); return CHEST_EINVAL; }
    if (ctx->d_twiddle == nullptr) { set_error("twiddle table missing"); return CHEST_ESTATE; }
    const size_t lds = (size_t)f.N * sizeof(float2);
    if (modulate)
        hipLaunchKernelGGL(ofdm_modulate_kernel, dim3(n), dim3(256), lds, st, f.N, logN, f.cp, f.nsc, f.half,
                           ctx->d_twiddle, (const float2*)in, (float2*)out);
    else
        hipLaunchKernelGGL(ofdm_demodulate_kernel, dim3(n), dim3(256), lds, st, f.N, logN, f.cp, f.nsc, f.half,
                           ctx->d_twiddle, (const float2*)in, (float2*)out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// ---------------------------------------------------------------------------------------------
// equalize_channel: per resource element x = (H^H H + load I)^-1 H^H y, float64 Gauss-Jordan on the
// T x T Hermitian positive-definite system (no pivoting needed).
// ---------------------------------------------------------------------------------------------
struct cd { double x, y; };
__device__ __forceinline__ cd cdmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

template <int TT>
__global__ void __launch_bounds__(256)
equalize_kernel(int S, int R, int nsc, int64_t total, const float2* __restrict__ rx, const float2* __restrict__ H,
                double load, float2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % nsc);
    const int64_t bs = i / nsc;                          // b*S + s
    cd Gm[TT][TT], v[TT];
#pragma unroll
    for (int a = 0; a < TT; ++a) {
        v[a] = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < TT; ++c) Gm[a][c] = {a == c ? load : 0.0, 0.0};
    }
    for (int r = 0; r < R; ++r) {
        const float2 yv = rx[(bs * R + r) * nsc + k];
        cd h[TT];
#pragma unroll
        for (int a = 0; a < TT; ++a) {
            const float2 hv = H[((bs * R + r) * TT + a) * nsc + k];
            h[a] = {(double)hv.x, (double)hv.y};
        }
#pragma unroll
        for (int a = 0; a < TT; ++a) {
            const cd hc = {h[a].x, -h[a].y};
            const cd p = cdmul(hc, {(double)yv.x, (double)yv.y});
            v[a].x += p.x; v[a].y += p.y;
#pragma unroll
            for (int c = 0; c < TT; ++c) { const cd q = cdmul(hc, h[c]); Gm[a][c].x += q.x; Gm[a][c].y += q.y; }
        }
    }
#pragma unroll
    for (int p = 0; p < TT; ++p) {                        // Gauss-Jordan
        const double d2 = Gm[p][p].x * Gm[p][p].x + Gm[p][p].y * Gm[p][p].y;
        const cd inv = {Gm[p][p].x / d2, -Gm[p][p].y / d2};
#pragma unroll
        for (int c = 0; c < TT; ++c) Gm[p][c] = cdmul(Gm[p][c], inv);
        v[p] = cdmul(v[p], inv);
#pragma unroll
        for (int a = 0; a < TT; ++a) {
            if (a == p) continue;
            const cd f = Gm[a][p];
#pragma unroll
            for (int c = 0; c < TT; ++c) { const cd q = cdmul(f, Gm[p][c]); Gm[a][c].x -= q.x; Gm[a][c].y -= q.y; }
            const cd q = cdmul(f, v[p]);
            v[a].x -= q.x; v[a].y -= q.y;
        }
    }
#pragma unroll
    for (int a = 0; a < TT; ++a) out[(bs * TT + a) * nsc + k] = make_float2((float)v[a].x, (float)v[a].y);
}

int launch_equalize(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* H, int32_t method,
                    float* out) {
    if (method != 0 && method != 1) { set_error("Unknown equalization method: %d", method); return CHEST_EINVAL; }
    if (B <= 0) return CHEST_OK;
    const auto& f = ctx->fp;
    const double load = method == 0 ? 1e-8 : 0.01;        // (:297, :305-306)
    const int64_t total = (int64_t)B * f.S * f.nsc;
    const unsigned blocks = (unsigned)((total + 255) / 256);
#define CHEST_EQ(TT)                                                                                              \
    hipLaunchKernelGGL(equalize_kernel<TT>, dim3(blocks), dim3(256), 0, st, f.S, f.R, f.nsc, total, (const float2*)rx, \
                       (const float2*)H, load, (float2*)out)
    switch (f.T) {
        case 1: CHEST_EQ(1); break;
        case 2: CHEST_EQ(2); break;
        case 3: CHEST_EQ(3); break;
        case 4: CHEST_EQ(4); break;
        case 8: CHEST_EQ(8); break;
        default: set_error("equalize: num_tx %d not supported (1,2,3,4,8)", f.T); return CHEST_EINVAL;
    }
#undef CHEST_EQ
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
