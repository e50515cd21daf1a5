// Kernels on either side of the hot path (SURVEY 8f "next" rows):
//   tdl_kernel      time-domain tapped-delay-line link: y[r][n] = sum_t sum_l h[n][r][t][d_l] x[t][n - d_l].
//                   The reference's pipeline never does this (it multiplies in the frequency domain, quirks Q2/Q4); the
//                   kernel closes the loop OFDMSystem.modulate -> multipath channel (ChannelModel.generate_time_varying_channel,
//                   src/channel_simulator.py:84-127) -> OFDMSystem.demodulate, exposing ISI/ICI when max delay > CP.
//   pack_ml_kernel  prepare_ml_inputs (src/dataset_generator.py:183-227) for a whole batch on the GPU:
//                   inputs [B][S][nsc][5] = (rx re, rx im, H_ls re, H_ls im, mask) of the first rx/tx pair, targets
//                   [B][S][nsc][2] = H_true of that pair, each optionally divided by its population std + 1e-8.
//   ml_stats_* / pack_train_kernel   the TRAINING-side packing of the same records: train.ChannelDataset
//                   (src/train.py:22-94: dataset-level mean / std per tensor, channels-first (5, S, nsc) inputs and (2, S, nsc)
//                   targets) and run_phase4_training.ChannelEstimationDataset (run_phase4_training.py:60-112: per-tensor std of
//                   |x| + 1e-8, no mean), so that a training loop consumes device tensors without the .npz round trip.
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

__global__ void __launch_bounds__(256)
tdl_kernel(const TapProfile prof, int64_t ns, int T, int R, const float2* __restrict__ x, const float2* __restrict__ h,
           float2* __restrict__ y) {
    const int r = blockIdx.y;
    const int D1 = prof.max_delay + 1;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < ns; n += (int64_t)gridDim.x * blockDim.x) {
        float2 acc = make_float2(0.f, 0.f);
        for (int t = 0; t < T; ++t) {
            const float2* hp = h + (((size_t)n * R + r) * T + t) * D1;
            const float2* xp = x + (size_t)t * ns;
            for (int l = 0; l < prof.ntaps; ++l) {
                const int d = prof.delay[l];
                if (n >= d) cfma(acc, hp[d], xp[n - d]);
            }
        }
        y[(size_t)r * ns + n] = acc;
    }
}

int launch_tdl(chest_ctx* ctx, hipStream_t st, int32_t type, int64_t ns, int32_t T, int32_t R, const float* x, const float* h,
               float* y) {
    if ((unsigned)type >= CHEST_NUM_PROFILES) { set_error("unknown channel_type %d", type); return CHEST_EINVAL; }
    if (ns <= 0 || T <= 0 || R <= 0) return CHEST_OK;
    int64_t bx = (ns + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(tdl_kernel, dim3((unsigned)bx, (unsigned)R), dim3(256), 0, st, ctx->fp.prof[type], ns, T, R,
                       (const float2*)x, (const float2*)h, (float2*)y);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// one workgroup per frame; two passes over the (L2-resident) first-pair slices: moments, then normalised writes
__global__ void __launch_bounds__(256)
pack_ml_kernel(int S, int R, int T, int nsc, const float2* __restrict__ rx, const float2* __restrict__ Hls,
               const float2* __restrict__ Htrue, const uint8_t* __restrict__ mask, int normalize,
               float* __restrict__ inputs, float* __restrict__ targets) {
    __shared__ float red[32];
    const int b = blockIdx.x, G = S * nsc;
    const float2* rxb = rx + (size_t)b * S * R * nsc;            // [S][R][nsc]      -> [:, 0, :]
    const float2* lsb = Hls + (size_t)b * S * R * T * nsc;       // [S][R][T][nsc]   -> [:, 0, 0, :]
    const float2* htb = Htrue + (size_t)b * S * R * T * nsc;
    float si = 0.f, si2 = 0.f, st = 0.f, st2 = 0.f;
    if (normalize) {
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const int s = g / nsc, k = g - s * nsc;
            const float2 a = rxb[(size_t)s * R * nsc + k], l = lsb[(size_t)s * R * T * nsc + k], h = htb[(size_t)s * R * T * nsc + k];
            si += a.x + a.y + l.x + l.y;
            si2 += a.x * a.x + a.y * a.y + l.x * l.x + l.y * l.y;
            st += h.x + h.y;
            st2 += h.x * h.x + h.y * h.y;
        }
        si = block_sum(si, red); si2 = block_sum(si2, red); st = block_sum(st, red); st2 = block_sum(st2, red);
    }
    float inv_i = 1.f, inv_t = 1.f;
    if (normalize) {
        // np.std (population) over the 4*G input values / the 2*G target values, then x / (std + 1e-8)  (:221-225)
        const double ni = 4.0 * G, nt = 2.0 * G;
        const double mi = (double)si / ni, mt = (double)st / nt;
        const double vi = (double)si2 / ni - mi * mi, vt = (double)st2 / nt - mt * mt;
        inv_i = (float)(1.0 / (sqrt(vi > 0 ? vi : 0.0) + 1e-8));
        inv_t = (float)(1.0 / (sqrt(vt > 0 ? vt : 0.0) + 1e-8));
    }
    float* ib = inputs + (size_t)b * G * 5;
    float* tb = targets + (size_t)b * G * 2;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const int s = g / nsc, k = g - s * nsc;
        const float2 a = rxb[(size_t)s * R * nsc + k], l = lsb[(size_t)s * R * T * nsc + k], h = htb[(size_t)s * R * T * nsc + k];
        float* o = ib + (size_t)g * 5;
        o[0] = a.x * inv_i; o[1] = a.y * inv_i; o[2] = l.x * inv_i; o[3] = l.y * inv_i;
        o[4] = mask[(size_t)b * G + g] ? 1.0f : 0.0f;
        *(float2*)(tb + (size_t)g * 2) = make_float2(h.x * inv_t, h.y * inv_t);
    }
}

int launch_pack_ml(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue,
                   const uint8_t* mask, int32_t normalize, float* inputs, float* targets) {
    if (B <= 0) return CHEST_OK;
    const auto& f = ctx->fp;
    hipLaunchKernelGGL(pack_ml_kernel, dim3(B), dim3(256), 0, st, f.S, f.R, f.T, f.nsc, (const float2*)rx, (const float2*)Hls,
                       (const float2*)Htrue, mask, normalize, inputs, targets);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// ---- training-side packing ------------------------------------------------------------------------------------------
// Dataset-level statistics of the first-antenna-pair slices rx[:, :, 0, :], H_ls[:, :, 0, 0, :], H_true[:, :, 0, 0, :]:
// per tensor sum re, sum im, sum re^2, sum im^2, sum |x| in float64; deterministic two-stage reduction (per-workgroup partials in
// a fixed grid, then one workgroup in a fixed order).
constexpr int ML_STAT_BLOCKS = 1024, ML_NSUM = 15;

__global__ void __launch_bounds__(256)
ml_stats_partial_kernel(int64_t rows, int S, int R, int T, int nsc, const float2* __restrict__ rx, const float2* __restrict__ Hls,
                        const float2* __restrict__ Htrue, double* __restrict__ partial) {
    __shared__ double red[4][ML_NSUM];
    double acc[ML_NSUM];
#pragma unroll
    for (int i = 0; i < ML_NSUM; ++i) acc[i] = 0.0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {          // row = b * S + s
        const float2* a = rx + (size_t)row * R * nsc;
        const float2* l = Hls + (size_t)row * R * T * nsc;
        const float2* h = Htrue + (size_t)row * R * T * nsc;
        for (int k = threadIdx.x; k < nsc; k += blockDim.x) {
            const float2 v[3] = {a[k], l[k], h[k]};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const double re = v[t].x, im = v[t].y;
                acc[5 * t + 0] += re; acc[5 * t + 1] += im; acc[5 * t + 2] += re * re; acc[5 * t + 3] += im * im;
                acc[5 * t + 4] += sqrt(re * re + im * im);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < ML_NSUM; ++i) {
        const double w = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = w;
    }
    __syncthreads();
    if (threadIdx.x < ML_NSUM)
        partial[(size_t)blockIdx.x * ML_NSUM + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stats[9] = rx_mean, rx_std, H_ls_mean, H_ls_std, H_true_mean, H_true_std   (src/train.py:41-58: the mean of the real-part and
//            imaginary-part statistics, population std)
//            rx_std, H_ls_std, H_true_std of run_phase4_training.py:73-76 (np.std(np.abs(x)) + 1e-8)
__global__ void __launch_bounds__(64)
ml_stats_final_kernel(int nblocks, double count, const double* __restrict__ partial, double* __restrict__ stats) {
    __shared__ double tot[ML_NSUM];
    if (threadIdx.x < ML_NSUM) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * ML_NSUM + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double* q = tot + 5 * threadIdx.x;
        const double mre = q[0] / count, mim = q[1] / count;
        const double vre = q[2] / count - mre * mre, vim = q[3] / count - mim * mim;
        stats[2 * threadIdx.x + 0] = 0.5 * (mre + mim);
        stats[2 * threadIdx.x + 1] = 0.5 * (sqrt(vre > 0 ? vre : 0.0) + sqrt(vim > 0 ? vim : 0.0));
        const double mab = q[4] / count, vab = (q[2] + q[3]) / count - mab * mab;
        stats[6 + threadIdx.x] = sqrt(vab > 0 ? vab : 0.0) + 1e-8;
    }
}

// inputs [B][5][S][nsc] = rx re, rx im, H_ls re, H_ls im, mask; targets [B][2][S][nsc] = H_true re, im; mask_out [B][S][nsc] f32.
// mode 0: raw; 1: (x - mean) / (std + 1e-8) per tensor (train.py:77-81, float32 arithmetic like the reference's);
// 2: x / std (run_phase4_training.py:92-96).  One workgroup per (frame, symbol) row, lanes along the subcarriers.
__global__ void __launch_bounds__(256)
pack_train_kernel(int S, int R, int T, int nsc, const float2* __restrict__ rx, const float2* __restrict__ Hls,
                  const float2* __restrict__ Htrue, const void* __restrict__ mask, int mask_is_f32, int mode,
                  const double* __restrict__ stats, float* __restrict__ inputs, float* __restrict__ targets,
                  float* __restrict__ mask_out) {
    const int64_t row = blockIdx.x;
    const int64_t b = row / S;
    const int s = (int)(row - b * S);
    float mean[3] = {0.f, 0.f, 0.f}, den[3] = {1.f, 1.f, 1.f};
    if (mode == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { mean[t] = (float)stats[2 * t]; den[t] = (float)stats[2 * t + 1] + 1e-8f; }
    } else if (mode == 2) {
#pragma unroll
        for (int t = 0; t < 3; ++t) den[t] = (float)stats[6 + t];
    }
    const float2* a = rx + (size_t)row * R * nsc;
    const float2* l = Hls + (size_t)row * R * T * nsc;
    const float2* h = Htrue + (size_t)row * R * T * nsc;
    const size_t plane = (size_t)S * nsc;
    float* ib = inputs + (size_t)b * 5 * plane + (size_t)s * nsc;
    float* tb = targets + (size_t)b * 2 * plane + (size_t)s * nsc;
    for (int k = threadIdx.x; k < nsc; k += blockDim.x) {
        const float2 va = a[k], vl = l[k], vh = h[k];
        const float m = mask_is_f32 ? ((const float*)mask)[(size_t)row * nsc + k]
                                    : (((const uint8_t*)mask)[(size_t)row * nsc + k] ? 1.0f : 0.0f);
        ib[k] = (va.x - mean[0]) / den[0];
        ib[plane + k] = (va.y - mean[0]) / den[0];
        ib[2 * plane + k] = (vl.x - mean[1]) / den[1];
        ib[3 * plane + k] = (vl.y - mean[1]) / den[1];
        ib[4 * plane + k] = m;
        tb[k] = (vh.x - mean[2]) / den[2];
        tb[plane + k] = (vh.y - mean[2]) / den[2];
        if (mask_out) mask_out[(size_t)row * nsc + k] = m;
    }
}

int64_t ml_stats_workspace_bytes() { return (int64_t)ML_STAT_BLOCKS * ML_NSUM * sizeof(double); }

int launch_ml_stats(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue, double* stats,
                    void* workspace, int64_t workspace_bytes) {
    if (workspace_bytes < ml_stats_workspace_bytes()) { set_error("chest_ml_dataset_stats: workspace holds %lld B, %lld needed", (long long)workspace_bytes, (long long)ml_stats_workspace_bytes()); return CHEST_EINVAL; }
    if (B <= 0) { set_error("chest_ml_dataset_stats: statistics of an empty dataset"); return CHEST_EINVAL; }
    const auto& f = ctx->fp;
    const int64_t rows = (int64_t)B * f.S;
    const int nb = (int)(rows < ML_STAT_BLOCKS ? rows : ML_STAT_BLOCKS);
    hipLaunchKernelGGL(ml_stats_partial_kernel, dim3(nb), dim3(256), 0, st, rows, f.S, f.R, f.T, f.nsc, (const float2*)rx,
                       (const float2*)Hls, (const float2*)Htrue, (double*)workspace);
    hipLaunchKernelGGL(ml_stats_final_kernel, dim3(1), dim3(64), 0, st, nb, (double)rows * f.nsc, (const double*)workspace, stats);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int launch_pack_train(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue,
                      const void* mask, int32_t mask_is_f32, int32_t mode, const double* stats, float* inputs, float* targets,
                      float* mask_out) {
    if (B <= 0) return CHEST_OK;
    const auto& f = ctx->fp;
    hipLaunchKernelGGL(pack_train_kernel, dim3((unsigned)((int64_t)B * f.S)), dim3(256), 0, st, f.S, f.R, f.T, f.nsc,
                       (const float2*)rx, (const float2*)Hls, (const float2*)Htrue, mask, mask_is_f32, mode, stats, inputs, targets,
                       mask_out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
