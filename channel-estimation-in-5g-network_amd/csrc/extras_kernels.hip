// Kernels on either side of the hot path (SURVEY 8f "next" rows):
//   tdl_kernel      time-domain tapped-delay-line link: y[r][n] = sum_t sum_l h[n][r][t][d_l] x[t][n - d_l].
//                   The reference's pipeline never does this (it multiplies in the frequency domain, quirks Q2/Q4); the
//                   kernel closes the loop OFDMSystem.modulate -> multipath channel (ChannelModel.generate_time_varying_channel,
//                   src/channel_simulator.py:84-127) -> OFDMSystem.demodulate, exposing ISI/ICI when max delay > CP.
//   pack_ml_kernel  prepare_ml_inputs (src/dataset_generator.py:183-227) for a whole batch on the GPU:
//                   inputs [B][S][nsc][5] = (rx re, rx im, H_ls re, H_ls im, mask) of the first rx/tx pair, targets
//                   [B][S][nsc][2] = H_true of that pair, each optionally divided by its population std + 1e-8.
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

}  // namespace chest
