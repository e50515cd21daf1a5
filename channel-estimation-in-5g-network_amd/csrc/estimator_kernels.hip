// Stand-alone estimator kernels for caller-supplied grids (the drop-in LSEstimator / MMSEEstimator
// surface, src/baseline_estimators.py:10-270) and the evaluate_estimator partial sums (:315-337).
// The frame pipeline (frame_kernel.hip) carries its own fused copy of this arithmetic.
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

// One workgroup per (frame, rx, tx) grid: gather pilots -> LDS, optional diagonal-statistics MMSE
// shrinkage, then the plan-driven 3-point interpolation over all S*nsc resource elements.
__global__ void __launch_bounds__(256)
estimate_kernel(int S, int R, int T, int nsc, const PatternDev* __restrict__ patterns, int slot,
                const float2* __restrict__ rx4d, const float2* __restrict__ tx_pilots, int pilot_stride,
                int mmse, float snr_db, float2* __restrict__ H_est) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const PatternDev pat = patterns[slot];
    const int P = pat.P, G = S * nsc, tid = threadIdx.x, nthr = blockDim.x;
    float2* hp = (float2*)lds_raw;
    float* red = (float*)(hp + P);
    const int grid_id = blockIdx.x;                  // ((b*R + r)*T + t)
    const int t = grid_id % T;
    const int r = (grid_id / T) % R;
    const int b = grid_id / (T * R);
    const size_t frame_base = (size_t)b * S * R * T * nsc;

    float acc = 0.f;
    for (int i = tid; i < P; i += nthr) {
        const int g = pat.pilot_idx[i];
        const int s = g / nsc, k = g - s * nsc;
        const float2 y = rx4d[frame_base + ((size_t)(s * R + r) * T + t) * nsc + k];
        const float2 x = tx_pilots[(size_t)b * pilot_stride + i];
        const float2 h = pilot_divide(y, x);
        hp[i] = h;
        acc += cabs2(h);
    }
    __syncthreads();
    if (mmse) {
        // sigma^2 = 10^(-snr/10); R_h = mean|h_ls|^2 I  =>  h *= p/(p+sigma^2)   (:174-190)
        const float tot = block_sum(acc, red);
        const double p = (double)tot / (double)P;
        const float alpha = (float)(p / (p + pow(10.0, -(double)snr_db / 10.0)));
        for (int i = tid; i < P; i += nthr) { float2 h = hp[i]; hp[i] = make_float2(alpha * h.x, alpha * h.y); }
        __syncthreads();
    }
    const gptr_plan plan = (gptr_plan)pat.plan;
    for (int g = tid; g < G; g += nthr) {
        const float2 v = interp(ld_plan(plan, g), hp, 1, 0);
        const int s = g / nsc, k = g - s * nsc;
        H_est[frame_base + ((size_t)(s * R + r) * T + t) * nsc + k] = v;
    }
}

int launch_estimate(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx4d, const float* tx_pilots,
                    int32_t pilot_stride, int32_t slot, int32_t mmse, float snr_db, float* H_est) {
    if (B <= 0) return CHEST_OK;
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    const PatternDev& pd = ctx->patterns[slot].dev;
    if (pd.plan == nullptr) { set_error("pattern slot %d has no interpolation plan", slot); return CHEST_ESTATE; }
    if (pd.pad != 0) { set_error("pattern slot %d carries a compact plan (chest_patterns_build method 2): only the fused frame pipeline reads those", slot); return CHEST_ESTATE; }
    const size_t lds = (size_t)pd.P * sizeof(float2) + 32 * sizeof(float);
    if (lds > 160 * 1024) { set_error("too many pilots for LDS"); return CHEST_EINVAL; }
    CHEST_LDS_OPT_IN(ctx, LDS_ESTIMATE, estimate_kernel, lds);
    const auto& f = ctx->fp;
    hipLaunchKernelGGL(estimate_kernel, dim3((unsigned)(B * f.R * f.T)), dim3(256), lds, st, f.S, f.R, f.T, f.nsc,
                       ctx->d_patterns, slot, (const float2*)rx4d, (const float2*)tx_pilots, pilot_stride, mmse,
                       snr_db, (float2*)H_est);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// estimate_at_pilots on a flat vector (single workgroup: n is a pilot count, a few thousand at most).
__global__ void __launch_bounds__(1024)
at_pilots_kernel(int n, const float2* __restrict__ rx, const float2* __restrict__ tx, int mmse, float snr_db,
                 float2* __restrict__ out) {
    __shared__ float red[32];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float2 h = pilot_divide(rx[i], tx[i]);
        out[i] = h;
        acc += cabs2(h);
    }
    if (!mmse) return;
    const float tot = block_sum(acc, red);
    const double p = (double)tot / (double)n;
    const float alpha = (float)(p / (p + pow(10.0, -(double)snr_db / 10.0)));
    for (int i = threadIdx.x; i < n; i += blockDim.x) {       // each thread rescales what it wrote
        float2 h = out[i];
        out[i] = make_float2(alpha * h.x, alpha * h.y);
    }
}

int launch_at_pilots(chest_ctx*, hipStream_t st, int32_t n, const float* rx, const float* tx, int32_t mmse,
                     float snr_db, float* out) {
    if (n <= 0) return CHEST_OK;
    hipLaunchKernelGGL(at_pilots_kernel, dim3(1), dim3(1024), 0, st, n, (const float2*)rx, (const float2*)tx, mmse,
                       snr_db, (float2*)out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// sum|a-b|^2 and sum|a|^2 with float64 accumulation (evaluate_estimator, :326-330).
__global__ void __launch_bounds__(256)
error_sums_kernel(int64_t n, const float2* __restrict__ a, const float2* __restrict__ b, double* __restrict__ out) {
    __shared__ double red[2][4];
    double e = 0.0, p = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 x = a[i], y = b[i];
        const double dx = (double)x.x - (double)y.x, dy = (double)x.y - (double)y.y;
        e += dx * dx + dy * dy;
        p += (double)x.x * x.x + (double)x.y * x.y;
    }
    e = wave_sum(e); p = wave_sum(p);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = e; red[1][wave] = p; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(out + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

int launch_error_sums(chest_ctx*, hipStream_t st, int64_t n, const float* a, const float* b, double* out) {
    CHEST_HIP(hipMemsetAsync(out, 0, 2 * sizeof(double), st));
    if (n <= 0) return CHEST_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(error_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, (const float2*)a,
                       (const float2*)b, out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
