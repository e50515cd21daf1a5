// Full-covariance MMSE: MMSEEstimator(channel_covariance=R, estimate_statistics=False)
// (reference src/baseline_estimators.py:181-190):   h_mmse = R (R + sigma^2 I)^-1 h_ls.
//
// prepare: W = R (R + s2 I)^-1 = I - s2 (R + s2 I)^-1.  The system is Hermitian positive definite and
//          fp32-ill-conditioned at high SNR (cond ~ 3.5e5), so the inverse is formed in float64
//          on the device by a BLOCKED in-place Gauss-Jordan sweep (64-wide pivot blocks, no pivoting needed for HPD),
//          then rounded once to c64.  It runs once per (pilot pattern, sigma^2).
// apply  : gather h_ls = y/(x+1e-12) for all B*R*T grids, complex GEMM W[P,P] x h_ls[P,M] on the exact-f32
//          MFMA pipe (v_mfma_f32_32x32x2_f32; 4 real MFMAs per complex k-pair), then the usual plan-driven
//          interpolation.  This is the only MFMA-shaped work on the hot path.
#include "chest_device.h"
#include "chest_internal.h"
#include "gj_blocked.h"

namespace chest {


// ---- prepare: blocked in-place Gauss-Jordan inversion in float64 (gj_blocked.h) ------------------------------------------
__global__ void __launch_bounds__(256) gjb_init_kernel(int P, int Ppad, const cdd* __restrict__ cov, double s2, cdd* __restrict__ A) {
    const size_t n = (size_t)Ppad * Ppad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / Ppad), j = (int)(e % Ppad);
        cdd v{0.0, 0.0};
        if (i < P && j < P) { v = cov[(size_t)i * P + j]; if (i == j) v.x += s2; }
        else if (i == j) v.x = 1.0;
        A[e] = v;
    }
}

__global__ void __launch_bounds__(256) gjb_finish_kernel(int P, int Ppad, const cdd* __restrict__ Ainv, double s2, float2* __restrict__ W) {
    const size_t n = (size_t)P * P;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / P), j = (int)(e % P);
        const cdd a = Ainv[(size_t)i * Ppad + j];
        W[e] = make_float2((float)((i == j ? 1.0 : 0.0) - s2 * a.x), (float)(-s2 * a.y));
    }
}

// ---- apply --------------------------------------------------------------------------------------
// h_ls for every (frame, rx, tx) grid, stored [M][Ppad] (pilot index contiguous = GEMM K dimension).
__global__ void __launch_bounds__(256)
gather_ls_kernel(int S, int R, int T, int nsc, int P, int Ppad, int M, const int32_t* __restrict__ pilot_idx,
                 const float2* __restrict__ rx4d, const float2* __restrict__ tx_pilots, int pilot_stride,
                 float2* __restrict__ h) {
    const int m = blockIdx.x;                       // (b*R + r)*T + t; blocks beyond the last grid zero the tile padding
    const int t = m % T, r = (m / T) % R, b = m / (T * R);
    const size_t frame_base = (size_t)b * S * R * T * nsc;
    for (int i = threadIdx.x; i < Ppad; i += blockDim.x) {
        float2 v = make_float2(0.f, 0.f);
        if (i < P && m < M) {
            const int g = pilot_idx[i];
            const int s = g / nsc, k = g - s * nsc;
            const float2 y = rx4d[frame_base + ((size_t)(s * R + r) * T + t) * nsc + k];
            const float2 x = tx_pilots[(size_t)b * pilot_stride + i];
            const float2 xe = make_float2(x.x + 1e-12f, x.y);
            const float inv = 1.0f / cabs2(xe);
            v = make_float2((y.x * xe.x + y.y * xe.y) * inv, (y.y * xe.x - y.x * xe.y) * inv);
        }
        h[(size_t)m * Ppad + i] = v;
    }
}

using f32x16 = __attribute__((ext_vector_type(16))) float;
// 128 x 128 block tile, 4 waves x (2 x 2) 32x32 MFMA tiles (16 MFMAs per four ds_read_b64).  Rows of the MFMA tile = grids m
// (registers), columns = pilot index i (lanes), so every C/D register is stored as a 256-B run of Hout[m][i0..i0+31].
// Both operands come from PACKED scratch copies -- rows padded to the tile, K padded to 16, zero filled -- so every thread
// fetches its 8 k-values of a row with four unconditional 16-B loads; the LDS tiles ([row][k], one pad element: conflict-
// free ds_read_b64) are double buffered: the registers of tile t+1 are written to the other buffer right after the MFMA
// block of tile t, one barrier per k-tile.
constexpr int T2_M = 128, T2_N = 128, T2_K = 16;
constexpr int T3_LD = T2_K + 1;
constexpr size_t T3_LDS_BYTES = (size_t)2 * 2 * T2_M * T3_LD * sizeof(float2);      // 2 buffers x (A, B) = 69 632 B

// Three-multiplication complex product ("3M"): (ar + j ai)(br + j bi) = (ar br - ai bi) + j((ar + ai)(br + bi) - ar br - ai bi),
// i.e. three real MFMAs per complex k-pair instead of four (-25 % matrix work) for three accumulator sets instead of two
// (192 AGPRs).  Rounding: measured 1.05e-6 of max|W h| against float64 at P = 1 678 (four-multiplication form: 7.8e-7).
__global__ void __launch_bounds__(256, 2)
cgemm_w_kernel_3m(int P, int Kpad, int M, int ldo, const float2* __restrict__ Wp, const float2* __restrict__ Hin,
                  float2* __restrict__ Hout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2* const lds = (float2*)lds_raw;          // [buf][A | B][128][T3_LD]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i0 = blockIdx.x * T2_N, m0 = blockIdx.y * T2_M;
    const int wm = (wave & 1) * 64, wi = (wave >> 1) * 64;
    f32x16 t1[2][2], t2[2][2], t3[2][2];          // sum ar*br, sum ai*bi, sum (ar+ai)*(br+bi)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 16; ++e) { t1[x][y][e] = 0.f; t2[x][y][e] = 0.f; t3[x][y][e] = 0.f; }
    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const float4* gh = (const float4*)(Hin + (size_t)(m0 + lrow) * Kpad + lk);      // 16-B aligned: Kpad % 16 == 0
    const float4* gw = (const float4*)(Wp + (size_t)(i0 + lrow) * Kpad + lk);
    float4 ph[4], pw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ph[q] = gh[(k0 >> 1) + q]; pw[q] = gw[(k0 >> 1) + q]; }
    };
    auto stage = [&](int buf) {
        float2* Hs = lds + (size_t)buf * 2 * T2_M * T3_LD;
        float2* Ws = Hs + T2_M * T3_LD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Hs[lrow * T3_LD + lk + 2 * q] = make_float2(ph[q].x, ph[q].y);
            Hs[lrow * T3_LD + lk + 2 * q + 1] = make_float2(ph[q].z, ph[q].w);
            Ws[lrow * T3_LD + lk + 2 * q] = make_float2(pw[q].x, pw[q].y);
            Ws[lrow * T3_LD + lk + 2 * q + 1] = make_float2(pw[q].z, pw[q].w);
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int nk = Kpad / T2_K;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) fetch((t + 1) * T2_K);
        const float2* Hs = lds + (size_t)(t & 1) * 2 * T2_M * T3_LD;
        const float2* Ws = Hs + T2_M * T3_LD;
#pragma unroll
        for (int kk = 0; kk < T2_K; kk += 2) {
            float2 a[2], b[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) a[x] = Hs[(wm + 32 * x + (lane & 31)) * T3_LD + kk + (lane >> 5)];
#pragma unroll
            for (int y = 0; y < 2; ++y) b[y] = Ws[(wi + 32 * y + (lane & 31)) * T3_LD + kk + (lane >> 5)];
            float as[2], bs[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) as[x] = a[x].x + a[x].y;
#pragma unroll
            for (int y = 0; y < 2; ++y) bs[y] = b[y].x + b[y].y;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    t1[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].x, b[y].x, t1[x][y], 0, 0, 0);
                    t2[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].y, b[y].y, t2[x][y], 0, 0, 0);
                    t3[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[x], bs[y], t3[x][y], 0, 0, 0);
                }
        }
        if (t + 1 < nk) stage((t + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int i = i0 + wi + 32 * y + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < M && i < P) {
                    const float p1 = t1[x][y][reg], p2 = t2[x][y][reg];
                    Hout[(size_t)m * ldo + i] = make_float2(p1 - p2, t3[x][y][reg] - p1 - p2);
                }
            }
        }
}

// W [P][P] -> Wp [Npad][Kpad], zero padded (Npad = P rounded up to the 128-row block tile, Kpad = P rounded up to 16)
__global__ void __launch_bounds__(256)
pack_w_kernel(int P, int Kpad, const float2* __restrict__ W, float2* __restrict__ Wp) {
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x)
        Wp[(size_t)i * Kpad + k] = (i < P && k < P) ? W[(size_t)i * P + k] : make_float2(0.f, 0.f);
}

// plan-driven interpolation of precomputed pilot values h[m][Ppad] (same arithmetic as estimate_kernel)
// tfold == 0: one block per grid m = (b*R + r)*T + t.  tfold != 0: one block per (b, r); its result is written to all T
// grids of that rx (the reference's callers repeat rx over tx, quirk Q9, so those grids are identical).
__global__ void __launch_bounds__(256)
interp_from_pilots_kernel(int S, int R, int T, int nsc, int Ppad, const PatternDev* __restrict__ patterns, int slot,
                          const float2* __restrict__ h, float2* __restrict__ H_est, int tfold) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2* hp = (float2*)lds_raw;
    const PatternDev pat = patterns[slot];
    const int P = pat.P, G = S * nsc, m = blockIdx.x;
    const int t0 = tfold ? 0 : m % T, nt = tfold ? T : 1;
    const int r = tfold ? m % R : (m / T) % R, b = tfold ? m / R : m / (T * R);
    for (int i = threadIdx.x; i < P; i += blockDim.x) hp[i] = h[(size_t)m * Ppad + i];
    __syncthreads();
    const gptr_plan plan = (gptr_plan)pat.plan;
    const size_t frame_base = (size_t)b * S * R * T * nsc;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float2 v = interp(ld_plan(plan, g), hp, 1, 0);
        const int s = g / nsc, k = g - s * nsc;
        for (int t = t0; t < t0 + nt; ++t) H_est[frame_base + ((size_t)(s * R + r) * T + t) * nsc + k] = v;
    }
}

// The tx-replicated form of the pipeline route (one workgroup per (frame, rx)) with the frame kernel's write path: a symbol's
// nsc interpolated values are staged in LDS once, and the T identical rows H_est[b][s][r][0..T) -- ONE contiguous span of
// T * nsc elements -- leave as consecutive 128-B-line-aligned 1 KiB wave stores that read the LDS row modulo nsc.
typedef float v4f_i __attribute__((ext_vector_type(4)));
typedef float v2f_i __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256)
interp_from_pilots_tfold_kernel(int S, int R, int T, int nsc, int Ppad, const PatternDev* __restrict__ patterns, int slot,
                                const float2* __restrict__ h, float2* __restrict__ H_est) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const PatternDev pat = patterns[slot];
    const int P = pat.P, m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* hp = (float2*)lds_raw;                                  // [P rounded up to even]
    float2* row = hp + ((P + 1) & ~1);                              // [nsc]
    const int r = m % R, b = m / R;
    for (int i = tid; i < P; i += blockDim.x) hp[i] = h[(size_t)m * Ppad + i];
    const gptr_plan plan = (gptr_plan)pat.plan;
    const int n = T * nsc;
    const float inv_nsc = 1.0f / (float)nsc;
    for (int s = 0; s < S; ++s) {
        __syncthreads();                                            // hp ready / previous symbol's row fully read
        for (int k = tid; k < nsc; k += blockDim.x) row[k] = interp(ld_plan(plan, s * nsc + k), hp, 1, 0);
        __syncthreads();
        float2* __restrict__ g0 = H_est + ((size_t)(b * S + s) * R + r) * T * nsc;
        const int head = (int)(((uintptr_t)g0 & 127u) >> 3);
        const int nst = (head + n + 127) >> 7;
        for (int q = wave; q < nst; q += 4) {
            const int e = (q << 7) - head + 2 * lane;
            if (e >= -1 && e < n) {
                const int ec = e < 0 ? 0 : e;
                int i0 = ec - (int)((float)ec * inv_nsc) * nsc;         // ec mod nsc (float quotient may be off by one)
                if (i0 < 0) i0 += nsc;
                if (i0 >= nsc) i0 -= nsc;
                if (e >= 0 && e + 1 < n) {
                    const int i1 = i0 + 1 == nsc ? 0 : i0 + 1;
                    const float2 a = row[i0], c = row[i1];
                    const v4f_i v = {a.x, a.y, c.x, c.y};
                    __builtin_nontemporal_store(v, (v4f_i*)(g0 + e));
                } else {                                             // e == -1: element 0 alone; e == n - 1: last element alone
                    const float2 a = e < 0 ? row[0] : row[i0];
                    const v2f_i v = {a.x, a.y};
                    __builtin_nontemporal_store(v, (v2f_i*)(g0 + ec));
                }
            }
        }
    }
}

// h_ls_pilots [M][stride] (as chest_frames_run writes it) -> GEMM A operand [Mpad][Ppad], zero padded
__global__ void __launch_bounds__(256)
pack_h_kernel(int M, int P, int stride, int Ppad, const float2* __restrict__ h, float2* __restrict__ hp) {
    const int m = blockIdx.x;
    for (int i = threadIdx.x; i < Ppad; i += blockDim.x) hp[(size_t)m * Ppad + i] = (m < M && i < P) ? h[(size_t)m * stride + i] : make_float2(0.f, 0.f);
}

// Dense interpolation operator (Clough-Tocher 'cubic'): H[m][g] = sum_p A[g][p] h[m][p], A real [G][P], h complex [M][Ppad].
// Rows of the MFMA tile = grids m, columns = resource elements g, so that the 32 lanes of a C/D register hold consecutive
// subcarriers of one grid row (coalesced 256-B stores into [B][S][R][T][nsc]).  Built like cgemm_w_kernel_3m: 128 (grids m)
// x 128 (resource elements g) block tile, 2 x 2 MFMA tiles per wave, operands from tile-padded scratch copies through 16-B
// loads, double-buffered LDS, one barrier per k-tile.  A is real: two MFMAs per k-pair and sub-tile.
constexpr int D2_LDH = T2_K + 1, D2_LDA = T2_K + 1;
constexpr size_t D2_LDS_BYTES = (size_t)2 * (T2_M * D2_LDH * sizeof(float2) + T2_N * D2_LDA * sizeof(float));   // 52 224 B

__global__ void __launch_bounds__(256, 2)
dense_interp_kernel(int S, int R, int T, int nsc, int Kpad, int M, const float* __restrict__ Ap,
                       const float2* __restrict__ Hin, float2* __restrict__ H_est) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr size_t BUF = T2_M * D2_LDH * sizeof(float2) + T2_N * D2_LDA * sizeof(float);
    const int G = S * nsc;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g0 = blockIdx.x * T2_N, m0 = blockIdx.y * T2_M;
    const int wm = (wave & 1) * 64, wg = (wave >> 1) * 64;
    f32x16 cre[2][2], cim[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int e = 0; e < 16; ++e) { cre[x][y][e] = 0.f; cim[x][y][e] = 0.f; }
    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const float4* gh = (const float4*)(Hin + (size_t)(m0 + lrow) * Kpad + lk);      // 8 float2 = 4 x 16 B
    const float4* ga = (const float4*)(Ap + (size_t)(g0 + lrow) * Kpad + lk);       // 8 float  = 2 x 16 B
    float4 ph[4], pa[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ph[q] = gh[(k0 >> 1) + q];
#pragma unroll
        for (int q = 0; q < 2; ++q) pa[q] = ga[(k0 >> 2) + q];
    };
    auto stage = [&](int buf) {
        float2* Hs = (float2*)(lds_raw + (size_t)buf * BUF);
        float* As = (float*)(Hs + T2_M * D2_LDH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Hs[lrow * D2_LDH + lk + 2 * q] = make_float2(ph[q].x, ph[q].y);
            Hs[lrow * D2_LDH + lk + 2 * q + 1] = make_float2(ph[q].z, ph[q].w);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            As[lrow * D2_LDA + lk + 4 * q] = pa[q].x; As[lrow * D2_LDA + lk + 4 * q + 1] = pa[q].y;
            As[lrow * D2_LDA + lk + 4 * q + 2] = pa[q].z; As[lrow * D2_LDA + lk + 4 * q + 3] = pa[q].w;
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int nk = Kpad / T2_K;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) fetch((t + 1) * T2_K);
        const float2* Hs = (const float2*)(lds_raw + (size_t)(t & 1) * BUF);
        const float* As = (const float*)(Hs + T2_M * D2_LDH);
#pragma unroll
        for (int kk = 0; kk < T2_K; kk += 2) {
            float2 a[2];
            float b[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) a[x] = Hs[(wm + 32 * x + (lane & 31)) * D2_LDH + kk + (lane >> 5)];
#pragma unroll
            for (int y = 0; y < 2; ++y) b[y] = As[(wg + 32 * y + (lane & 31)) * D2_LDA + kk + (lane >> 5)];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    cre[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].x, b[y], cre[x][y], 0, 0, 0);
                    cim[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x].y, b[y], cim[x][y], 0, 0, 0);
                }
        }
        if (t + 1 < nk) stage((t + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int g = g0 + wg + 32 * y + (lane & 31);
        if (g >= G) continue;
        const int s = g / nsc, k = g - s * nsc;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < M) {
                    const int t = m % T, r = (m / T) % R, b = m / (T * R);
                    H_est[(((size_t)b * S + s) * R + r) * T * nsc + (size_t)t * nsc + k] = make_float2(cre[x][y][reg], cim[x][y][reg]);
                }
            }
    }
}

// A [G][P] -> Ap [Gpad][Kpad] float, zero padded
__global__ void __launch_bounds__(256)
pack_a_kernel(int G, int P, int Kpad, const float* __restrict__ A, float* __restrict__ Ap) {
    const int g = blockIdx.x;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) Ap[(size_t)g * Kpad + k] = (g < G && k < P) ? A[(size_t)g * P + k] : 0.f;
}

static inline int pad16(int p) { return (p + 15) & ~15; }
static inline int pad128(int p) { return (p + 127) & ~127; }

}  // namespace chest

using namespace chest;

extern "C" {

int64_t chest_mmse_full_workspace_bytes(int32_t P) {
    if (P <= 0) return 0;
    const int64_t Ppad = ((int64_t)P + GJB - 1) / GJB * GJB;
    // A [Ppad][Ppad] | Rp [64][Ppad] | Cp [Ppad / 64][64][64] | Dinv x 2 [64][64] (current + look-ahead), complex float64
    return (int64_t)sizeof(double) * 2 * (Ppad * Ppad + (int64_t)gj_aux_elems((int)Ppad));
}

int chest_mmse_full_prepare(chest_ctx* ctx, void* stream, int32_t P, const double* cov, float snr_db, float* W,
                            double* workspace, int64_t workspace_bytes) {
    if (!ctx || P <= 0 || !cov || !W || !workspace) { set_error("chest_mmse_full_prepare: bad argument"); return CHEST_EINVAL; }
    if (workspace_bytes < chest_mmse_full_workspace_bytes(P)) {
        set_error("chest_mmse_full_prepare: workspace %lld B < %lld B", (long long)workspace_bytes,
                  (long long)chest_mmse_full_workspace_bytes(P));
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const double s2 = 1.0 / pow(10.0, (double)snr_db / 10.0);              // (:174-175)
    const int Ppad = (P + GJB - 1) / GJB * GJB;
    cdd* A = (cdd*)workspace;
    const size_t n = (size_t)Ppad * Ppad;
    const unsigned nb = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(gjb_init_kernel, dim3(nb), dim3(256), 0, st, P, Ppad, (const cdd*)cov, s2, A);
    const int rc = launch_gj_inverse<cdd>(ctx, st, Ppad, A, A + n, LDS_GJ_PANEL, LDS_GJ_UPDATE);
    if (rc != CHEST_OK) return rc;
    hipLaunchKernelGGL(gjb_finish_kernel, dim3(nb), dim3(256), 0, st, P, Ppad, A, s2, (float2*)W);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int64_t chest_mmse_full_apply_scratch_bytes(int32_t P, int32_t num_grids) {
    if (P <= 0 || num_grids <= 0) return 0;
    const int64_t Ppad = pad16(P);
    return (int64_t)sizeof(float2) * Ppad * ((int64_t)pad128(num_grids) + num_grids + pad128(P));
}

int chest_mmse_full_apply(chest_ctx* ctx, void* stream, int32_t B, const float* rx4d, const float* tx_pilots,
                          int32_t pilot_stride, int32_t slot, const float* W, float* h_scratch, float* H_est) {
    if (!ctx || B < 0 || (B > 0 && (!rx4d || !tx_pilots || !W || !h_scratch || !H_est))) {
        set_error("chest_mmse_full_apply: bad argument");
        return CHEST_EINVAL;
    }
    if (B == 0) return CHEST_OK;
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.slot == nullptr ||
        ctx->patterns[slot].dev.plan == nullptr || ctx->patterns[slot].dev.pad != 0) {
        set_error("pattern slot %d is not set, has no plan or carries a compact plan", slot);
        return CHEST_ESTATE;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R * f.T;
    // scratch layout (chest_mmse_full_apply_scratch_bytes): h_in [Mpad][Ppad] | h_out [M][Ppad] | Wp [Npad][Ppad]
    const int Mpad = pad128(M), Npad = pad128(P);
    float2* h_in = (float2*)h_scratch;
    float2* h_out = h_in + (size_t)Mpad * Ppad;
    float2* w_packed = h_out + (size_t)M * Ppad;
    hipLaunchKernelGGL(gather_ls_kernel, dim3(Mpad), dim3(256), 0, st, f.S, f.R, f.T, f.nsc, P, Ppad, M, pd.pilot_idx,
                       (const float2*)rx4d, (const float2*)tx_pilots, pilot_stride, h_in);
    hipLaunchKernelGGL(pack_w_kernel, dim3(Npad), dim3(256), 0, st, P, Ppad, (const float2*)W, w_packed);
    CHEST_LDS_OPT_IN(ctx, LDS_GEMM3M, cgemm_w_kernel_3m, T3_LDS_BYTES);
    hipLaunchKernelGGL(cgemm_w_kernel_3m, dim3(Npad / T2_N, Mpad / T2_M), dim3(256), T3_LDS_BYTES, st, P, Ppad, M, Ppad,
                       (const float2*)w_packed, h_in, h_out);
    const size_t lds = (size_t)P * sizeof(float2);
    CHEST_LDS_OPT_IN(ctx, LDS_INTERP, interp_from_pilots_kernel, lds);
    hipLaunchKernelGGL(interp_from_pilots_kernel, dim3(M), dim3(256), lds, st, f.S, f.R, f.T, f.nsc, Ppad, ctx->d_patterns,
                       slot, h_out, (float2*)H_est, 0);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

// The two stages of chest_mmse_full_apply_pilots, separately launchable (a pipeline runs the filter of batch n + 1 on the MFMA
// pipe beside the HBM-bound interpolation of batch n on another stream).  Scratch layout (chest_mmse_full_apply_scratch_bytes):
// packed right-hand sides [Mpad][Ppad] | filtered pilot values [M][Ppad] | packed W [Npad][Ppad].
static int full_pilots_check(chest_ctx* ctx, const char* who, int32_t B, int32_t slot, const void* scratch) {
    if (!ctx || B < 0 || (B > 0 && !scratch)) { set_error("%s: bad argument", who); return CHEST_EINVAL; }
    if (B == 0) return CHEST_OK;
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.slot == nullptr ||
        ctx->patterns[slot].dev.plan == nullptr || ctx->patterns[slot].dev.pad != 0) {
        set_error("pattern slot %d is not set, has no plan or carries a compact plan", slot);
        return CHEST_ESTATE;
    }
    return CHEST_OK;
}

int chest_mmse_full_filter_pilots(chest_ctx* ctx, void* stream, int32_t B, const float* h_ls_pilots, int32_t h_ls_stride,
                                  int32_t slot, const float* W, float* h_scratch) {
    int rc = full_pilots_check(ctx, "chest_mmse_full_filter_pilots", B, slot, h_scratch);
    if (rc != CHEST_OK || B == 0) return rc;
    if (!h_ls_pilots || !W) { set_error("chest_mmse_full_filter_pilots: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R;                     // one right-hand side per (frame, rx)
    if (h_ls_stride < P) { set_error("h_ls_stride %d < %d pilots", h_ls_stride, P); return CHEST_EINVAL; }
    const int Mpad = pad128(M), Npad = pad128(P);
    float2* h_in = (float2*)h_scratch;
    float2* h_out = h_in + (size_t)Mpad * Ppad;
    float2* w_packed = h_out + (size_t)M * Ppad;
    // chest_frames_run writes h_ls_pilots with the entries P .. h_ls_stride - 1 zeroed: with the row stride of the GEMM operand and
    // a whole number of 128-row tiles it IS the packed operand -- no copy
    const bool h_in_place = h_ls_stride == Ppad && M == Mpad && ((uintptr_t)h_ls_pilots & 15) == 0;
    const float2* h_op = h_in_place ? (const float2*)h_ls_pilots : h_in;
    const int sp_pack = prof_open(ctx, st, CHEST_K_PACK);
    if (!h_in_place) hipLaunchKernelGGL(pack_h_kernel, dim3(Mpad), dim3(256), 0, st, M, P, h_ls_stride, Ppad, (const float2*)h_ls_pilots, h_in);
    hipLaunchKernelGGL(pack_w_kernel, dim3(Npad), dim3(256), 0, st, P, Ppad, (const float2*)W, w_packed);
    prof_close(ctx, st, sp_pack);
    CHEST_LDS_OPT_IN(ctx, LDS_GEMM3M, cgemm_w_kernel_3m, T3_LDS_BYTES);
    const int sp_gemm = prof_open(ctx, st, CHEST_K_GEMM);
    hipLaunchKernelGGL(cgemm_w_kernel_3m, dim3(Npad / T2_N, Mpad / T2_M), dim3(256), T3_LDS_BYTES, st, P, Ppad, M, Ppad,
                       (const float2*)w_packed, h_op, h_out);
    prof_close(ctx, st, sp_gemm);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int chest_mmse_full_interp_pilots(chest_ctx* ctx, void* stream, int32_t B, int32_t slot, const float* h_scratch, float* H_est) {
    int rc = full_pilots_check(ctx, "chest_mmse_full_interp_pilots", B, slot, h_scratch);
    if (rc != CHEST_OK || B == 0) return rc;
    if (!H_est) { set_error("chest_mmse_full_interp_pilots: bad argument"); return CHEST_EINVAL; }
    if (((uintptr_t)H_est & 15) != 0) { set_error("chest_mmse_full_interp_pilots: H_est must be 16-byte aligned"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R;
    const float2* h_out = (const float2*)h_scratch + (size_t)pad128(M) * Ppad;
    const size_t lds = ((size_t)((P + 1) & ~1) + (size_t)f.nsc) * sizeof(float2);
    CHEST_LDS_OPT_IN(ctx, LDS_INTERP_TFOLD, interp_from_pilots_tfold_kernel, lds);
    const int sp_int = prof_open(ctx, st, CHEST_K_INTERP);
    hipLaunchKernelGGL(interp_from_pilots_tfold_kernel, dim3(M), dim3(256), lds, st, f.S, f.R, f.T, f.nsc, Ppad, ctx->d_patterns,
                       slot, h_out, (float2*)H_est);
    prof_close(ctx, st, sp_int);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int chest_mmse_full_apply_pilots(chest_ctx* ctx, void* stream, int32_t B, const float* h_ls_pilots, int32_t h_ls_stride,
                                 int32_t slot, const float* W, float* h_scratch, float* H_est) {
    if (B > 0 && (!H_est || ((uintptr_t)H_est & 15) != 0)) { set_error("chest_mmse_full_apply_pilots: H_est must be a 16-byte aligned device pointer"); return CHEST_EINVAL; }
    const int rc = chest_mmse_full_filter_pilots(ctx, stream, B, h_ls_pilots, h_ls_stride, slot, W, h_scratch);
    return rc != CHEST_OK ? rc : chest_mmse_full_interp_pilots(ctx, stream, B, slot, h_scratch, H_est);
}

int64_t chest_estimate_dense_scratch_bytes(chest_ctx* ctx, int32_t P, int32_t num_grids) {
    if (!ctx || P <= 0 || num_grids <= 0) return 0;
    const int64_t Ppad = pad16(P);
    return Ppad * ((int64_t)sizeof(float2) * pad128(num_grids) + (int64_t)sizeof(float) * pad128(ctx->fp.G));
}

int chest_estimate_dense(chest_ctx* ctx, void* stream, int32_t B, const float* rx4d, const float* tx_pilots,
                         int32_t pilot_stride, int32_t slot, const float* A, float* h_scratch, float* H_est) {
    if (!ctx || B < 0 || (B > 0 && (!rx4d || !tx_pilots || !A || !h_scratch || !H_est))) {
        set_error("chest_estimate_dense: bad argument");
        return CHEST_EINVAL;
    }
    if (B == 0) return CHEST_OK;
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.slot == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R * f.T;
    // scratch layout (chest_estimate_dense_scratch_bytes): h [Mpad][Ppad] c64 | Ap [Gpad][Ppad] f32
    const int Mpad = pad128(M), Gpad = pad128(f.G);
    float2* h_in = (float2*)h_scratch;
    float* a_packed = (float*)(h_in + (size_t)Mpad * Ppad);
    hipLaunchKernelGGL(gather_ls_kernel, dim3(Mpad), dim3(256), 0, st, f.S, f.R, f.T, f.nsc, P, Ppad, M, pd.pilot_idx,
                       (const float2*)rx4d, (const float2*)tx_pilots, pilot_stride, h_in);
    hipLaunchKernelGGL(pack_a_kernel, dim3(Gpad), dim3(256), 0, st, f.G, P, Ppad, A, a_packed);
    hipLaunchKernelGGL(dense_interp_kernel, dim3(Gpad / T2_N, Mpad / T2_M), dim3(256), D2_LDS_BYTES, st, f.S, f.R, f.T,
                       f.nsc, Ppad, M, (const float*)a_packed, (const float2*)h_in, (float2*)H_est);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // extern "C"
