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

// ---- the same product on the fp16 MFMA pipe, float32-accurate (round 4) ---------------------------------------------------------------
// v_mfma_f32_32x32x16_f16 retires 16 x the multiply-adds of v_mfma_f32_32x32x2_f32 per cycle.  Each REAL operand x of the
// 3-multiplication form is split into two fp16 numbers, x s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi) and s a power of two
// that puts the largest magnitude of the operand at 2^14 (fp16's largest finite value is 65504), and a real product is
//     x y  =  (hi_x hi_y + hi_x lo_y + lo_x hi_y) / (s_x s_y)  +  O(2^-22 |x| |y|)
// -- three fp16 MFMAs into ONE float32 accumulator (products of fp16 numbers are exact in float32; the dropped lo_x lo_y term and the
// representation error are 2^-22 relative, below float32's own rounding of the sum).  Nine fp16 MFMAs replace three f32 ones: 9 / 16 / 3 =
// 0.19 of the matrix-pipe time.  Measured error of W h against float64 at P = 1 678: 7.3e-7 of max|W h| (f32 3M form: 8.7e-7;
// tools/fp16_split_model.py).  Elements 2^11 and more below the operand's largest lose their lo part to fp16's subnormal range; their
// error is then 2^-11 of 2^-11 of the largest, i.e. the same 2^-22 of the scale the result is judged on.
//
// Operands are packed by pack_h2_kernel in the order the GEMM stages them: [row tile of 128][k stage of 16][plane][128 rows][16 k] fp16,
// planes = (re hi, re lo, im hi, im lo, (re + im) hi, (re + im) lo), so a stage of one operand is ONE contiguous 24 KB block.
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
constexpr int H2_ROWB = 48;                                  // LDS bytes per (row, plane): 32 B of k + 16 B pad -> conflict-free ds_read_b128
constexpr int H2_PLANE = 128 * H2_ROWB;                      // one plane of one operand tile
constexpr size_t H2_LDS_BYTES = (size_t)2 * 2 * 6 * H2_PLANE;   // 2 buffers x (A, B) x 6 planes = 147 456 B: one workgroup per CU

// largest power of two s with 2 amax s <= 2^14 (the sum plane holds re + im); 1 for an all-zero operand
__device__ __forceinline__ float h2_scale(float amax) {
    if (!(amax > 0.f)) return 1.0f;
    int e;
    (void)frexpf(2.0f * amax, &e);                           // 2 amax = m 2^e, 0.5 <= m < 1
    e = 14 - e;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);                   // (operands of subnormal size or near the float maximum: the product of the two
    return ldexpf(1.0f, e);                                  //  scales and its reciprocal stay finite; such operands lose precision, not sign)
}

// largest FINITE |x[i]| of a contiguous, 16-B aligned float array (pad entries inside it are zeros): eight float4 loads in flight per
// thread.  NaN / Inf entries do not take part in the scale (the GEMM still propagates them into the rows they belong to).
__device__ __forceinline__ float finite_abs(float v) { const float a = fabsf(v); return a < __builtin_inff() ? a : 0.f; }   // (NaN compares false)
__device__ __forceinline__ float absmax4(float m, const float4 v) {
    return fmaxf(m, fmaxf(fmaxf(finite_abs(v.x), finite_abs(v.y)), fmaxf(finite_abs(v.z), finite_abs(v.w))));
}
// blockIdx.y selects a segment of `seg` floats (a multiple of 4) starting at x + blockIdx.y * seg and its own result word out[blockIdx.y]
// (the last segment ends at n): the filter's right-hand sides get ONE SCALE PER 128-ROW GEMM TILE, so a row's fp16 split depends on the
// rows of its own tile only -- an absurd value in one frame cannot flush the other tiles' operands to zero.
__global__ void __launch_bounds__(256)
absmax_kernel(size_t n, size_t seg, const float* __restrict__ x, unsigned* __restrict__ out) {
    const size_t lo = (size_t)blockIdx.y * seg;
    if (lo >= n) return;
    x += lo; n = n - lo < seg ? n - lo : seg;
    float m = 0.f;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const float4* x4 = (const float4*)x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = absmax4(m, v[u]);
    }
    for (; i < n4; i += stride) m = absmax4(m, x4[i]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, finite_abs(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out + blockIdx.y, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

// X [rows][ld] complex64 (valid: nr rows x P columns) -> packed fp16 planes (layout above).  Workgroup (row tile, chunk of k stages):
// thread (row tid >> 1, half tid & 1) reads its 8 complex values of a stage (two threads = 128 contiguous bytes of a row) and writes
// one 16-B piece of each plane; the 256 pieces of a plane are 4 KB contiguous.
constexpr int H2_PACK_KS = 15;                               // k stages per workgroup
__global__ void __launch_bounds__(256)
pack_h2_kernel(int nr, int P, int ld, int Kpad, const float2* __restrict__ X, const unsigned* __restrict__ amax, _Float16* __restrict__ Q) {
    const int tile = blockIdx.x, r = threadIdx.x >> 1, half = threadIdx.x & 1, row = tile * 128 + r;
    const int nks = Kpad >> 4;
    const float s = h2_scale(__uint_as_float(*amax));
    const int ks1 = min(nks, ((int)blockIdx.y + 1) * H2_PACK_KS);
    for (int ks = (int)blockIdx.y * H2_PACK_KS; ks < ks1; ++ks) {
        h8_t pl[6];
        const int k0 = ks * 16 + half * 8;
        float2 v[8];
        if (row < nr && k0 + 8 <= P && (((uintptr_t)(X + (size_t)row * ld + k0)) & 15) == 0) {
            const float4* src = (const float4*)(X + (size_t)row * ld + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float4 q = src[j]; v[2 * j] = make_float2(q.x, q.y); v[2 * j + 1] = make_float2(q.z, q.w); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (row < nr && k0 + j < P) ? X[(size_t)row * ld + k0 + j] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c[3] = {v[j].x * s, v[j].y * s, (v[j].x + v[j].y) * s};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const _Float16 hi = (_Float16)c[q];
                pl[2 * q][j] = hi;
                pl[2 * q + 1][j] = (_Float16)(c[q] - (float)hi);
            }
        }
        _Float16* base = Q + ((((size_t)tile * nks + ks) * 6) * 128 + r) * 16 + 8 * half;
#pragma unroll
        for (int q = 0; q < 6; ++q) *(h8_t*)(base + (size_t)q * 128 * 16) = pl[q];
    }
}

// Hout[m][i] = sum_k W[i][k] h[m][k]: 128 x 128 block tile, 4 waves x (2 x 2) 32x32 tiles (rows of the MFMA tile = m, columns = i, as
// in cgemm_w_kernel_3m: the epilogue is the same), ONE workgroup per CU (147 KB of LDS, up to 512 registers per lane).
__global__ void __launch_bounds__(256, 1)
cgemm_w_kernel_h2(int P, int Kpad, int M, int ldo, const _Float16* __restrict__ Wq, const float2* __restrict__ Hf, int lda,
                  const unsigned* __restrict__ amax_w, const unsigned* __restrict__ amax_h, float2* __restrict__ Hout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // Tile order: workgroups are dealt round-robin to the 8 XCDs; each XCD walks a contiguous run of tiles in i-major order, so the
    // workgroups resident on an XCD at one time share one or two W panels (2.6 MB of planes each: L2 hits) and stream different h tiles.
    const int nty = (M + T2_M - 1) / T2_M, ntx = (P + T2_N - 1) / T2_N, ntiles = ntx * nty;
    const int per = (ntiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) return;             // (uniform)
    const int bx = tile / nty, by = tile - bx * nty;
    const int i0 = bx * T2_N, m0 = by * T2_M;
    const int wm = (wave & 1) * 64, wi = (wave >> 1) * 64;
    const int nks = Kpad >> 4;
    f32x16 acc[3][2][2];                          // sum re*re, sum im*im, sum (re+im)*(re+im), each already the three-term fp16 product
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[p][x][y][e] = 0.f;
    // a stage of an operand is 1 536 16-B chunks, chunk q = plane * 256 + row * 2 + half: thread tid takes (row tid >> 1, half tid & 1)
    // of every plane
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    // A operand (h): complex64 rows [Mpad][lda], zero padded to the tile and to Kpad, split into its fp16 planes HERE, on the way into
    // LDS (8 B per element from memory instead of 12 B of planes, and no packing pass over the right-hand sides): thread (row tid >> 1,
    // half tid & 1) fetches its 8 complex values of a stage = 64 B; two threads cover 128 contiguous bytes of a row
    const u4_t* gh = (const u4_t*)(Hf + (size_t)(m0 + (tid >> 1)) * lda + (tid & 1) * 8);
    const u4_t* gw = (const u4_t*)(Wq + (size_t)bx * nks * 6 * 128 * 16) + tid;
    const float sh = h2_scale(__uint_as_float(amax_h[by]));     // one scale per 128-row tile of right-hand sides (absmax_kernel)
    const unsigned st_off = (unsigned)((tid >> 1) * H2_ROWB + (tid & 1) * 16);
    // fragment of a 32-row tile: lane l holds row (l & 31), k = 8 (l >> 5) ... + 7: one 16-B read
    const unsigned fr_off = (unsigned)((lane & 31) * H2_ROWB + (lane >> 5) * 16);
    u4_t pa[12];                                 // the next stage, in flight while this one is multiplied
#define H2_FETCH(R, ks)                                                                                                   \
    do {                                                                                                                  \
        const int ks_ = (ks) < nks ? (ks) : nks - 1;             /* (past the end: re-read the last stage, no branch) */    \
        const u4_t* a_ = gh + (size_t)ks_ * 8; const u4_t* b_ = gw + (size_t)ks_ * 1536;     /* a stage = 16 complex = 8 x 16 B of a row */ \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) R[q_] = a_[q_];                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) R[6 + q_] = b_[q_ * 256];                                         \
    } while (0)
#define H2_STAGE(R, buf)                                                                                                  \
    do {                                                                                                                  \
        unsigned char* Hs_ = lds_raw + (size_t)(buf) * 2 * 6 * H2_PLANE + st_off; unsigned char* Ws_ = Hs_ + 6 * H2_PLANE;  \
        h8_t pl_[6];                                                                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                \
            const float c_[2][3] = {{__uint_as_float(R[q_].x) * sh, __uint_as_float(R[q_].y) * sh, 0.f},                    \
                                    {__uint_as_float(R[q_].z) * sh, __uint_as_float(R[q_].w) * sh, 0.f}};                   \
            _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                                            \
                const float v_[3] = {c_[e_][0], c_[e_][1], c_[e_][0] + c_[e_][1]};                                         \
                _Pragma("unroll") for (int z_ = 0; z_ < 3; ++z_) {                                                        \
                    const _Float16 hi_ = (_Float16)v_[z_];                                                                \
                    pl_[2 * z_][2 * q_ + e_] = hi_; pl_[2 * z_ + 1][2 * q_ + e_] = (_Float16)(v_[z_] - (float)hi_);        \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) { *(h8_t*)(Hs_ + q_ * H2_PLANE) = pl_[q_]; *(u4_t*)(Ws_ + q_ * H2_PLANE) = R[6 + q_]; } \
    } while (0)
    // One wave per SIMD: nothing but this wave's own instruction order hides latencies.  The fragments of a real product are read from
    // LDS one product AHEAD of the MFMAs that use them (two fragment sets), the next stage's global loads are issued at the top of a
    // stage and its LDS writes go out between the MFMAs of the stage's last product; sched_barriers pin that order (left alone the
    // compiler sinks every read to the instruction in front of its first use).
    h8_t fa[2][2][2], fb[2][2][2];                // [set][hi | lo][tile]: A (h) and B (W) fragments
#define H2_FRAGS(set, buf, p)                                                                                             \
    do {                                                                                                                  \
        const unsigned char* Hs_ = lds_raw + (size_t)(buf) * 2 * 6 * H2_PLANE + fr_off; const unsigned char* Ws_ = Hs_ + 6 * H2_PLANE; \
        _Pragma("unroll") for (int z_ = 0; z_ < 2; ++z_) {                                                                \
            fa[set][0][z_] = *(const h8_t*)(Hs_ + (2 * (p)) * H2_PLANE + (wm + 32 * z_) * H2_ROWB);                        \
            fa[set][1][z_] = *(const h8_t*)(Hs_ + (2 * (p) + 1) * H2_PLANE + (wm + 32 * z_) * H2_ROWB);                    \
            fb[set][0][z_] = *(const h8_t*)(Ws_ + (2 * (p)) * H2_PLANE + (wi + 32 * z_) * H2_ROWB);                        \
            fb[set][1][z_] = *(const h8_t*)(Ws_ + (2 * (p) + 1) * H2_PLANE + (wi + 32 * z_) * H2_ROWB);                    \
        }                                                                                                                 \
    } while (0)
#define H2_MFMAS(set, p)                                                                                                  \
    do {                                                                                                                  \
        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                                                     \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                               \
                acc[p][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][1][x], fb[set][0][y], acc[p][x][y], 0, 0, 0);   /* small terms first */ \
                acc[p][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][0][x], fb[set][1][y], acc[p][x][y], 0, 0, 0);   \
                acc[p][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][0][x], fb[set][0][y], acc[p][x][y], 0, 0, 0);   \
            }                                                                                                             \
    } while (0)
    // (A second register set -- stage t + 2 in flight while t + 1 waits for its LDS writes -- was tried: the compiler spills 104 registers
    //  and the kernel falls back to the f32 kernel's time.  One stage ahead it is.)
    H2_FETCH(pa, 0);
    H2_STAGE(pa, 0);
    __syncthreads();
    H2_FRAGS(0, 0, 0);
    for (int t = 0; t < nks; ++t) {
        const int buf = t & 1;
        // region 1: the 10 global loads of stage t + 1 go out ONE PER MFMA of product 0 (issued in a block they cost the wave ~200
        // cycles in which its matrix pipe drains); product 1's fragments are read meanwhile
        H2_FRAGS(1, buf, 1);
        H2_FETCH(pa, t + 1);
        H2_MFMAS(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // 8 DS reads
#pragma unroll
        for (int i = 0; i < 10; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }   // MFMA, VMEM read (4 of h + 6 of W)
        __builtin_amdgcn_sched_barrier(0);
        // region 2: product 2's fragments, product 1's MFMAs
        H2_FRAGS(0, buf, 2);
        H2_MFMAS(1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
        // region 3: stage t + 1 goes into the other LDS buffer (free since the barrier that ended stage t - 1), ONE write per MFMA of
        // product 2 (a 1-KB ds_write_b128 occupies the wave ~16-20 cycles; twelve in a row idle the matrix pipe)
        H2_STAGE(pa, buf ^ 1);
        H2_MFMAS(0, 2);
#pragma unroll
        for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }   // MFMA, DS write
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        H2_FRAGS(0, buf ^ 1, 0);                  // product 0 of the next stage
    }
#undef H2_FRAGS
#undef H2_MFMAS
#undef H2_FETCH
#undef H2_STAGE
    const float inv = 1.0f / (h2_scale(__uint_as_float(*amax_w)) * sh);     // exact: powers of two
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int i = i0 + wi + 32 * y + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < M && i < P) {
                    const float p1 = acc[0][x][y][reg], p2 = acc[1][x][y][reg];
                    Hout[(size_t)m * ldo + i] = make_float2((p1 - p2) * inv, (acc[2][x][y][reg] - p1 - p2) * inv);
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

// (Round 5, measured and dropped: TWO row buffers and one barrier per symbol, so that symbol s + 1 is interpolated while slower waves
//  still stream symbol s -- 0.970-0.982 ms per 2 304 config-5 frames against 0.951-0.966 for this one-buffer form in the same process
//  (profiles/r05_interp_two_rows_ab.txt): the second buffer costs two of the six workgroups a CU holds.)
// The tx-replicated form of the pipeline route (one workgroup per (frame, rx)) with the frame kernel's write path: a symbol's
// nsc interpolated values are staged in LDS once, and the T identical rows H_est[b][s][r][0..T) -- ONE contiguous span of
// T * nsc elements -- leave as consecutive 128-B-line-aligned 1 KiB wave stores that read the LDS row modulo nsc.
typedef float v4f_i __attribute__((ext_vector_type(4)));
typedef float v2f_i __attribute__((ext_vector_type(2)));
// MODE 0: the product.  MODE 2 / 3 (diagnostic, chest_tune "interp_mode"): no plan reads -- the pure store pattern -- / no stores -- the
// gather alone.  Round 5 (profiles/r05_interp_modes.txt): 0.92 / 0.92 / 0.13 ms per 2 304 config-5 frames: the kernel IS its store stream,
// and a pure-store model of one workgroup per 38-KB span runs at the same 5.2-5.6 TB/s (tools/micro/store_spans.hip) while a fill of the
// same tensor in 4-KiB tiles runs at 6.9-7.0.  Built, bit-identical, measured and dropped (EXPERIMENTS.md, round 5): next symbol's plan
// records prefetched (0.93), one workgroup per span in memory order (0.92), the page-class form -- every XCD writing only the 4-KiB
// pages of one residue class mod 8, which is what makes the tile fill fast -- (1.10; 0.98 with its gather removed).
template <int MODE>
__global__ void __launch_bounds__(256)
interp_from_pilots_tfold_kernel(int S, int R, int T, int nsc, int Ppad, const PatternDev* __restrict__ patterns, int slot,
                                const float2* __restrict__ h, float2* __restrict__ H_est, int xcd_ranges, int touch_ahead) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const PatternDev pat = patterns[slot];
    const int P = pat.P, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m = blockIdx.x;
    if (xcd_ranges) {
        // Workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of (frame, rx) rows, as the frame kernel
        // does with its frames -- the four rx spans of a (frame, symbol) then leave through one L2 and an XCD's writes are one region.
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, x = m & 7;
        m = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (m >> 3);
    }
    float2* hp = (float2*)lds_raw;                                  // [P rounded up to even]
    float2* row = hp + ((P + 1) & ~1);                              // [nsc]
    const int r = m % R, b = m / R;
    for (int i = tid; i < P; i += blockDim.x) hp[i] = h[(size_t)m * Ppad + i];
    // Touch the pilot values of the row whose workgroup will take this workgroup's place: one dword per 128-B line, landing in
    // this XCD's L2 while this row is interpolated.  Inside the step these values come from HBM (the filter GEMM has just written
    // them) and a cold read inside a saturated write stream waits microseconds -- at the START of a workgroup, with its LDS and
    // wave slots already held.  `touch_ahead` = rows between a workgroup and its successor on the same XCD (0: off).
    uint32_t touched = 0;
    if (touch_ahead > 0 && m + touch_ahead < (int)gridDim.x) {
        const uint32_t* __restrict__ nx = (const uint32_t*)(h + (size_t)(m + touch_ahead) * Ppad);
        const int nlines = (P * 8 + 127) >> 7;
        if (tid < nlines) touched = nx[tid * 32];
    }
    const gptr_plan plan = (gptr_plan)pat.plan;
    const int n = T * nsc;
    const float inv_nsc = 1.0f / (float)nsc;
    for (int s = 0; s < S; ++s) {
        __syncthreads();                                            // hp ready / previous symbol's row fully read
        if constexpr (MODE == 2) {
            for (int k = tid; k < nsc; k += blockDim.x) row[k] = hp[k + s < P ? k + s : 0];
        } else {
            for (int k = tid; k < nsc; k += blockDim.x) row[k] = interp(ld_plan(plan, s * nsc + k), hp, 1, 0);
        }
        __syncthreads();
        float2* __restrict__ g0 = H_est + ((size_t)(b * S + s) * R + r) * T * nsc;
        const int head = (int)(((uintptr_t)g0 & 127u) >> 3);
        const int nst = (head + n + 127) >> 7;
        for (int q = wave; q < (MODE == 3 ? (s == 0 ? 1 : 0) : nst); q += 4) {
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
    asm volatile("" :: "v"(touched));                               // the touch has landed (nothing uses its value)
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
static inline int64_t amax_h_bytes(int M) { return (((int64_t)pad128(M) / 128) * 4 + 63) & ~(int64_t)63; }   // per-tile maxima of the right-hand sides

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
    // h_in [Mpad][Ppad] c64 | h_out [M][Ppad] c64 | Wq: six fp16 planes [Npad][Ppad] (the f32 path's packed W [Npad][Ppad] c64 needs
    // 8 of those 12 bytes per entry) | 16 words: the filter's largest magnitude | one word per 128-row tile: the right-hand sides' maxima
    return Ppad * ((int64_t)sizeof(float2) * (pad128(num_grids) + num_grids) + 12 * (int64_t)pad128(P)) + 64 + amax_h_bytes(num_grids);
}

// W h for M right-hand sides on the matrix pipe.  h_src [M][h_ld] complex64 (rows beyond M and columns beyond P are taken as zero).
// Default: the fp16-split form (cgemm_w_kernel_h2); chest_tune "gemm_f32" 1: the exact-f32 3-multiplication kernel (A/B).
// `Wpacked` != nullptr: the filter as chest_mmse_full_pack_filter left it (planes + its maximum); W is then not looked at.
static int launch_w_gemm(chest_ctx* ctx, hipStream_t st, int P, int M, const float2* h_src, int h_ld, const float2* W, float* h_scratch,
                         bool h_src_is_packed_f32, const void* Wpacked = nullptr) {
    const int Ppad = pad16(P), Mpad = pad128(M), Npad = pad128(P);
    float2* h_in = (float2*)h_scratch;
    float2* h_out = h_in + (size_t)Mpad * Ppad;
    _Float16* Wq = (_Float16*)(h_out + (size_t)M * Ppad);
    unsigned* amax = (unsigned*)(Wq + (size_t)6 * Npad * Ppad);      // [0]: the filter's maximum; [16 ...]: one per row tile of h
    unsigned* amax_h = amax + 16;
    if (ctx->tune_gemm_f32) {
        if (Wpacked != nullptr) { set_error("a packed filter is read by the fp16-split kernel only (chest_tune gemm_f32 is set)"); return CHEST_ESTATE; }
        float2* w_packed = (float2*)Wq;
        const float2* h_op = h_src;
        const int sp_pack = prof_open(ctx, st, CHEST_K_PACK);
        if (!h_src_is_packed_f32) { hipLaunchKernelGGL(pack_h_kernel, dim3(Mpad), dim3(256), 0, st, M, P, h_ld, Ppad, h_src, h_in); h_op = h_in; }
        hipLaunchKernelGGL(pack_w_kernel, dim3(Npad), dim3(256), 0, st, P, Ppad, W, w_packed);
        prof_close(ctx, st, sp_pack);
        CHEST_LDS_OPT_IN(ctx, LDS_GEMM3M, cgemm_w_kernel_3m, T3_LDS_BYTES);
        const int sp_gemm = prof_open(ctx, st, CHEST_K_GEMM);
        hipLaunchKernelGGL(cgemm_w_kernel_3m, dim3(Npad / T2_N, Mpad / T2_M), dim3(256), T3_LDS_BYTES, st, P, Ppad, M, Ppad,
                           (const float2*)w_packed, h_op, h_out);
        prof_close(ctx, st, sp_gemm);
        CHEST_HIP(hipGetLastError());
        return CHEST_OK;
    }
    // the right-hand sides are read as they are if they already are a zero-padded [Mpad][Ppad] array (the frame kernel's h_ls_pilots at
    // whole tiles, the gather of chest_mmse_full_apply); else they are copied into one first
    const int sp_pack = prof_open(ctx, st, CHEST_K_PACK);
    const float2* h_op = h_src;
    int h_op_ld = h_ld;
    if (!h_src_is_packed_f32) { hipLaunchKernelGGL(pack_h_kernel, dim3(Mpad), dim3(256), 0, st, M, P, h_ld, Ppad, h_src, h_in); h_op = h_in; h_op_ld = Ppad; }
    CHEST_HIP(hipMemsetAsync(amax, 0, 64 + (size_t)amax_h_bytes(M), st));
    if (Wpacked == nullptr && ((uintptr_t)W & 15) != 0) { set_error("W must be a 16-byte aligned device pointer"); return CHEST_EINVAL; }
    const _Float16* Wq_use = Wq;
    const unsigned* amax_w = amax;
    if (Wpacked != nullptr) {                                    // the packed filter carries its own maximum behind the planes
        Wq_use = (const _Float16*)Wpacked;
        amax_w = (const unsigned*)((const unsigned char*)Wpacked + (size_t)12 * Npad * Ppad);
    } else {
        hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, (size_t)2 * P * P, (size_t)2 * P * P, (const float*)W, amax);
        hipLaunchKernelGGL(pack_h2_kernel, dim3(Npad / 128, (Ppad / 16 + H2_PACK_KS - 1) / H2_PACK_KS), dim3(256), 0, st, P, P, P, Ppad, W, amax, Wq);
    }
    // (rows m0 .. m0 + 127 of the operand are 256 * h_op_ld contiguous floats, zero pads inside; 16 workgroups per tile)
    hipLaunchKernelGGL(absmax_kernel, dim3(16, (unsigned)(Mpad / 128)), dim3(256), 0, st, (size_t)2 * M * h_op_ld, (size_t)256 * h_op_ld,
                       (const float*)h_op, amax_h);
    prof_close(ctx, st, sp_pack);
    CHEST_LDS_OPT_IN(ctx, LDS_GEMM, cgemm_w_kernel_h2, H2_LDS_BYTES);
    const int sp_gemm = prof_open(ctx, st, CHEST_K_GEMM);
    hipLaunchKernelGGL(cgemm_w_kernel_h2, dim3((((Npad / T2_N) * (Mpad / T2_M) + 7) / 8) * 8), dim3(256), H2_LDS_BYTES, st, P, Ppad, M, Ppad,
                       Wq_use, h_op, h_op_ld, amax_w, (const unsigned*)amax_h, h_out);
    prof_close(ctx, st, sp_gemm);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int chest_mmse_full_apply(chest_ctx* ctx, void* stream, int32_t B, const float* rx4d, const float* tx_pilots,
                          int32_t pilot_stride, int32_t slot, const float* W, float* h_scratch, float* H_est) {
    if (!ctx || B < 0 || (B > 0 && (!rx4d || !tx_pilots || !W || !h_scratch || !H_est))) {
        set_error("chest_mmse_full_apply: bad argument");
        return CHEST_EINVAL;
    }
    if (B == 0) return CHEST_OK;
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr ||
        ctx->patterns[slot].dev.plan == nullptr || ctx->patterns[slot].dev.pad != 0) {
        set_error("pattern slot %d is not set, has no plan or carries a compact plan", slot);
        return CHEST_ESTATE;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R * f.T;
    // scratch layout: chest_mmse_full_apply_scratch_bytes
    const int Mpad = pad128(M);
    float2* h_in = (float2*)h_scratch;
    float2* h_out = h_in + (size_t)Mpad * Ppad;
    hipLaunchKernelGGL(gather_ls_kernel, dim3(Mpad), dim3(256), 0, st, f.S, f.R, f.T, f.nsc, P, Ppad, M, pd.pilot_idx,
                       (const float2*)rx4d, (const float2*)tx_pilots, pilot_stride, h_in);
    {
        const int rc = launch_w_gemm(ctx, st, P, M, h_in, Ppad, (const float2*)W, h_scratch, true);
        if (rc != CHEST_OK) return rc;
    }
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
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr ||
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
    // chest_frames_run writes h_ls_pilots with the entries P .. h_ls_stride - 1 zeroed: with the row stride of the f32 GEMM operand and
    // a whole number of 128-row tiles it IS that kernel's packed operand (no copy); the fp16-split form always repacks (into planes)
    const int Mpad = pad128(M);
    const bool h_in_place = h_ls_stride == Ppad && M == Mpad && ((uintptr_t)h_ls_pilots & 15) == 0;
    return launch_w_gemm(ctx, st, P, M, (const float2*)h_ls_pilots, h_ls_stride, (const float2*)W, h_scratch, h_in_place);
}

int64_t chest_mmse_full_packed_filter_bytes(int32_t P) {
    if (P <= 0) return 0;
    return (int64_t)12 * pad128(P) * pad16(P) + 64;             // six fp16 planes [Npad][Ppad] | the filter's largest magnitude
}

int chest_mmse_full_pack_filter(chest_ctx* ctx, void* stream, int32_t P, const float* W, void* packed) {
    if (!ctx || P <= 0 || !W || !packed || ((uintptr_t)packed & 15) != 0 || ((uintptr_t)W & 15) != 0) {
        set_error("chest_mmse_full_pack_filter: bad argument (W and packed must be 16-byte aligned device pointers)");
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const int Ppad = pad16(P), Npad = pad128(P);
    unsigned* amax = (unsigned*)((unsigned char*)packed + (size_t)12 * Npad * Ppad);
    CHEST_HIP(hipMemsetAsync(amax, 0, 64, st));
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, (size_t)2 * P * P, (size_t)2 * P * P, W, amax);
    hipLaunchKernelGGL(pack_h2_kernel, dim3(Npad / 128, (Ppad / 16 + H2_PACK_KS - 1) / H2_PACK_KS), dim3(256), 0, st, P, P, P, Ppad,
                       (const float2*)W, amax, (_Float16*)packed);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int chest_mmse_full_filter_pilots_packed(chest_ctx* ctx, void* stream, int32_t B, const float* h_ls_pilots, int32_t h_ls_stride,
                                         int32_t slot, const void* packed_filter, float* h_scratch) {
    int rc = full_pilots_check(ctx, "chest_mmse_full_filter_pilots_packed", B, slot, h_scratch);
    if (rc != CHEST_OK || B == 0) return rc;
    if (!h_ls_pilots || !packed_filter) { set_error("chest_mmse_full_filter_pilots_packed: bad argument"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    const auto& f = ctx->fp;
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, Ppad = pad16(P), M = B * f.R;
    if (h_ls_stride < P) { set_error("h_ls_stride %d < %d pilots", h_ls_stride, P); return CHEST_EINVAL; }
    const bool h_in_place = h_ls_stride == Ppad && M == pad128(M) && ((uintptr_t)h_ls_pilots & 15) == 0;
    return launch_w_gemm(ctx, (hipStream_t)stream, P, M, (const float2*)h_ls_pilots, h_ls_stride, nullptr, h_scratch, h_in_place, packed_filter);
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
    const int mode = ctx->tune_interp_mode;
    // rows between a workgroup and the one that follows it in the same CU slot: workgroups per CU (by LDS, at most 8 by waves) x 32 CUs
    // of an XCD, in the XCD's contiguous row range (chest_tune "interp_touch": -1 = off, > 0 = that many rows)
    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / ((lds + 1279) / 1280 * 1280));
    const int touch_ahead = ctx->tune_interp_touch < 0 ? 0 : (ctx->tune_interp_touch > 0 ? ctx->tune_interp_touch : (ctx->tune_interp_xcd >= 0 ? per_cu * 32 : 0));
    const int sp_int = prof_open(ctx, st, CHEST_K_INTERP);
#define CHEST_INTERP_LAUNCH(MODE, OPT)                                                                                             \
    {                                                                                                                              \
        CHEST_LDS_OPT_IN(ctx, OPT, interp_from_pilots_tfold_kernel<MODE>, lds);                                                   \
        hipLaunchKernelGGL(interp_from_pilots_tfold_kernel<MODE>, dim3(M), dim3(256), lds, st, f.S, f.R, f.T, f.nsc, Ppad,        \
                           ctx->d_patterns, slot, h_out, (float2*)H_est, ctx->tune_interp_xcd >= 0 ? 1 : 0, touch_ahead);         \
    }
    if (mode == 2) CHEST_INTERP_LAUNCH(2, LDS_INTERP_TFOLD2)
    else if (mode == 3) CHEST_INTERP_LAUNCH(3, LDS_INTERP_TFOLD3)
    else CHEST_INTERP_LAUNCH(0, LDS_INTERP_TFOLD)
#undef CHEST_INTERP_LAUNCH
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
    if (slot < 0 || slot >= (int)ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
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
