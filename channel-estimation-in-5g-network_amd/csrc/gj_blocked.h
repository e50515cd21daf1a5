// Blocked in-place Gauss-Jordan inversion of a Hermitian / symmetric positive definite matrix in float64, templated on the
// scalar: complex (the full-covariance MMSE filter, mmse_full.hip) or real (the Clough-Tocher gradient system,
// cubic_kernels.hip).
#pragma once
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

struct cdd { double x, y; };
__device__ __forceinline__ cdd zmul(cdd a, cdd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// scalar operations of the sweep
__device__ __forceinline__ void gj_fma(cdd& acc, cdd a, cdd b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ void gj_fma(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ cdd gj_mul(cdd a, cdd b) { return zmul(a, b); }
__device__ __forceinline__ double gj_mul(double a, double b) { return a * b; }
__device__ __forceinline__ cdd gj_sub(cdd a, cdd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ double gj_sub(double a, double b) { return a - b; }
__device__ __forceinline__ cdd gj_neg(cdd a) { return {-a.x, -a.y}; }
__device__ __forceinline__ double gj_neg(double a) { return -a; }
__device__ __forceinline__ cdd gj_recip(cdd a) { const double d2 = a.x * a.x + a.y * a.y; return {a.x / d2, -a.y / d2}; }
__device__ __forceinline__ double gj_recip(double a) { return 1.0 / a; }
__device__ __forceinline__ cdd gj_shfl(cdd v, int src) { return cdd{__shfl(v.x, src, 64), __shfl(v.y, src, 64)}; }
__device__ __forceinline__ double gj_shfl(double v, int src) { return __shfl(v, src, 64); }

// A = R + s2 I is padded with an identity block to a multiple of GJB = 64 rows.  One step per 64-wide pivot block kb, two
// launches per step (2 * ceil(P / 64) launches in all instead of the 2 P of an unblocked sweep):
//   gjb_panel   every workgroup inverts the 64 x 64 pivot block D in LDS (64 pivots, no pivoting: the pivot blocks of a
//               Hermitian positive definite matrix are Schur complements, hence HPD); column-tile workgroups form the new
//               pivot row panel Dinv * A[kb, j] (written to A and to Rp), row-tile workgroups copy the old column panel
//               A[i, kb] to Cp, so that the update below reads neither while it is being overwritten;
//   gjb_update  one workgroup per 64 x 64 tile (i, j), i != kb:  A[i, j] -= Cp_i Rp_j  (j != kb),  A[i, kb] = -Cp_i Dinv;
//               tile (kb, kb) takes Dinv.
// The tile products are complex float64 GEMMs on the vector pipe (register tile 4 x 4 per thread, operands through LDS in
// two k-halves): gfx950's float64 matrix and vector peaks are equal (78.6 TFLOP/s), so v_mfma_f64_16x16x4 buys no rate here.
constexpr int GJB = 64;
template <class T> constexpr size_t gjb_tile_bytes() { return (size_t)GJB * GJB * sizeof(T); }          // cdd: 64 KiB
template <class T> constexpr size_t gjb_panel_lds() { return 2 * gjb_tile_bytes<T>() + (size_t)GJB * 16 * sizeof(T); }   // D | row tile | inversion scratch

// C[64][64] = X[64][64] * Y[64][64] for one thread's 4 x 4 register tile; Xs / Ys are LDS tiles [row][64] / [k][64]
template <class T>
__device__ __forceinline__ void tile_gemm_acc(const T* __restrict__ Xs, const T* __restrict__ Ys, int ty, int tx, int k0, int k1,
                                               T (&acc)[4][4]) {
    for (int k = k0; k < k1; ++k) {
        T a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = Xs[(4 * ty + r) * GJB + k];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = Ys[k * GJB + 4 * tx + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) gj_fma(acc[r][c], a[r], b[c]);
    }
}

// In-place inverse of the 64 x 64 Hermitian positive definite pivot block D (LDS, [64][64]) by a two-level Gauss-Jordan
// sweep: four 16-wide sub-blocks; each is inverted by ONE wave entirely in registers (lane l holds rows l/16 + 4m of
// column l%16; the pivot row / column travel through shuffles: no barrier inside the 16 pivots), then the whole
// workgroup applies the block step to the other 48 rows / columns.  12 workgroup barriers instead of 128.


template <class T>
__device__ __forceinline__ void invert_pivot_block(T* __restrict__ D, T* __restrict__ Cs, int tid) {
    // Cs: LDS scratch [64][16] for the saved sub-column panel
    for (int q = 0; q < GJB / 16; ++q) {
        const int o = 16 * q;
        if (tid < 64) {
            // ---- 16 x 16 diagonal sub-block: register Gauss-Jordan in one wave -----------------------------------
            const int lr = tid >> 4, lc = tid & 15;
            T e[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) e[m] = D[(o + lr + 4 * m) * GJB + o + lc];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const T inv = gj_recip(gj_shfl(e[p >> 2], (p & 3) * 16 + p));        // 1 / element (p, p)
                const T rp = gj_shfl(e[p >> 2], (p & 3) * 16 + lc);                  // (p, my column)
                const T rowv = (lc == p) ? inv : gj_mul(rp, inv);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const T colv = gj_shfl(e[m], lr * 16 + p);                       // (my row, p)
                    const int i = lr + 4 * m;
                    const T qv = gj_mul(colv, rowv);
                    T v = gj_sub(e[m], qv);
                    if (lc == p) v = gj_neg(qv);
                    if (i == p) v = rowv;
                    e[m] = v;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) D[(o + lr + 4 * m) * GJB + o + lc] = e[m];
        }
        // save the old sub-column panel D[:, o..o+16) (rows outside the sub-block)
        for (int x = tid; x < GJB * 16; x += 256) { const int i = x >> 4, c = x & 15; Cs[x] = D[i * GJB + o + c]; }
        __syncthreads();
        // ---- sub-row panel: D[o.., j] = inv * D[o.., j] for the 48 columns j outside the sub-block -----------------
        {
            T acc[3];
            const int r = tid >> 4, jj = tid & 15;                                   // row o + r, columns jj, jj + 16, jj + 32 of the 48
            int col[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int j = jj + 16 * u; col[u] = j < o ? j : j + 16; acc[u] = T{}; }
            for (int k = 0; k < 16; ++k) {
                const T a = D[(o + r) * GJB + o + k];                                // inverse of the sub-block (written above)
#pragma unroll
                for (int u = 0; u < 3; ++u) gj_fma(acc[u], a, D[(o + k) * GJB + col[u]]);
            }
            __syncthreads();                                                         // all reads of the old sub-row panel done
#pragma unroll
            for (int u = 0; u < 3; ++u) D[(o + r) * GJB + col[u]] = acc[u];
        }
        __syncthreads();
        // ---- rest: D[i, j] -= C[i] R[j] (j outside), D[i, o..] = -C[i] inv, for the 48 rows i outside the sub-block -----
        for (int x = tid; x < 48 * GJB; x += 256) {
            const int ii = x / GJB, j = x % GJB;
            const int i = ii < o ? ii : ii + 16;
            const bool incol = j >= o && j < o + 16;
            T acc{};
            for (int k = 0; k < 16; ++k) gj_fma(acc, Cs[i * 16 + k], D[(o + k) * GJB + j]);   // j inside: the inverse; outside: the new sub-row panel
            D[i * GJB + j] = incol ? gj_neg(acc) : gj_sub(D[i * GJB + j], acc);
        }
        __syncthreads();
    }
}

template <class T>
__global__ void __launch_bounds__(256)
gjb_panel_kernel(int Ppad, int kb, T* __restrict__ A, T* __restrict__ Dinv_g, T* __restrict__ Rp, T* __restrict__ Cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* D = (T*)lds_raw;                            // [64][64] pivot block, inverted in place
    T* Ts = D + GJB * GJB;                         // [64][64] this workgroup's tile of the pivot row
    T* Cs = Ts + GJB * GJB;                        // [64][16] scratch of the block inversion
    const int tid = threadIdx.x, nbk = Ppad / GJB, w = blockIdx.x;
    const size_t k0 = (size_t)kb * GJB;
    if (w >= nbk) {                                // row-tile job: save the old column panel A[i, kb]
        const int i = w - nbk;
        for (int e = tid; e < GJB * GJB; e += 256) {
            const int r = e / GJB, c = e % GJB;
            Cp[((size_t)i * GJB + r) * GJB + c] = A[((size_t)i * GJB + r) * Ppad + k0 + c];
        }
        return;
    }
    for (int e = tid; e < GJB * GJB; e += 256) D[e] = A[(k0 + e / GJB) * Ppad + k0 + e % GJB];
    if (w != kb)
        for (int e = tid; e < GJB * GJB; e += 256) Ts[e] = A[(k0 + e / GJB) * Ppad + (size_t)w * GJB + e % GJB];
    __syncthreads();
    invert_pivot_block(D, Cs, tid);
    if (w == kb) {                                 // the inverse itself (the update kernel copies it into A[kb, kb])
        for (int e = tid; e < GJB * GJB; e += 256) Dinv_g[e] = D[e];
        return;
    }
    // pivot row panel tile: Dinv * A[kb, w]
    const int ty = tid / 16, tx = tid % 16;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = T{};
    tile_gemm_acc(D, Ts, ty, tx, 0, GJB, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * ty + r, j = 4 * tx + c;
            A[(k0 + i) * Ppad + (size_t)w * GJB + j] = acc[r][c];
            Rp[(size_t)i * Ppad + (size_t)w * GJB + j] = acc[r][c];
        }
}

template <class T>
__global__ void __launch_bounds__(256)
gjb_update_kernel(int Ppad, int kb, T* __restrict__ A, const T* __restrict__ Dinv_g, const T* __restrict__ Rp,
                  const T* __restrict__ Cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* Xs = (T*)lds_raw;                           // [64 rows][64 k]: Cp_i
    T* Ys = Xs + GJB * GJB;                        // [64 k][64 cols]: Rp_j or Dinv
    const int tid = threadIdx.x, ti = blockIdx.y, tj = blockIdx.x;
    const size_t k0 = (size_t)kb * GJB;
    if (ti == kb) {
        if (tj == kb) for (int e = tid; e < GJB * GJB; e += 256) A[(k0 + e / GJB) * Ppad + k0 + e % GJB] = Dinv_g[e];
        return;                                    // the rest of the pivot row was written by the panel kernel
    }
    const bool colpanel = tj == kb;
    for (int e = tid; e < GJB * GJB; e += 256) {
        Xs[e] = Cp[(size_t)ti * GJB * GJB + e];
        Ys[e] = colpanel ? Dinv_g[e] : Rp[(size_t)(e / GJB) * Ppad + (size_t)tj * GJB + e % GJB];
    }
    __syncthreads();
    const int ty = tid / 16, tx = tid % 16;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = T{};
    tile_gemm_acc(Xs, Ys, ty, tx, 0, GJB, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            T* dst = A + ((size_t)ti * GJB + 4 * ty + r) * Ppad + (size_t)tj * GJB + 4 * tx + c;
            *dst = colpanel ? gj_neg(acc[r][c]) : gj_sub(*dst, acc[r][c]);
        }
}


// workspace elements (of T) behind the padded matrix A [Ppad][Ppad]: Rp [64][Ppad] | Cp [Ppad/64][64][64] | Dinv [64][64]
inline size_t gj_aux_elems(int Ppad) { return (size_t)2 * GJB * Ppad + (size_t)GJB * GJB; }

// In-place inverse of the padded matrix A [Ppad][Ppad] (Ppad a multiple of 64; pad with an identity block): 2 Ppad / 64 launches.
template <class T>
int launch_gj_inverse(chest_ctx* ctx, hipStream_t st, int Ppad, T* A, T* aux, int lds_slot_panel, int lds_slot_update) {
    const int nbk = Ppad / GJB;
    T* Rp = aux;
    T* Cp = Rp + (size_t)GJB * Ppad;
    T* Dinv = Cp + (size_t)GJB * Ppad;
    CHEST_LDS_OPT_IN(ctx, lds_slot_panel, gjb_panel_kernel<T>, gjb_panel_lds<T>());
    CHEST_LDS_OPT_IN(ctx, lds_slot_update, gjb_update_kernel<T>, 2 * gjb_tile_bytes<T>());
    for (int kb = 0; kb < nbk; ++kb) {
        hipLaunchKernelGGL(gjb_panel_kernel<T>, dim3((unsigned)(2 * nbk)), dim3(256), gjb_panel_lds<T>(), st, Ppad, kb, A, Dinv, Rp, Cp);
        hipLaunchKernelGGL(gjb_update_kernel<T>, dim3((unsigned)nbk, (unsigned)nbk), dim3(256), 2 * gjb_tile_bytes<T>(), st, Ppad, kb, A,
                           (const T*)Dinv, (const T*)Rp, (const T*)Cp);
    }
    return CHEST_OK;
}

}  // namespace chest
