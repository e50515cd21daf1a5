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

// 16 x 16 result blocks on the float64 matrix pipe (v_mfma_f64_16x16x4_f64: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4]
// [col = lane & 15], result register j of a lane = element (row (lane >> 4) + 4 j, col lane & 15)).  The rate equals the
// vector pipe's on gfx950; what it buys is one instruction and two LDS reads per 1 024 multiply-adds instead of sixteen and eight.
typedef double v4d_t __attribute__((ext_vector_type(4)));
template <class T> struct GjAcc;
template <> struct GjAcc<double> {
    v4d_t r;
    __device__ __forceinline__ void zero() { r = v4d_t{0.0, 0.0, 0.0, 0.0}; }
    __device__ __forceinline__ void mac(double a, double b) { r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, r, 0, 0, 0); }
    __device__ __forceinline__ double get(int j) const { return r[j]; }
};
template <> struct GjAcc<cdd> {
    v4d_t r, i;
    __device__ __forceinline__ void zero() { r = v4d_t{0.0, 0.0, 0.0, 0.0}; i = r; }
    __device__ __forceinline__ void mac(cdd a, cdd b) {
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, r, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, r, 0, 0, 0);
        i = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, i, 0, 0, 0);
        i = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, i, 0, 0, 0);
    }
    __device__ __forceinline__ cdd get(int j) const { return cdd{r[j], i[j]}; }
};

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
    // Cs: LDS scratch [16][64] for the saved sub-column panel (transposed)
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
        // save the old sub-column panel D[:, o..o+16), transposed (CsT[c][i]: consecutive rows on consecutive lanes of an A read)
        for (int x = tid; x < GJB * 16; x += 256) { const int i = x & (GJB - 1), c = x >> 6; Cs[c * GJB + i] = D[i * GJB + o + c]; }
        __syncthreads();
        const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
        // ---- sub-row panel: D[o.., j] = inv * D[o.., j] for the three 16-column blocks outside the sub-block, one per wave
        //      (16 x 16 x 16 products on the matrix pipe) ------------------------------------------------------------
        const int cbk = wv < q ? wv : wv + 1;
        GjAcc<T> pr;
        pr.zero();
        if (wv < 3) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + l4;
                pr.mac(D[(o + l15) * GJB + o + k], D[(o + k) * GJB + 16 * cbk + l15]);
            }
        }
        __syncthreads();                                                             // all reads of the old sub-row panel done
        if (wv < 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) D[(o + l4 + 4 * j) * GJB + 16 * cbk + l15] = pr.get(j);
        }
        __syncthreads();
        // ---- rest: D[i, j] -= C[i] R[j] (j outside), D[i, o..] = -C[i] inv, for the three 16-row blocks outside the sub-block;
        //      wave wv owns column block wv (rows o.. of D hold the inverse inside the sub-block and the new panel outside) ----
#pragma unroll
        for (int rbi = 0; rbi < 3; ++rbi) {
            const int rbk = rbi < q ? rbi : rbi + 1;
            GjAcc<T> acc;
            acc.zero();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + l4;
                acc.mac(Cs[k * GJB + 16 * rbk + l15], D[(o + k) * GJB + 16 * wv + l15]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T* d = D + (16 * rbk + l4 + 4 * j) * GJB + 16 * wv + l15;
                *d = (wv == q) ? gj_neg(acc.get(j)) : gj_sub(*d, acc.get(j));
            }
        }
        __syncthreads();
    }
}

// The first pivot block's inverse (one workgroup); every later one is produced by the update launch of the step before it.
template <class T>
__global__ void __launch_bounds__(256)
gjb_first_pivot_kernel(int Ppad, const T* __restrict__ A, T* __restrict__ Dinv_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* D = (T*)lds_raw;
    T* Cs = D + GJB * GJB;
    const int tid = threadIdx.x;
    for (int e = tid; e < GJB * GJB; e += 256) D[e] = A[(size_t)(e / GJB) * Ppad + e % GJB];
    __syncthreads();
    invert_pivot_block(D, Cs, tid);
    for (int e = tid; e < GJB * GJB; e += 256) Dinv_g[e] = D[e];
}

// Step kb, first launch.  Column-tile workgroups w < nbk form the new pivot row panel Dinv * A[kb, w] (written to A and to Rp);
// row-tile workgroups w >= nbk copy the old column panel A[i, kb] to Cp, so that the update reads neither while it is being
// overwritten.  Dinv (the inverse of the pivot block) is already there: look-ahead of the previous update launch.
template <class T>
__global__ void __launch_bounds__(256)
gjb_panel_kernel(int Ppad, int kb, T* __restrict__ A, const T* __restrict__ Dinv_g, T* __restrict__ Rp, T* __restrict__ Cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* D = (T*)lds_raw;                            // [64][64] inverse of the pivot block
    T* Ts = D + GJB * GJB;                         // [64][64] this workgroup's tile of the pivot row
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
    if (w == kb) return;                           // A[kb, kb] = Dinv is written by the update launch
    for (int e = tid; e < GJB * GJB; e += 256) {
        D[e] = Dinv_g[e];
        Ts[e] = A[(k0 + e / GJB) * Ppad + (size_t)w * GJB + e % GJB];
    }
    __syncthreads();
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

// Step kb, second launch: one workgroup per 64 x 64 tile (i, j), i != kb:  A[i, j] -= Cp_i Rp_j  (j != kb),
// A[i, kb] = -Cp_i Dinv; tile (kb, kb) takes Dinv.  LOOK-AHEAD: the workgroup of tile (kb+1, kb+1) holds the next pivot block
// in its registers after the update; it inverts it on the spot (LDS is free by then) and leaves the inverse in Dinv_next, so
// that the 64-pivot sweep never sits on the critical path between two launches.
template <class T>
__global__ void __launch_bounds__(256)
gjb_update_kernel(int Ppad, int kb, T* __restrict__ A, const T* __restrict__ Dinv_g, T* __restrict__ Dinv_next, const T* __restrict__ Rp,
                  const T* __restrict__ Cp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* Xs = (T*)lds_raw;                           // one tile: the two operand halves; later the next pivot block
    T* Ys = Xs + GJB * GJB;                        // [64][16] scratch of the look-ahead inversion
    const int tid = threadIdx.x;
    // the look-ahead tile (kb+1, kb+1) carries the longest job of the launch (update + 64-pivot sweep): it trades places with
    // tile (0, 0) so that it is dispatched first
    int ti = blockIdx.y, tj = blockIdx.x;
    if (kb + 1 < (int)gridDim.x) {
        if (ti == 0 && tj == 0) ti = tj = kb + 1;
        else if (ti == kb + 1 && tj == kb + 1) ti = tj = 0;
    }
    const size_t k0 = (size_t)kb * GJB;
    if (ti == kb) {
        if (tj == kb) for (int e = tid; e < GJB * GJB; e += 256) A[(k0 + e / GJB) * Ppad + k0 + e % GJB] = Dinv_g[e];
        return;                                    // the rest of the pivot row was written by the panel kernel
    }
    const bool colpanel = tj == kb;
    // wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the tile as 2 x 2 MFMA blocks
    const int lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1, l15 = lane & 15, l4 = lane >> 4;
    GjAcc<T> acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb].zero();
    // operands through LDS in two k-halves (X half [64][32 + 1 pad], Y half [32][64]): half the footprint, so that two
    // workgroups share a CU and one's loads overlap the other's arithmetic; the pad keeps the 16 rows of an A read apart
    constexpr int KH = GJB / 2, XP = KH + 1;
    T* Xh = Xs;                                    // [64][XP]
    T* Yh = Xs + GJB * XP;                         // [KH][64]
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        for (int e = tid; e < GJB * KH; e += 256) {
            const int r = e / KH, kk = e % KH;
            Xh[r * XP + kk] = Cp[(size_t)ti * GJB * GJB + (size_t)r * GJB + half * KH + kk];
            const int k = e / GJB, c = e % GJB;
            Yh[e] = colpanel ? Dinv_g[(half * KH + k) * GJB + c] : Rp[(size_t)(half * KH + k) * Ppad + (size_t)tj * GJB + c];
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < KH / 4; ++ks) {
            const int k = 4 * ks + l4;
            T av[2], bv[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) av[rb] = Xh[(32 * wr + 16 * rb + l15) * XP + k];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) bv[cb] = Yh[k * GJB + 32 * wc + 16 * cb + l15];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[rb][cb].mac(av[rb], bv[cb]);
        }
    }
    const bool next_pivot = ti == kb + 1 && tj == kb + 1;          // workgroup-uniform
    if (next_pivot) __syncthreads();                               // every thread is done reading the operand halves
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 32 * wr + 16 * rb + l4 + 4 * j, col = 32 * wc + 16 * cb + l15;
                T* dst = A + ((size_t)ti * GJB + row) * Ppad + (size_t)tj * GJB + col;
                const T v = colpanel ? gj_neg(acc[rb][cb].get(j)) : gj_sub(*dst, acc[rb][cb].get(j));
                *dst = v;
                if (next_pivot) Xs[row * GJB + col] = v;
            }
    if (next_pivot) {
        __syncthreads();
        invert_pivot_block(Xs, Ys, tid);
        for (int e = tid; e < GJB * GJB; e += 256) Dinv_next[e] = Xs[e];
    }
}


// workspace elements (of T) behind the padded matrix A [Ppad][Ppad]: Rp [64][Ppad] | Cp [Ppad/64][64][64] | Dinv x 2 [64][64]
inline size_t gj_aux_elems(int Ppad) { return (size_t)2 * GJB * Ppad + (size_t)2 * GJB * GJB; }

// In-place inverse of the padded matrix A [Ppad][Ppad] (Ppad a multiple of 64; pad with an identity block): 2 Ppad / 64 + 1 launches.
template <class T>
int launch_gj_inverse(chest_ctx* ctx, hipStream_t st, int Ppad, T* A, T* aux, int lds_slot_panel, int lds_slot_update) {
    const int nbk = Ppad / GJB;
    T* Rp = aux;
    T* Cp = Rp + (size_t)GJB * Ppad;
    T* Dinv[2] = {Cp + (size_t)GJB * Ppad, Cp + (size_t)GJB * Ppad + (size_t)GJB * GJB};
    CHEST_LDS_OPT_IN(ctx, lds_slot_panel, gjb_panel_kernel<T>, 2 * gjb_tile_bytes<T>());
    const size_t lds_update = gjb_tile_bytes<T>() + (size_t)GJB * 16 * sizeof(T);          // 80 KiB (complex): two per CU
    CHEST_LDS_OPT_IN(ctx, lds_slot_update, gjb_update_kernel<T>, lds_update);
    // the first-pivot kernel shares the panel kernel's LDS slot bookkeeping (same size class)
    if (hipFuncSetAttribute((const void*)gjb_first_pivot_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gjb_panel_lds<T>()) != hipSuccess) {
        set_error("hipFuncSetAttribute(gjb_first_pivot_kernel) failed");
        return CHEST_EHIP;
    }
    hipLaunchKernelGGL(gjb_first_pivot_kernel<T>, dim3(1), dim3(256), gjb_panel_lds<T>(), st, Ppad, (const T*)A, Dinv[0]);
    for (int kb = 0; kb < nbk; ++kb) {
        hipLaunchKernelGGL(gjb_panel_kernel<T>, dim3((unsigned)(2 * nbk)), dim3(256), 2 * gjb_tile_bytes<T>(), st, Ppad, kb, A, (const T*)Dinv[kb & 1], Rp, Cp);
        hipLaunchKernelGGL(gjb_update_kernel<T>, dim3((unsigned)nbk, (unsigned)nbk), dim3(256), lds_update, st, Ppad, kb, A,
                           (const T*)Dinv[kb & 1], Dinv[(kb + 1) & 1], (const T*)Rp, (const T*)Cp);
    }
    return CHEST_OK;
}

}  // namespace chest
