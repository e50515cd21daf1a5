// On-device construction of the dense Clough-Tocher ('cubic') interpolation operator of a pilot pattern.
//
//   reference: LSEstimator('cubic') -> griddata(method='cubic') (src/baseline_estimators.py:65-79) =
//   scipy.interpolate.CloughTocher2DInterpolator: Delaunay triangulation, global gradient estimation at the pilots,
//   C1 piecewise-cubic Bezier patch per triangle.  Every step is linear in the pilot values, so the interpolant is a dense
//   operator A [G][P]; tools/ct_model.py is the float64 CPU model of exactly these steps (equal to SciPy's interpolator to
//   7e-13 at a tight solver tolerance, 6e-7 at SciPy's default tol = 1e-6).
//
//   1. gradient system.  SciPy iterates Gauss-Seidel on
//        sum_j (4 e e^T / L^3) g_i + sum_j (2 e e^T / L^3) g_j = 6 sum_j (f_j - f_i) e / L^3      (j: edge neighbours of i,
//      e = p_j - p_i, L = |e|); here the symmetric positive definite 2P x 2P matrix M is assembled from the unique edges of
//      the triangulation (float64 atomics) and inverted directly with the blocked Gauss-Jordan sweep (gj_blocked.h).
//   2. gradient operator Gop = M^-1 B (2P x P): per unique edge (i, j) the vector u = (Minv[:, 2i:2i+2] + Minv[:, 2j:2j+2])
//      (6 e / L^3) is added to column j and subtracted from column i.
//   3. per resource element: containing triangle, barycentric coordinates, the neighbour-centroid factors g_k and the 9
//      linear coefficients of the patch value in (f, grad f) of the three vertices (the patch evaluated on unit inputs).
//   4. A[g][:] = sum_v cf_v delta_v + cg_v . Gop[2v:2v+2][:]  (float32 out), 0 outside the convex hull (griddata fill_value).
#include "chest_device.h"
#include "chest_internal.h"
#include "gj_blocked.h"

namespace chest {

struct CtCoef { int32_t v[3]; int32_t tri; double cf[3]; double cg[6]; };      // 96 bytes per resource element

__global__ void __launch_bounds__(256)
ct_init_kernel(int M2, int Mpad, double* __restrict__ M, size_t gop_elems, double* __restrict__ Gop) {
    const size_t n = (size_t)Mpad * Mpad, tot = n + gop_elems;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        if (e < n) { const int i = (int)(e / Mpad), j = (int)(e % Mpad); M[e] = (i == j && i >= M2) ? 1.0 : 0.0; }
        else Gop[e - n] = 0.0;
    }
}

__device__ __forceinline__ void pt_of(const int32_t* __restrict__ pil, int nsc, int i, double& x, double& y) {
    const int g = pil[i];
    const int s = g / nsc;
    x = (double)s; y = (double)(g - s * nsc);
}

// one thread per (triangle, edge slot); an edge shared by two triangles is taken from the smaller triangle index
__global__ void __launch_bounds__(256)
ct_edges_kernel(int ns, int nsc, int Mpad, const int32_t* __restrict__ pil, const int32_t* __restrict__ simp,
                const int32_t* __restrict__ nbr, double* __restrict__ M) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 3 * ns) return;
    const int t = id / 3, k = id % 3;
    const int nb = nbr[id];
    if (nb >= 0 && nb < t) return;
    const int i = simp[3 * t + (k + 1) % 3], j = simp[3 * t + (k + 2) % 3];
    double xi, yi, xj, yj;
    pt_of(pil, nsc, i, xi, yi); pt_of(pil, nsc, j, xj, yj);
    const double ex = xj - xi, ey = yj - yi, L = sqrt(ex * ex + ey * ey), L3 = L * L * L;
    const double exx = ex * ex / L3, exy = ex * ey / L3, eyy = ey * ey / L3;
    auto add2x2 = [&](int r0, int c0, double f) {
        atomicAdd(&M[(size_t)r0 * Mpad + c0], f * exx); atomicAdd(&M[(size_t)r0 * Mpad + c0 + 1], f * exy);
        atomicAdd(&M[(size_t)(r0 + 1) * Mpad + c0], f * exy); atomicAdd(&M[(size_t)(r0 + 1) * Mpad + c0 + 1], f * eyy);
    };
    add2x2(2 * i, 2 * i, 4.0); add2x2(2 * i, 2 * j, 2.0);
    add2x2(2 * j, 2 * j, 4.0); add2x2(2 * j, 2 * i, 2.0);       // e -> -e leaves e e^T unchanged
}

// Gop[r][:] for one unique edge per blockIdx.y, 256 rows per blockIdx.x
__global__ void __launch_bounds__(256)
ct_gop_kernel(int ns, int nsc, int P, int Mpad, const int32_t* __restrict__ pil, const int32_t* __restrict__ simp,
              const int32_t* __restrict__ nbr, const double* __restrict__ Minv, double* __restrict__ Gop) {
    const int id = blockIdx.y;
    const int t = id / 3, k = id % 3;
    const int nb = nbr[id];
    if (nb >= 0 && nb < t) return;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2 * P) return;
    const int i = simp[3 * t + (k + 1) % 3], j = simp[3 * t + (k + 2) % 3];
    double xi, yi, xj, yj;
    pt_of(pil, nsc, i, xi, yi); pt_of(pil, nsc, j, xj, yj);
    const double ex = xj - xi, ey = yj - yi, L = sqrt(ex * ex + ey * ey), s6 = 6.0 / (L * L * L);
    // Minv is symmetric: column 2i of row r = row 2i, entry r (contiguous in r)
    const double u = (Minv[(size_t)(2 * i) * Mpad + r] + Minv[(size_t)(2 * j) * Mpad + r]) * (s6 * ex) +
                     (Minv[(size_t)(2 * i + 1) * Mpad + r] + Minv[(size_t)(2 * j + 1) * Mpad + r]) * (s6 * ey);
    atomicAdd(&Gop[(size_t)r * P + j], u);
    atomicAdd(&Gop[(size_t)r * P + i], -u);
}

// containing triangle of every resource element (exact integer tests; first hit in simplex order) and the 9 patch coefficients
__global__ void __launch_bounds__(256)
ct_coef_kernel(int G, int nsc, int ns, const int32_t* __restrict__ pil, const int32_t* __restrict__ simp,
               const int32_t* __restrict__ nbr, CtCoef* __restrict__ out) {
    __shared__ int4 tri[512];                       // x: v0, y: v1, z: v2 packed (row << 16 | col); w: index
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = g < G;
    const int qs = act ? g / nsc : 0, qk = act ? g - qs * nsc : 0;
    int found = -1;
    for (int base = 0; base < ns; base += 512) {
        __syncthreads();
        for (int u = threadIdx.x; u < 512 && base + u < ns; u += blockDim.x) {
            int v[3];
            for (int c = 0; c < 3; ++c) { const int gi = pil[simp[3 * (base + u) + c]]; const int s = gi / nsc; v[c] = (s << 16) | (gi - s * nsc); }
            tri[u] = make_int4(v[0], v[1], v[2], base + u);
        }
        __syncthreads();
        const int n = min(512, ns - base);
        if (found < 0 && act)
            for (int u = 0; u < n; ++u) {
                const int4 tq = tri[u];
                const int x0 = (tq.x >> 16) - qs, y0 = (tq.x & 0xFFFF) - qk, x1 = (tq.y >> 16) - qs, y1 = (tq.y & 0xFFFF) - qk,
                          x2 = (tq.z >> 16) - qs, y2 = (tq.z & 0xFFFF) - qk;
                int n0 = x1 * y2 - y1 * x2, n1 = x2 * y0 - y2 * x0, n2 = x0 * y1 - y0 * x1;
                int den = n0 + n1 + n2;
                if (den == 0) continue;
                if (den < 0) { n0 = -n0; n1 = -n1; n2 = -n2; }
                if ((n0 | n1 | n2) < 0) continue;
                found = tq.w;
                break;
            }
    }
    if (!act) return;
    CtCoef o;
    o.tri = found;
    for (int c = 0; c < 3; ++c) { o.v[c] = 0; o.cf[c] = 0.0; }
    for (int c = 0; c < 6; ++c) o.cg[c] = 0.0;
    if (found >= 0) {
        double px[3], py[3];
        for (int c = 0; c < 3; ++c) { o.v[c] = simp[3 * found + c]; pt_of(pil, nsc, o.v[c], px[c], py[c]); }
        // barycentric transform of this triangle
        const double t00 = px[0] - px[2], t01 = px[1] - px[2], t10 = py[0] - py[2], t11 = py[1] - py[2];
        const double det = t00 * t11 - t01 * t10;
        const double i00 = t11 / det, i01 = -t01 / det, i10 = -t10 / det, i11 = t00 / det;
        auto bary = [&](double x, double y, double* bq) {
            const double dx = x - px[2], dy = y - py[2];
            bq[0] = i00 * dx + i01 * dy; bq[1] = i10 * dx + i11 * dy; bq[2] = 1.0 - bq[0] - bq[1];
        };
        double b[3];
        bary((double)qs, (double)qk, b);
        double gk[3];
        for (int k = 0; k < 3; ++k) {
            const int nt = nbr[3 * found + k];
            if (nt < 0) { gk[k] = -0.5; continue; }
            double cx = 0.0, cy = 0.0;
            for (int c = 0; c < 3; ++c) { double x, y; pt_of(pil, nsc, simp[3 * nt + c], x, y); cx += x; cy += y; }
            double c[3];
            bary(cx / 3.0, cy / 3.0, c);
            const int a1 = (k + 2) % 3, a2 = (k + 1) % 3;       // k = 0: (c2, c1); k = 1: (c0, c2); k = 2: (c1, c0)
            gk[k] = (2.0 * c[a1] + c[a2] - 1.0) / (2.0 - 3.0 * c[a1] - 3.0 * c[a2]);
        }
        const double e12x = px[1] - px[0], e12y = py[1] - py[0], e23x = px[2] - px[1], e23y = py[2] - py[1],
                     e31x = px[0] - px[2], e31y = py[0] - py[2];
        double mn = fmin(b[0], fmin(b[1], b[2]));
        const double b1 = b[0] - mn, b2 = b[1] - mn, b3 = b[2] - mn, b4 = 3.0 * mn;
        for (int unit = 0; unit < 9; ++unit) {
            double f[3] = {0.0, 0.0, 0.0}, dfx[3] = {0.0, 0.0, 0.0}, dfy[3] = {0.0, 0.0, 0.0};
            if (unit < 3) f[unit] = 1.0;
            else if ((unit - 3) % 2 == 0) dfx[(unit - 3) / 2] = 1.0;
            else dfy[(unit - 3) / 2] = 1.0;
            const double df12 = dfx[0] * e12x + dfy[0] * e12y, df21 = -(dfx[1] * e12x + dfy[1] * e12y);
            const double df23 = dfx[1] * e23x + dfy[1] * e23y, df32 = -(dfx[2] * e23x + dfy[2] * e23y);
            const double df31 = dfx[2] * e31x + dfy[2] * e31y, df13 = -(dfx[0] * e31x + dfy[0] * e31y);
            const double c3000 = f[0], c2100 = (df12 + 3 * c3000) / 3, c2010 = (df13 + 3 * c3000) / 3;
            const double c0300 = f[1], c1200 = (df21 + 3 * c0300) / 3, c0210 = (df23 + 3 * c0300) / 3;
            const double c0030 = f[2], c1020 = (df31 + 3 * c0030) / 3, c0120 = (df32 + 3 * c0030) / 3;
            const double c2001 = (c2100 + c2010 + c3000) / 3, c0201 = (c1200 + c0300 + c0210) / 3, c0021 = (c1020 + c0120 + c0030) / 3;
            const double c0111 = (gk[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2;
            const double c1011 = (gk[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2;
            const double c1101 = (gk[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2;
            const double c1002 = (c1101 + c1011 + c2001) / 3, c0102 = (c1101 + c0111 + c0201) / 3, c0012 = (c1011 + c0111 + c0021) / 3;
            const double c0003 = (c1002 + c0102 + c0012) / 3;
            const double w = b1 * b1 * b1 * c3000 + 3 * b1 * b1 * b2 * c2100 + 3 * b1 * b1 * b3 * c2010 + 3 * b1 * b1 * b4 * c2001 +
                             3 * b1 * b2 * b2 * c1200 + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * b3 * b3 * c1020 + 6 * b1 * b3 * b4 * c1011 +
                             3 * b1 * b4 * b4 * c1002 + b2 * b2 * b2 * c0300 + 3 * b2 * b2 * b3 * c0210 + 3 * b2 * b2 * b4 * c0201 +
                             3 * b2 * b3 * b3 * c0120 + 6 * b2 * b3 * b4 * c0111 + 3 * b2 * b4 * b4 * c0102 + b3 * b3 * b3 * c0030 +
                             3 * b3 * b3 * b4 * c0021 + 3 * b3 * b4 * b4 * c0012 + b4 * b4 * b4 * c0003;
            if (unit < 3) o.cf[unit] = w; else o.cg[unit - 3] = w;
        }
    }
    out[g] = o;
}

// A[g][c] = sum_v cf_v [c == v] + cgx_v Gop[2v][c] + cgy_v Gop[2v+1][c];  one workgroup row per resource element
__global__ void __launch_bounds__(256)
ct_apply_kernel(int G, int P, const CtCoef* __restrict__ coef, const double* __restrict__ Gop, float* __restrict__ A) {
    const int g = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P) return;
    const CtCoef q = coef[g];
    double a = 0.0;
    if (q.tri >= 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            a += q.cg[2 * v] * Gop[(size_t)(2 * q.v[v]) * P + c] + q.cg[2 * v + 1] * Gop[(size_t)(2 * q.v[v] + 1) * P + c];
            if (c == q.v[v]) a += q.cf[v];
        }
    }
    A[(size_t)g * P + c] = (float)a;
}

}  // namespace chest

using namespace chest;

extern "C" int64_t chest_cubic_workspace_bytes(chest_ctx* ctx, int32_t num_pilots) {
    if (!ctx || num_pilots <= 0) return 0;
    const int64_t Mpad = ((int64_t)2 * num_pilots + GJB - 1) / GJB * GJB;
    return (int64_t)sizeof(double) * (Mpad * Mpad + (int64_t)gj_aux_elems((int)Mpad) + (int64_t)2 * num_pilots * num_pilots) +
           (int64_t)sizeof(CtCoef) * ctx->fp.G + 256;
}

extern "C" int chest_cubic_operator(chest_ctx* ctx, void* stream, int32_t slot, const int32_t* simplices, const int32_t* neighbors,
                                    int32_t num_simplices, float* A, void* workspace, int64_t workspace_bytes) {
    if (!ctx || !simplices || !neighbors || num_simplices < 1 || !A || !workspace) { set_error("chest_cubic_operator: bad argument"); return CHEST_EINVAL; }
    if (slot < 0 || (size_t)slot >= ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    const PatternDev& pd = ctx->patterns[slot].dev;
    const int P = pd.P, G = ctx->fp.G, nsc = ctx->fp.nsc;
    if (workspace_bytes < chest_cubic_workspace_bytes(ctx, P)) {
        set_error("chest_cubic_operator: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)chest_cubic_workspace_bytes(ctx, P));
        return CHEST_EINVAL;
    }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const int M2 = 2 * P, Mpad = (M2 + GJB - 1) / GJB * GJB;
    double* M = (double*)workspace;
    double* aux = M + (size_t)Mpad * Mpad;
    double* Gop = aux + gj_aux_elems(Mpad);
    CtCoef* coef = (CtCoef*)(((uintptr_t)(Gop + (size_t)M2 * P) + 15) & ~(uintptr_t)15);
    const int sp = prof_open(ctx, st, CHEST_K_PLAN);
    hipLaunchKernelGGL(ct_init_kernel, dim3(2048), dim3(256), 0, st, M2, Mpad, M, (size_t)M2 * P, Gop);
    hipLaunchKernelGGL(ct_edges_kernel, dim3((unsigned)((3 * num_simplices + 255) / 256)), dim3(256), 0, st, num_simplices, nsc, Mpad,
                       pd.pilot_idx, simplices, neighbors, M);
    const int rc = launch_gj_inverse<double>(ctx, st, Mpad, M, aux, LDS_GJ_PANEL_R, LDS_GJ_UPDATE_R);
    if (rc != CHEST_OK) return rc;
    hipLaunchKernelGGL(ct_gop_kernel, dim3((unsigned)((M2 + 255) / 256), (unsigned)(3 * num_simplices)), dim3(256), 0, st, num_simplices, nsc, P,
                       Mpad, pd.pilot_idx, simplices, neighbors, (const double*)M, Gop);
    hipLaunchKernelGGL(ct_coef_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, G, nsc, num_simplices, pd.pilot_idx, simplices,
                       neighbors, coef);
    hipLaunchKernelGGL(ct_apply_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)G), dim3(256), 0, st, G, P, (const CtCoef*)coef,
                       (const double*)Gop, A);
    prof_close(ctx, st, sp);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}
