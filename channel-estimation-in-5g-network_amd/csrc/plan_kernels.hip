// Device-side construction of the griddata('linear') interpolation plan from a pattern's Delaunay triangulation.
//
// LSEstimator.interpolate_channel (src/baseline_estimators.py:44-81) -> scipy.interpolate.griddata(method='linear') =
// Qhull Delaunay of the pilot positions, then for every resource element: find the simplex that contains it,
// barycentric weights, 0 outside the convex hull.  The triangulation itself stays Qhull's (its choice among co-circular
// lattice points decides the values inside every degenerate quadrilateral, so no other triangulator reproduces the
// reference); what moves to the GPU is everything after it -- point location and weights, 90 % of the host cost.
// Pilots and resource elements are integer lattice points, so containment is decided EXACTLY with integer orientation
// determinants (scipy uses float barycentrics with a 100-eps tolerance; on a lattice a point outside a triangle misses it
// by >= 1/(2*area) >= 1e-6, so the two tests select the same set).  A point on an edge shared by two simplices may be
// attributed to either; the interpolant is continuous there, so the value is the same.  Zero-area simplices are skipped
// (their scipy transform is NaN and find_simplex never returns them).
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

constexpr int LOC_TILE = 1024;     // simplices staged in LDS at a time (16 KB)

__global__ void __launch_bounds__(256)
plan_locate_kernel(int G, int nsc, const int32_t* __restrict__ pilot_idx, const int32_t* __restrict__ simp, int ns,
                   PlanRec* __restrict__ plan) {
    __shared__ uint4 tri[LOC_TILE];                // v0, v1, v2 as (sym << 16 | sc), kmin | kmax << 16
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = g < G;
    const int ps = act ? g / nsc : 0, pk = act ? g - ps * nsc : 0;
    bool found = !act;
    PlanRec rec{0.f, 0.f, 0u, 0u};
    for (int t0 = 0; t0 < ns; t0 += LOC_TILE) {
        const int nt = min(LOC_TILE, ns - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += blockDim.x) {
            uint32_t v[3];
            int kmin = 0x7fffffff, kmax = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int gi = pilot_idx[simp[(size_t)(t0 + i) * 3 + c]];
                const int s = gi / nsc, k = gi - s * nsc;
                v[c] = ((uint32_t)s << 16) | (uint32_t)k;
                kmin = min(kmin, k); kmax = max(kmax, k);
            }
            tri[i] = make_uint4(v[0], v[1], v[2], (uint32_t)kmin | ((uint32_t)kmax << 16));
        }
        __syncthreads();
        if (found) continue;
        for (int i = 0; i < nt; ++i) {
            const uint4 t = tri[i];                                  // same address in every lane: LDS broadcast
            if (pk < (int)(t.w & 0xFFFFu) || pk > (int)(t.w >> 16)) continue;
            const int x0 = (int)(t.x >> 16) - ps, y0 = (int)(t.x & 0xFFFFu) - pk;     // vertices relative to the point
            const int x1 = (int)(t.y >> 16) - ps, y1 = (int)(t.y & 0xFFFFu) - pk;
            const int x2 = (int)(t.z >> 16) - ps, y2 = (int)(t.z & 0xFFFFu) - pk;
            int n0 = x1 * y2 - y1 * x2;                              // 2*area(p, v1, v2)
            int n1 = x2 * y0 - y2 * x0;                              // 2*area(p, v2, v0)
            int n2 = x0 * y1 - y0 * x1;                              // 2*area(p, v0, v1)
            int den = n0 + n1 + n2;                                  // 2*area(v0, v1, v2)
            if (den == 0) continue;
            if (den < 0) { n0 = -n0; n1 = -n1; n2 = -n2; den = -den; }
            if ((n0 | n1 | n2) < 0) continue;                        // some area negative: outside
            const int32_t* sp = simp + (size_t)(t0 + i) * 3;
            rec.w0 = (float)n0 / (float)den;
            rec.w1 = (float)n1 / (float)den;
            rec.i01 = (uint32_t)sp[0] | ((uint32_t)sp[1] << 16);
            rec.i2v = (uint32_t)sp[2] | (1u << 16);
            found = true;
            break;
        }
    }
    if (act) plan[g] = rec;
}

int launch_plan_locate(chest_ctx* ctx, hipStream_t st, const int32_t* d_pilot_idx, const int32_t* d_simp, int32_t ns, void* d_plan) {
    const int G = ctx->fp.G;
    hipLaunchKernelGGL(plan_locate_kernel, dim3((G + 255) / 256), dim3(256), 0, st, G, ctx->fp.nsc, d_pilot_idx, d_simp, ns,
                       (PlanRec*)d_plan);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
