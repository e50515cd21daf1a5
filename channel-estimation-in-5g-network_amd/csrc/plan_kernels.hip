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

// simplex -> packed vertices (sym << 16 | sc) + subcarrier range, so that the scan below is one independent 16-B load per simplex
__global__ void __launch_bounds__(256)
plan_pack_kernel(int nsc, const int32_t* __restrict__ pilot_idx, const int32_t* __restrict__ simp, int ns, uint4* __restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    uint32_t v[3];
    int kmin = 0x7fffffff, kmax = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int gi = pilot_idx[simp[(size_t)i * 3 + c]];
        const int s = gi / nsc, k = gi - s * nsc;
        v[c] = ((uint32_t)s << 16) | (uint32_t)k;
        kmin = min(kmin, k); kmax = max(kmax, k);
    }
    packed[i] = make_uint4(v[0], v[1], v[2], (uint32_t)kmin | ((uint32_t)kmax << 16));
}

constexpr int LOC_CAP = 2048;      // compacted simplices kept in LDS per pass (32 KB)

// One wave per 64 consecutive resource elements.  Pass 1: the wave scans all simplices, 64 per step, and keeps those whose
// bounding box meets the block's (symbol, subcarrier) box -- a few per cent of them -- in an LDS list (ballot + prefix
// count, so the list keeps Qhull's order).  Pass 2: every lane tests its element against that short list.  If the list
// overflows (dense patterns) the two passes repeat on the remaining simplices.
__global__ void __launch_bounds__(64)
plan_locate_kernel(int G, int nsc, const uint4* __restrict__ packed, const int32_t* __restrict__ simp, int ns,
                   PlanRec* __restrict__ plan) {
    __shared__ uint4 tri[LOC_CAP];                 // v0, v1, v2 as (sym << 16 | sc), simplex number
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64, g = g0 + lane;
    const bool act = g < G;
    const int ps = act ? g / nsc : 0, pk = act ? g - ps * nsc : 0;
    // the block's box: 64 consecutive elements lie in at most two symbols
    const int gl = min(g0 + 63, G - 1);
    const int s_lo = g0 / nsc, s_hi = gl / nsc;
    const int k_lo = (s_lo == s_hi) ? g0 - s_lo * nsc : 0, k_hi = (s_lo == s_hi) ? gl - s_lo * nsc : nsc - 1;
    bool found = !act;
    PlanRec rec{0.f, 0.f, 0u, 0u};
    int next = 0;                                  // first simplex not yet scanned
    while (next < ns) {
        int cnt = 0;
        // ---- pass 1: compact the candidates of simplices [next, ...) until the list is full
        while (next < ns && cnt + 256 <= LOC_CAP) {
            uint4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                 // four independent 16-B loads in flight
                const int i = next + u * 64 + lane;
                t[u] = i < ns ? packed[i] : make_uint4(0u, 0u, 0u, 0xFFFFu);      // kmin > kmax: never kept
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = next + u * 64 + lane;
                const int kmin = (int)(t[u].w & 0xFFFFu), kmax = (int)(t[u].w >> 16);
                const int a = (int)(t[u].x >> 16), b = (int)(t[u].y >> 16), c = (int)(t[u].z >> 16);
                const bool keep = kmax >= k_lo && kmin <= k_hi && max(a, max(b, c)) >= s_lo && min(a, min(b, c)) <= s_hi;
                const unsigned long long m = __ballot(keep);
                if (keep) tri[cnt + __popcll(m & ((1ull << lane) - 1ull))] = make_uint4(t[u].x, t[u].y, t[u].z, (uint32_t)i);
                cnt += __popcll(m);
            }
            next += 256;
        }
        __syncthreads();
        // ---- pass 2: exact containment against the short list
        if (!found) {
            for (int i = 0; i < cnt; ++i) {
                const uint4 t = tri[i];                                  // same address in every lane: LDS broadcast
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
                const int32_t* sp = simp + (size_t)t.w * 3;
                rec.w0 = (float)n0 / (float)den;
                rec.w1 = (float)n1 / (float)den;
                rec.i01 = (uint32_t)sp[0] | ((uint32_t)sp[1] << 16);
                rec.i2v = (uint32_t)sp[2] | (1u << 16);
                found = true;
                break;
            }
        }
        __syncthreads();
    }
    if (act) plan[g] = rec;
}

int launch_plan_locate(chest_ctx* ctx, hipStream_t st, const int32_t* d_pilot_idx, const int32_t* d_simp, int32_t ns, void* d_packed,
                       void* d_plan) {
    const int G = ctx->fp.G;
    hipLaunchKernelGGL(plan_pack_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, ctx->fp.nsc, d_pilot_idx, d_simp, ns, (uint4*)d_packed);
    hipLaunchKernelGGL(plan_locate_kernel, dim3((G + 63) / 64), dim3(64), 0, st, G, ctx->fp.nsc, (const uint4*)d_packed, d_simp, ns,
                       (PlanRec*)d_plan);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
