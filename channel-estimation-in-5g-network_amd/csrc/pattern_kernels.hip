// Batched on-device construction of pilot patterns and their interpolation plans: one workgroup per pattern, no host work.
//
//   reference path replaced, once per frame in the reference (src/channel_simulator.py:390-391, src/baseline_estimators.py:65-79):
//     PilotPattern.__init__            src/channel_simulator.py:209-236   shuffle, take P, sort  -> a uniformly random P-subset
//     griddata(method='linear')        scipy: Qhull Delaunay of the pilot positions, then per resource element the containing
//                                      simplex and its barycentric weights, 0 outside the convex hull
//     griddata(method='nearest')       scipy: cKDTree nearest pilot
//
// Per pattern (tools/delaunay_model.py is the CPU model of steps 3-4, used by the tests):
//   1. pilot selection: every resource element gets a Philox key (seed, pattern id, element); the P smallest keys (ties by
//      element index) are the pilots -- the order statistics of a random permutation, i.e. the same distribution as the
//      reference's shuffle-and-take-P.  (Or the caller supplies the sorted pilot list.)
//   2. slot table (draw slot of every resource element) by binary search in the sorted pilot list.
//   3. initial triangulation.  Pilots are lattice points on S parallel rows: each strip between two consecutive occupied
//      rows is triangulated by merging the two sorted rows; the step number of every triangle and of its three neighbours
//      follows from rank arithmetic (two binary searches per triangle, all triangles in parallel).  The pockets between the
//      left / right end chains and the convex hull are filled by one thread (a monotone-chain scan over <= S points a side).
//   4. Lawson flips with the EXACT integer in-circle test (|det| < 2^50 in int64) in rounds of independent flips until every
//      interior edge is locally Delaunay.  The result is a valid Delaunay triangulation; it equals Qhull's wherever the
//      triangulation is unique, i.e. everywhere except inside co-circular lattice polygons, where any choice is Delaunay
//      and the reference's is an artefact of Qhull's facet merging (parity definition: SURVEY 7, hard part 1, option ii).
//   5. plan: every resource element walks (visibility walk, exact integer orientation tests; acyclic on Delaunay
//      triangulations) from its predecessor's triangle to the one containing it and stores the three pilot ordinals and two
//      barycentric weights; elements outside the hull get weight 0 (griddata's fill_value).
//   'nearest': per occupied row a binary search for the closest pilot columns, minimum squared distance, ties to the
//   smaller pilot ordinal (cKDTree's tie choice depends on its internal split order and is not reproduced).
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

#ifndef CHEST_PB_THREADS
#define CHEST_PB_THREADS 256
#endif
constexpr int PB_THREADS = CHEST_PB_THREADS;      // >= 256 (the radix selection keeps its 256 bins on the first 256 threads)
constexpr uint32_t TNONE = 0xFFFFu;
constexpr int PB_MAX_ROWS = CHEST_MAX_SYMBOLS;

struct PatBuildArgs {
    PatternDev* table;            // device pattern table; entries first_slot + j are written by the kernel
    unsigned char* pool;          // count blocks of `stride` bytes: slot u16[G] | pilot_idx i32[Pcap] | plan PlanRec[G]
    size_t stride, o_pil, o_plan;
    const int32_t* num_pilots;    // [count]
    const int32_t* given_idx;     // [count][idx_stride] sorted flat indices, or nullptr: draw them
    const uint64_t* pattern_ids;  // [count]
    uint64_t seed;
    int32_t idx_stride, first_slot, S, nsc, method, Pcap;
    int32_t* simp_out;            // optional [count][2*Pcap][3]
    int32_t* nbr_out;             // optional [count][2*Pcap][3]: triangle across the edge opposite each vertex, -1 on the hull
    int32_t* nsimp_out;           // optional [count]
    int32_t* status;
};

struct alignas(16) TriRec { uint16_t h[6]; uint32_t claim; };
// a whole triangle record in one ds_read_b128 / ds_write_b128
struct TriU { uint32_t v[3], n[3], claim; };
__device__ __forceinline__ TriU ld_tri(const TriRec* TR, uint32_t t) {
    const uint4 r = *(const uint4*)&TR[t];
    return TriU{{r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu}, {r.y >> 16, r.z & 0xFFFFu, r.z >> 16}, r.w};
}
__device__ __forceinline__ void st_tri(TriRec* TR, uint32_t t, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t n0, uint32_t n1, uint32_t n2,
                                       uint32_t claim) {
    *(uint4*)&TR[t] = make_uint4(v0 | (v1 << 16), v2 | (n0 << 16), n1 | (n2 << 16), claim);
}
__device__ __forceinline__ int px_of(uint32_t v) { return (int)(v >> 16); }
__device__ __forceinline__ int py_of(uint32_t v) { return (int)(v & 0xFFFFu); }
__device__ __forceinline__ int orient2(uint32_t a, uint32_t b, uint32_t c) {
    return (px_of(b) - px_of(a)) * (py_of(c) - py_of(a)) - (py_of(b) - py_of(a)) * (px_of(c) - px_of(a));
}
// > 0 iff d lies strictly inside the circumcircle of the counter-clockwise triangle (a, b, c)
// Coordinate differences are < 2^15, so every squared length and every 2x2 minor fits int32 and only the three final
// products need 64 bits (one v_mad_i64_i32 each; a full 64 x 64 multiply costs a dozen quarter-rate instructions).
__device__ __forceinline__ long long incircle(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const int adx = px_of(a) - px_of(d), ady = py_of(a) - py_of(d), bdx = px_of(b) - px_of(d), bdy = py_of(b) - py_of(d),
              cdx = px_of(c) - px_of(d), cdy = py_of(c) - py_of(d);
    const int al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
    const int bc = bdx * cdy - cdx * bdy, ac = adx * cdy - cdx * ady, ab = adx * bdy - bdx * ady;
    return (long long)al * (long long)bc - (long long)bl * (long long)ac + (long long)cl * (long long)ab;
}
// number of entries of the ascending array a[0..n) that are < key / <= key
__device__ __forceinline__ int count_less(const uint32_t* a, int n, uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int count_leq(const uint32_t* a, int n, uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

// exclusive scan of one value per thread over the workgroup (wave shuffles + one LDS hop); `scr` >= 8 ints of LDS;
// returns (exclusive prefix, total)
__device__ __forceinline__ int2 block_exscan(int v, int* scr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) scr[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PB_THREADS / 64; ++w) { const int x = scr[w]; if (w < wave) base += x; tot += x; }
    __syncthreads();
    return make_int2(base + incl - v, tot);
}

size_t pattern_build_lds_bytes(int G, int Pcap) {
    const size_t tri = (size_t)4 * Pcap + 16 + (size_t)2 * Pcap * (sizeof(uint16_t) * 6 + 4 + 2);        // pts | triangle records | partner
    const size_t keys = (size_t)4 * G;
    return (tri > keys ? tri : keys) + (size_t)(PB_THREADS + 8 + 8 * (PB_MAX_ROWS + 2)) * sizeof(int) + 64;
}

#ifdef CHEST_STAMPS
__device__ unsigned long long* g_pat_stamp = nullptr;           // [count][16]: cycles of wave 0 per phase, then detail slots (diagnostic build only)
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                       if (tid == 0 && g_pat_stamp) g_pat_stamp[(size_t)blockIdx.x * 16 + (i)] = now_ - pst_; pst_ = now_; \
                       __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(PB_THREADS)
pattern_build_kernel(const PatBuildArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, j = blockIdx.x;
#ifdef CHEST_STAMPS
    unsigned long long pst_ = __builtin_amdgcn_s_memtime();
#endif
    const int S = A.S, nsc = A.nsc, G = S * nsc;
    const int P = A.num_pilots[j];
    unsigned char* base = A.pool + (size_t)j * A.stride;
    uint16_t* __restrict__ slotp = (uint16_t*)base;
    int32_t* __restrict__ pil = (int32_t*)(base + A.o_pil);
    PlanRec* __restrict__ plan = (PlanRec*)(base + A.o_plan);
    if (tid == 0) {
        PatternDev pd;
        pd.slot = slotp; pd.plan = A.method >= 0 ? (const void*)plan : nullptr; pd.pilot_idx = pil; pd.P = P; pd.pad = 0;
        A.table[A.first_slot + j] = pd;
        if (A.nsimp_out) A.nsimp_out[j] = 0;
    }
    if (P < 1 || P > A.Pcap || P > G) { if (tid == 0) atomicExch(A.status, 1); return; }

    // ---- LDS carve-up: the key array of step 1 shares its space with the triangle arrays of steps 3-5 ----------------
    const int NT = 2 * A.Pcap;
    int* misc = (int*)lds_raw;                                   // [PB_THREADS + 8 + 4 * (rows + 2)]
    int* scan = misc;                                            // PB_THREADS + 1
    int* cnt_s = misc + PB_THREADS + 1;                          // a few scalars
    int* rowstart = misc + PB_THREADS + 8;                       // [rows + 1] first pilot of every row
    int* occ = rowstart + PB_MAX_ROWS + 2;                       // occupied rows
    int* toff = occ + PB_MAX_ROWS + 2;                           // first triangle of every strip
    int* stack = toff + PB_MAX_ROWS + 2;                         // pocket scan (thread 0): chain stack,
    uint32_t* eown = (uint32_t*)(stack + PB_MAX_ROWS + 2);       //   owner of the edge (stack[i-1], stack[i]): tri << 2 | slot, or ~0u,
    int* eq = (int*)(eown + PB_MAX_ROWS + 2);                    //   strip of that edge if it is the bare chain edge of a strip without triangles,
    uint32_t* rung = (uint32_t*)(eq + PB_MAX_ROWS + 2);          //   and who owns such an edge after the left pass
    // (byte offsets from the LDS base, never pointer -> integer -> pointer: such a round trip loses the LDS address space and
    // every access below becomes a FLAT instruction)
    const unsigned off_big = ((unsigned)(PB_THREADS + 8 + 8 * (PB_MAX_ROWS + 2)) * 4u + 15u) & ~15u;
    uint32_t* keys = (uint32_t*)(lds_raw + off_big);
    uint32_t* pts = (uint32_t*)(lds_raw + off_big);
    // one 16-B record per triangle: vertices h[0..3), neighbours across the edge opposite each vertex h[3..6), claim word
    const unsigned off_tr = (off_big + 4u * (unsigned)A.Pcap + 15u) & ~15u;
    TriRec* TR = (TriRec*)(lds_raw + off_tr);
    uint16_t* partner = (uint16_t*)(lds_raw + off_tr + (unsigned)NT * (unsigned)sizeof(TriRec));

    // ================= 1. pilots =================================================================================
    const int per = (G + PB_THREADS - 1) / PB_THREADS;           // contiguous element block of this thread
    const int g_lo = min(G, tid * per), g_hi = min(G, g_lo + per);
    if (A.given_idx != nullptr) {
        for (int i = tid; i < P; i += PB_THREADS) pil[i] = A.given_idx[(size_t)j * A.idx_stride + i];
    } else {
        const uint64_t pid = A.pattern_ids[j];
        const RngKey key{(uint32_t)A.seed, (uint32_t)(A.seed >> 32), (uint32_t)pid, (uint32_t)(pid >> 32)};
        for (int q = tid; q < (G + 3) / 4; q += PB_THREADS) {                 // one Philox call keys four elements
            const u32x4 w = draw(key, STREAM_PATTERN, (uint32_t)q);
            const int g = 4 * q;
            keys[g] = w.x;
            if (g + 1 < G) keys[g + 1] = w.y;
            if (g + 2 < G) keys[g + 2] = w.z;
            if (g + 3 < G) keys[g + 3] = w.w;
        }
        __syncthreads();
        // tau = the P-th smallest key, by radix selection: four passes over 8-bit digits, most significant first; each pass
        // histograms the digit of the keys that match the prefix found so far (LDS atomics), scans the 256 bins and keeps
        // the bin in which the P-th key falls
        uint32_t prefix = 0;
        int need = P;                                                         // rank of tau among the keys matching the prefix
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) scan[tid] = 0;                                     // 256 bins
            __syncthreads();
            for (int g = g_lo; g < g_hi; ++g) {
                const uint32_t kq = keys[g];
                if (shift == 24 || (kq >> (shift + 8)) == prefix) atomicAdd(&scan[(kq >> shift) & 255u], 1);
            }
            __syncthreads();
            const int h = tid < 256 ? scan[tid] : 0;
            __syncthreads();
            const int2 ex = block_exscan(h, scan);                            // keys in lower bins
            if (ex.x < need && need <= ex.x + h) { cnt_s[0] = tid; cnt_s[4] = need - ex.x; }     // exactly one bin qualifies
            __syncthreads();
            prefix = (prefix << 8) | (uint32_t)cnt_s[0];
            need = cnt_s[4];
            __syncthreads();
        }
        const uint32_t tau = prefix;
        int nl = 0, ne = 0;
        for (int g = g_lo; g < g_hi; ++g) { nl += keys[g] < tau ? 1 : 0; ne += keys[g] == tau ? 1 : 0; }
        const int2 sl = block_exscan(nl, scan);
        const int2 se = block_exscan(ne, scan);
        const int need_eq = P - sl.y;                                         // keys equal to tau to take, in element order
        int pos = sl.x + min(se.x, need_eq), eq_seen = se.x;
        for (int g = g_lo; g < g_hi; ++g) {
            const uint32_t kq = keys[g];
            bool take = kq < tau;
            if (kq == tau) { take = eq_seen < need_eq; ++eq_seen; }
            if (take) pil[pos++] = g;
        }
        __threadfence_block();
    }
    __syncthreads();                                             // pil complete (global, read back by this workgroup only)
    for (int i = tid; i < P; i += PB_THREADS) {
        const int g = pil[i];
        const int s = g / nsc;
        pts[i] = ((uint32_t)s << 16) | (uint32_t)(g - s * nsc);
    }
    __syncthreads();

    PSTAMP(0);                                                   // pilot selection
    // ================= 2. slot table ==============================================================================
    {
        int g = g_lo;
        int s = g / nsc, k = g - s * nsc;
        int pos = g < g_hi ? count_less(pts, P, ((uint32_t)s << 16) | (uint32_t)k) : 0;      // pilots before the block
        uint32_t nextp = pos < P ? pts[pos] : 0xFFFFFFFFu;
        for (; g < g_hi; ++g) {
            const uint32_t key = ((uint32_t)s << 16) | (uint32_t)k;
            if (key == nextp) { slotp[g] = (uint16_t)pos; ++pos; nextp = pos < P ? pts[pos] : 0xFFFFFFFFu; }
            else slotp[g] = (uint16_t)(P + (g - pos));
            if (++k == nsc) { k = 0; ++s; }
        }
    }
    PSTAMP(1);                                                   // slot table
    if (A.method < 0) return;                                    // simulation-only pattern: no plan

    // per-row tables
    for (int r = tid; r <= S; r += PB_THREADS) rowstart[r] = count_less(pts, P, (uint32_t)r << 16);
    __syncthreads();
    if (tid == 0) {
        int m = 0, t = 0;
        for (int r = 0; r < S; ++r) if (rowstart[r + 1] > rowstart[r]) occ[m++] = r;
        for (int q = 0; q + 1 < m; ++q) {
            toff[q] = t;
            t += (rowstart[occ[q] + 1] - rowstart[occ[q]]) + (rowstart[occ[q + 1] + 1] - rowstart[occ[q + 1]]) - 2;
        }
        toff[m > 0 ? m - 1 : 0] = t;
        cnt_s[1] = m; cnt_s[2] = t;                              // occupied rows, strip triangles
    }
    __syncthreads();
    const int m = cnt_s[1];

    if (A.method == 1) {
        // ================= 'nearest' plan ========================================================================
        int g = g_lo;
        int s = g / nsc, k = g - s * nsc;
        for (; g < g_hi; ++g) {
            int best = 0x7fffffff, bi = 0;
            for (int q = 0; q < m; ++q) {
                const int r = occ[q], a = rowstart[r], n = rowstart[r + 1] - a;
                const int c = count_less(pts + a, n, ((uint32_t)r << 16) | (uint32_t)k);      // first column >= k
                const int dr = (r - s) * (r - s);
                if (c > 0) { const int dc = k - py_of(pts[a + c - 1]); const int d2 = dr + dc * dc; if (d2 < best) { best = d2; bi = a + c - 1; } }
                if (c < n) { const int dc = py_of(pts[a + c]) - k; const int d2 = dr + dc * dc; if (d2 < best) { best = d2; bi = a + c; } }
            }
            plan[g] = PlanRec{1.0f, 0.0f, (uint32_t)bi | ((uint32_t)bi << 16), (uint32_t)bi | (1u << 16)};
            if (++k == nsc) { k = 0; ++s; }
        }
        return;
    }

    // ================= 3. initial triangulation ====================================================================
    PSTAMP(2);                                                   // row tables
    const int nstrip = cnt_s[2];
    // every pilot emits the triangle that advances from it to its right-hand neighbour in its row: once as a point of the
    // upper row of the strip below it (A-step) and once as a point of the lower row of the strip above it (B-step)
    for (int p = tid; p < P; p += PB_THREADS) {
        const int r = px_of(pts[p]);
        int q = 0;
        while (occ[q] != r) ++q;                                 // occupied-row index (<= S steps)
        const int a0 = rowstart[r], na = rowstart[r + 1] - a0, i = p - a0;
        if (i + 1 >= na) continue;                               // last point of its row: no step starts here
        const uint32_t nextcol = (uint32_t)py_of(pts[p + 1]);
        if (q + 1 < m) {
            // A-step of strip q: triangle (a_i, b_j, a_{i+1}), j = B-steps taken before = #{j' >= 1 : colB[j'] < col(a_{i+1})}
            const int rb = occ[q + 1], b0 = rowstart[rb], nb = rowstart[rb + 1] - b0;
            const uint32_t kb = ((uint32_t)rb << 16) | nextcol;
            const int jb = count_less(pts + b0, nb, kb) - ((pts[b0] < kb) ? 1 : 0);
            const int k = i + jb, t = toff[q] + k, nq = na + nb - 2;
            TR[t].h[0] = (uint16_t)p; TR[t].h[1] = (uint16_t)(b0 + jb); TR[t].h[2] = (uint16_t)(p + 1);
            TR[t].h[3] = (uint16_t)(k + 1 < nq ? t + 1 : TNONE);     // rung after the step
            TR[t].h[5] = (uint16_t)(k > 0 ? t - 1 : TNONE);          // rung before the step
            // across the row edge: the B-step of the strip above that advances over the same edge
            uint32_t up = TNONE;
            if (q > 0) {
                const int ru = occ[q - 1], u0 = rowstart[ru], nu = rowstart[ru + 1] - u0;
                const uint32_t ku = ((uint32_t)ru << 16) | nextcol;
                const int ia = count_leq(pts + u0, nu, ku) - ((pts[u0] <= ku) ? 1 : 0);     // A-steps before it (ties: upper row first)
                up = (uint32_t)(toff[q - 1] + i + ia);
            }
            TR[t].h[4] = (uint16_t)up;
        }
        if (q > 0) {
            // B-step of strip q-1: triangle (b_j, b_{j+1}, a_i) with j = i here, i_A = #{i' >= 1 : colA[i'] <= col(b_{j+1})}
            const int ru = occ[q - 1], u0 = rowstart[ru], nu = rowstart[ru + 1] - u0;
            const uint32_t ku = ((uint32_t)ru << 16) | nextcol;
            const int ia = count_leq(pts + u0, nu, ku) - ((pts[u0] <= ku) ? 1 : 0);
            const int k = i + ia, t = toff[q - 1] + k, nq = nu + na - 2;
            TR[t].h[0] = (uint16_t)p; TR[t].h[1] = (uint16_t)(p + 1); TR[t].h[2] = (uint16_t)(u0 + ia);
            TR[t].h[3] = (uint16_t)(k + 1 < nq ? t + 1 : TNONE);     // rung after
            TR[t].h[4] = (uint16_t)(k > 0 ? t - 1 : TNONE);          // rung before
            uint32_t dn = TNONE;
            if (q + 1 < m) {
                const int rb = occ[q + 1], b0 = rowstart[rb], nb = rowstart[rb + 1] - b0;
                const uint32_t kb = ((uint32_t)rb << 16) | nextcol;
                const int jb = count_less(pts + b0, nb, kb) - ((pts[b0] < kb) ? 1 : 0);
                dn = (uint32_t)(toff[q] + i + jb);
            }
            TR[t].h[5] = (uint16_t)dn;
        }
    }
    __syncthreads();
    PSTAMP(3);                                                   // strips
    // pockets between the end chains and the convex hull (one thread; <= 2 (m - 2) triangles).  A monotone-chain scan over the
    // first (left) / last (right) pilot of every occupied row; every boundary edge of the scan carries its owner (triangle,
    // slot), so each pocket triangle is linked to its neighbours in O(1).
    if (tid == 0) {
        int nt = nstrip;
        auto link = [&](uint32_t own, int t, int c) {            // make `own` and slot c of triangle t neighbours
            if (own == 0xFFFFFFFFu) return;
            TR[t].h[3 + (c)] = (uint16_t)(own >> 2);
            TR[own >> 2].h[3 + ((int)(own & 3u))] = (uint16_t)t;
        };
        auto slot_of_edge = [&](int t, uint32_t x, uint32_t y) -> int {     // slot of t whose edge joins x and y
            for (int d = 0; d < 3; ++d) {
                const uint32_t e0 = TR[t].h[(d + 1) % 3], e1 = TR[t].h[(d + 2) % 3];
                if ((e0 == x && e1 == y) || (e0 == y && e1 == x)) return d;
            }
            return 0;
        };
        for (int q = 0; q < m; ++q) rung[q] = 0xFFFFFFFFu;
        for (int side = 0; side < 2; ++side) {
            int sp = 0;
            for (int q = 0; q < m; ++q) {
                const int r = occ[q];
                const int p = side == 0 ? rowstart[r] : rowstart[r + 1] - 1;
                // owner of the chain edge (previous chain point, p): the first / last triangle of strip q - 1
                uint32_t cur = 0xFFFFFFFFu;
                if (q > 0) {
                    const int nq = toff[q] - toff[q - 1];
                    if (nq > 0) {
                        const int te = side == 0 ? toff[q - 1] : toff[q - 1] + nq - 1;
                        cur = ((uint32_t)te << 2) | (uint32_t)slot_of_edge(te, (uint32_t)stack[sp - 1], (uint32_t)p);
                    } else cur = rung[q - 1];                    // both rows hold one pilot: the edge belongs to both chains
                }
                const bool empty_strip = q > 0 && toff[q] - toff[q - 1] <= 0;
                int cur_q = empty_strip ? q - 1 : -1;            // cur is still the bare chain edge of that (triangle-less) strip
                while (sp >= 2) {
                    const int a = stack[sp - 2], b = stack[sp - 1];
                    const int o = orient2(pts[a], pts[b], pts[p]);
                    // b lies in a pocket if it is on the inner side of a -> p (left chain: o < 0, right chain: o > 0)
                    if (!(side == 0 ? o < 0 : o > 0)) break;
                    // counter-clockwise: (a, p, b) on the left chain, (a, b, p) on the right one
                    const int t = nt++;
                    TR[t].h[0] = (uint16_t)a; TR[t].h[1] = (uint16_t)(side == 0 ? p : b); TR[t].h[2] = (uint16_t)(side == 0 ? b : p);
                    TR[t].h[3] = TR[t].h[4] = TR[t].h[5] = (uint16_t)TNONE;
                    const int s_ab = side == 0 ? 1 : 2;          // edge (a, b) is opposite p
                    const int s_bp = 0;                          // edge (b, p) is opposite a
                    const int s_ap = side == 0 ? 2 : 1;          // edge (a, p) is opposite b
                    link(eown[sp - 1], t, s_ab);
                    link(cur, t, s_bp);
                    // a chain edge shared by both chains (strip without triangles): remember who took it on the left pass
                    if (side == 0 && eq[sp - 1] >= 0) rung[eq[sp - 1]] = ((uint32_t)t << 2) | (uint32_t)s_ab;
                    if (side == 0 && cur_q >= 0) rung[cur_q] = ((uint32_t)t << 2) | (uint32_t)s_bp;
                    cur_q = -1;
                    cur = ((uint32_t)t << 2) | (uint32_t)s_ap;
                    --sp;
                }
                eown[sp] = cur;
                eq[sp] = cur_q;
                stack[sp++] = p;
            }
        }
        cnt_s[2] = nt;
    }
    __syncthreads();
    const int ntri = cnt_s[2];
    PSTAMP(4);                                                   // pockets

    // ================= 4. Lawson flips ==============================================================================
    // Rounds of independent flips.  A triangle is ACTIVE (its claim word starts the round as CLAIM_DIRTY) if it changed or lost a
    // claim in the previous round; everything is active at first.  Only active triangles test their edges -- an edge between
    // two active triangles from the smaller index -- so once the bulk of the flips is done a round costs one cheap pass.
    // Every thread keeps the active set of its own triangles (t = tid + 256 i) as a register bit mask.
    constexpr uint32_t CLAIM_NONE = 0xFFFFFFFFu, CLAIM_DIRTY = 0xFFFFFFFEu;
    uint32_t active = 0;
    for (int i = 0, t = tid; t < ntri; t += PB_THREADS, ++i) { TR[t].claim = CLAIM_DIRTY; partner[t] = (uint16_t)TNONE; active |= 1u << i; }
    __syncthreads();
#ifdef CHEST_STAMPS
    unsigned dbg_rounds = 0, dbg_cross = 0;
    unsigned long long dbg_ph[4] = {0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#define DBG_PH(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg_ph[i] += n_ - dbg_t; dbg_t = n_; } while (0)
#else
#define DBG_PH(i) do { } while (0)
#endif
    for (int round = 0; round < 4 * NT + 64; ++round) {
#ifdef CHEST_STAMPS
        ++dbg_rounds;
#endif
        if (tid == 0) cnt_s[3] = 0;
        __syncthreads();
        // phase 0: active triangles test their edges; an edge that fails the in-circle test claims both of its triangles.
        // One 16-B read brings a triangle; the three neighbours and then the three opposite pilots are fetched together.
        uint32_t prop = 0;                                       // my triangles that proposed a flip this round
        for (uint32_t m = active; m != 0; m &= m - 1) {
            const int bit = __ffs(m) - 1;
            const int t = tid + PB_THREADS * bit;
            const TriU T0 = ld_tri(TR, (uint32_t)t);
            const uint32_t pv[3] = {pts[T0.v[0]], pts[T0.v[1]], pts[T0.v[2]]};
            TriU U[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) U[i] = ld_tri(TR, T0.n[i] == TNONE ? (uint32_t)t : T0.n[i]);
            uint32_t pd[3];
            bool need[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t u = T0.n[i];
                need[i] = !(u == TNONE || (u < (uint32_t)t && U[i].claim != CLAIM_NONE));       // else the other (active) side tests it
                const uint32_t b = T0.v[(i + 1) % 3], c = T0.v[(i + 2) % 3];
                const uint32_t d = (U[i].v[0] != b && U[i].v[0] != c) ? U[i].v[0] : ((U[i].v[1] != b && U[i].v[1] != c) ? U[i].v[1] : U[i].v[2]);
                pd[i] = pts[need[i] ? d : T0.v[i]];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (need[i] && incircle(pv[i], pv[(i + 1) % 3], pv[(i + 2) % 3], pd[i]) > 0) {
                    const uint32_t key = (uint32_t)t * 4u + (uint32_t)i;
                    atomicMin(&TR[t].claim, key);
                    atomicMin(&TR[T0.n[i]].claim, key);
                    cnt_s[3] = 1;
                    prop |= 1u << bit;
                }
            }
        }
        __syncthreads();
        DBG_PH(0);
        if (cnt_s[3] == 0) break;
        // phase 1: winners (an edge that holds the claim on both of its triangles) rewrite the two triangles; outer neighbours
        // keep their OLD ids for now.  Only triangles that proposed can own a winning edge.
        uint32_t won = 0;
        for (uint32_t m = prop; m != 0; m &= m - 1) {
            const int bit = __ffs(m) - 1;
            const int t = tid + PB_THREADS * bit;
            const TriU T0 = ld_tri(TR, (uint32_t)t);
            const uint32_t key = T0.claim;
            if ((key >> 2) != (uint32_t)t) continue;             // the winning edge of t was tested from another triangle
            const int i = (int)(key & 3u);
            const uint32_t u = T0.n[i];
            const TriU U0 = ld_tri(TR, u);
            if (U0.claim != key) continue;
            const uint32_t a = T0.v[i], b = T0.v[(i + 1) % 3], c = T0.v[(i + 2) % 3];
            const uint32_t n_ab = T0.n[(i + 2) % 3], n_ca = T0.n[(i + 1) % 3];
            const int jd = (U0.v[0] != b && U0.v[0] != c) ? 0 : ((U0.v[1] != b && U0.v[1] != c) ? 1 : 2);
            const uint32_t d = U0.v[jd];
            // u = (d, c, b) rotated so that u[jd] = d: the edge opposite c is (b, d), the edge opposite b is (d, c)
            const uint32_t n_db = U0.n[(jd + 1) % 3], n_cd = U0.n[(jd + 2) % 3];
            st_tri(TR, (uint32_t)t, a, b, d, n_db, u, n_ab, key);
            st_tri(TR, u, a, d, c, n_cd, n_ca, (uint32_t)t, key);
            partner[t] = (uint16_t)u; partner[u] = (uint16_t)t;
            won |= 1u << bit;
        }
        __syncthreads();
        DBG_PH(1);
        // phase 2: the owner of every flip resolves the outer neighbours of its two new triangles (the old neighbour, or its
        // partner if that one flipped too) and writes the back-pointer of neighbours that did not flip
        for (uint32_t m = won; m != 0; m &= m - 1) {
            const int t = tid + PB_THREADS * (__ffs(m) - 1);
            const uint32_t pT = partner[t];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const uint32_t X = side == 0 ? (uint32_t)t : pT;
                const uint32_t pX = side == 0 ? pT : (uint32_t)t;
                const TriU TX = ld_tri(TR, X);
                TriU TY[3];
                uint32_t pY[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t Y = (TX.n[k] == TNONE || TX.n[k] == pX) ? X : TX.n[k];
                    TY[k] = ld_tri(TR, Y);
                    pY[k] = partner[Y];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    uint32_t Y = TX.n[k];
                    if (Y == TNONE || Y == pX) continue;
                    const uint32_t x = TX.v[(k + 1) % 3], y = TX.v[(k + 2) % 3];
                    const uint32_t y0 = TY[k].v[0], y1 = TY[k].v[1], y2 = TY[k].v[2];
                    if (pY[k] != TNONE) {
                        const bool has = (y0 == x || y1 == x || y2 == x) && (y0 == y || y1 == y || y2 == y);
                        if (!has) Y = pY[k];
                        TR[X].h[3 + k] = (uint16_t)Y;
                    } else {
                        const int k2 = (y0 != x && y0 != y) ? 0 : ((y1 != x && y1 != y) ? 1 : 2);
                        TR[Y].h[3 + k2] = (uint16_t)X;
                    }
                }
            }
        }
        __syncthreads();
        DBG_PH(2);
        // claimed triangles (flipped, or candidates that lost) are the next round's active set; marks that drew no claim clear
        active = 0;
        for (int i = 0, t = tid; t < ntri; t += PB_THREADS, ++i) {
            const uint32_t c = TR[t].claim;
            if (c == CLAIM_NONE) continue;
            const bool act = c < CLAIM_DIRTY;
            TR[t].claim = act ? CLAIM_DIRTY : CLAIM_NONE;
            partner[t] = (uint16_t)TNONE;
            if (act) active |= 1u << i;
        }
        __syncthreads();
        DBG_PH(3);
    }

    PSTAMP(5);                                                   // flips
    if (A.simp_out != nullptr) {
        int32_t* so = A.simp_out + (size_t)j * 2 * A.Pcap * 3;
        for (int t = tid; t < ntri; t += PB_THREADS) { so[3 * t] = TR[t].h[0]; so[3 * t + 1] = TR[t].h[1]; so[3 * t + 2] = TR[t].h[2]; }
        if (A.nbr_out != nullptr) {
            int32_t* no = A.nbr_out + (size_t)j * 2 * A.Pcap * 3;
            for (int t = tid; t < ntri; t += PB_THREADS) {
                no[3 * t] = TR[t].h[3] == TNONE ? -1 : (int)TR[t].h[3]; no[3 * t + 1] = TR[t].h[4] == TNONE ? -1 : (int)TR[t].h[4];
                no[3 * t + 2] = TR[t].h[5] == TNONE ? -1 : (int)TR[t].h[5];
            }
        }
        if (tid == 0 && A.nsimp_out) A.nsimp_out[j] = ntri;
    }

    // ================= 5. plan: walk to the containing triangle =====================================================
    // Every lane owns a run of consecutive elements and runs ONE loop in which an iteration either crosses into the
    // neighbouring triangle or emits the current element and moves one column on (lanes do not wait for each other per
    // element).  While the triangle stays the same the three edge functions are updated incrementally (they are affine in
    // the column); they are recomputed only after a crossing.
    if (ntri > 0) {
        // Start of a walk: a triangle incident to the pilot nearest (in element order: the one before or the one after, whichever
        // is closer in the plane) to the element.  After the flips triangle numbers say nothing about position, and a walk
        // from a far triangle crosses hundreds of them along the 599-column axis (stamps: 407 crossings for 33 elements).
        // (the smallest incident triangle number, so that a rebuilt pattern gives the bit-identical plan: an element ON an
        // edge belongs to whichever of the two triangles the walk reaches first)
        uint32_t* __restrict__ v2t = (uint32_t*)partner;         // partner[] (2 * Pcap uint16) is free after the flips
        for (int i = tid; i < P; i += PB_THREADS) v2t[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (int t = tid; t < ntri; t += PB_THREADS) { atomicMin(&v2t[TR[t].h[0]], (uint32_t)t); atomicMin(&v2t[TR[t].h[1]], (uint32_t)t); atomicMin(&v2t[TR[t].h[2]], (uint32_t)t); }
        __syncthreads();
        auto near_tri = [&](int s_, int k_) -> int {
            const uint32_t q = ((uint32_t)s_ << 16) | (uint32_t)k_;
            int ip = count_less(pts, P, q);                      // first pilot at or after the element
            if (ip >= P) ip = P - 1;
            if (ip > 0) {
                const uint32_t a = pts[ip - 1], b = pts[ip];
                const int da = (s_ - px_of(a)) * (s_ - px_of(a)) + (k_ - py_of(a)) * (k_ - py_of(a));
                const int db = (s_ - px_of(b)) * (s_ - px_of(b)) + (k_ - py_of(b)) * (k_ - py_of(b));
                if (da < db) --ip;
            }
            return (int)v2t[ip];
        };
        int g = g_lo;
        int s = g / nsc, k = g - s * nsc;
        int t = g < g_hi ? near_tri(s, k) : 0;
        uint32_t i0 = 0, i1 = 0, i2 = 0;
        int o0 = 0, o1 = 0, o2 = 0, d0 = 0, d1 = 0, d2 = 0;
        bool fresh = true;                                        // triangle t not loaded yet
        int since = 0;                                            // crossings since the last emitted element
        while (g < g_hi) {
            if (fresh) {
                i0 = TR[t].h[0]; i1 = TR[t].h[1]; i2 = TR[t].h[2];
                const uint32_t a = pts[i0], b = pts[i1], c = pts[i2], q = ((uint32_t)s << 16) | (uint32_t)k;
                o0 = orient2(b, c, q); o1 = orient2(c, a, q); o2 = orient2(a, b, q);
                d0 = px_of(c) - px_of(b); d1 = px_of(a) - px_of(c); d2 = px_of(b) - px_of(a);   // d(orient(u, v, q)) / d(column of q) = v.x - u.x
                fresh = false;
            }
            uint32_t nx = TNONE;
            bool cross = false;
            if (o0 < 0) { nx = TR[t].h[3]; cross = true; }
            else if (o1 < 0) { nx = TR[t].h[4]; cross = true; }
            else if (o2 < 0) { nx = TR[t].h[5]; cross = true; }
#ifdef CHEST_STAMPS
            if (cross && nx != TNONE) ++dbg_cross;
#endif
            if (cross && nx != TNONE && since <= ntri + 8) { t = (int)nx; fresh = true; ++since; continue; }
            PlanRec rec{0.f, 0.f, 0u, 0u};                        // leaving through a hull edge: outside the convex hull
            if (!cross) {
                const float den = (float)(o0 + o1 + o2);
                rec.w0 = (float)o0 / den; rec.w1 = (float)o1 / den;
                rec.i01 = i0 | (i1 << 16); rec.i2v = i2 | (1u << 16);
            } else if (nx != TNONE) {
                // not reached on a Delaunay triangulation (the visibility walk is acyclic); exhaustive scan as a safety net
                const uint32_t q = ((uint32_t)s << 16) | (uint32_t)k;
                for (int u = 0; u < ntri; ++u) {
                    const uint32_t j0 = TR[u].h[0], j1 = TR[u].h[1], j2 = TR[u].h[2];
                    const int p0 = orient2(pts[j1], pts[j2], q), p1 = orient2(pts[j2], pts[j0], q), p2 = orient2(pts[j0], pts[j1], q);
                    if ((p0 | p1 | p2) >= 0) {
                        const float den = (float)(p0 + p1 + p2);
                        rec.w0 = (float)p0 / den; rec.w1 = (float)p1 / den;
                        rec.i01 = j0 | (j1 << 16); rec.i2v = j2 | (1u << 16);
                        break;
                    }
                }
            }
            plan[g] = rec;
            ++g; since = 0;
            if (++k == nsc) {                                     // next row: far from here, start again from the layout guess
                k = 0; ++s;
                if (g < g_hi) t = near_tri(s, k);
                fresh = true;
            } else { o0 += d0; o1 += d1; o2 += d2; }
        }
    } else {
        for (int g = g_lo; g < g_hi; ++g) plan[g] = PlanRec{0.f, 0.f, 0u, 0u};
    }
    PSTAMP(6);                                                   // plan walk
#ifdef CHEST_STAMPS
    if (g_pat_stamp) {                                           // slot 7: flip rounds | largest crossing count of a lane of wave 0 << 32
        for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(dbg_cross, off, 64); dbg_cross = o > dbg_cross ? o : dbg_cross; }
        if (tid == 0) {
            unsigned long long* d_ = g_pat_stamp + (size_t)blockIdx.x * 16;
            d_[7] = (unsigned long long)dbg_rounds | ((unsigned long long)dbg_cross << 32);
            for (int i_ = 0; i_ < 4; ++i_) d_[8 + i_] = dbg_ph[i_];
        }
    }
#endif
}

}  // namespace chest

#ifdef CHEST_STAMPS
extern "C" int chest_debug_set_pattern_stamp_buffer(void* dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(chest::g_pat_stamp), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif

using namespace chest;

extern "C" int chest_patterns_build(chest_ctx* ctx, void* stream, int32_t first_slot, int32_t count, const int32_t* num_pilots,
                                    const int32_t* pilot_idx, int32_t idx_stride, uint64_t seed, const uint64_t* pattern_ids,
                                    int32_t method, int32_t* simplices_out, int32_t* num_simplices_out, int32_t* neighbors_out) {
    if (!ctx || first_slot < 0 || count < 0 || first_slot + (int64_t)count > (1 << 24) || (count > 0 && !num_pilots)) {
        set_error("chest_patterns_build: bad argument");
        return CHEST_EINVAL;
    }
    if (count == 0) return CHEST_OK;
    if (method < -1 || method > 1) { set_error("chest_patterns_build: method must be -1 (no plan), 0 ('linear') or 1 ('nearest')"); return CHEST_EINVAL; }
    if (pilot_idx == nullptr && pattern_ids == nullptr) { set_error("chest_patterns_build: give pilot_idx or pattern_ids"); return CHEST_EINVAL; }
    CHEST_GUARD(ctx);
    hipStream_t st = (hipStream_t)stream;
    const int G = ctx->fp.G;
    int Pmax = 1;
    for (int i = 0; i < count; ++i) {
        if (num_pilots[i] < 1 || num_pilots[i] > G || (pilot_idx && num_pilots[i] > idx_stride)) {
            set_error("chest_patterns_build: num_pilots[%d] = %d out of range", i, num_pilots[i]);
            return CHEST_EINVAL;
        }
        Pmax = num_pilots[i] > Pmax ? num_pilots[i] : Pmax;
    }
    if (2 * Pmax > 32 * PB_THREADS) { set_error("chest_patterns_build: at most %d pilots per pattern", 16 * PB_THREADS); return CHEST_EINVAL; }
    const size_t lds = pattern_build_lds_bytes(G, Pmax);
    if (lds > 160 * 1024) {
        set_error("chest_patterns_build: %d pilots need %zu B of LDS (> 160 KiB); use chest_pattern_set_triangulation for such patterns", Pmax, lds);
        return CHEST_EINVAL;
    }
    auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    // pool reuse: the slot range is served by one pool whose blocks are large enough -> build in place (stream ordered)
    PatternPool* pool = nullptr;
    for (auto& pl : ctx->pools)
        if (pl.first_slot == first_slot && pl.count >= count && pl.Pcap >= Pmax) pool = &pl;
    if (pool == nullptr) {
        CHEST_HIP(hipDeviceSynchronize());
        // release whatever owned these slots before
        for (auto it = ctx->pools.begin(); it != ctx->pools.end();) {
            const bool overlap = it->first_slot < first_slot + count && first_slot < it->first_slot + it->count;
            if (overlap) {
                for (int s = it->first_slot; s < it->first_slot + it->count && s < (int)ctx->patterns.size(); ++s) ctx->patterns[s] = PatternHost{};
                (void)hipFree(it->block); (void)hipFree(it->d_meta);
                it = ctx->pools.erase(it);
            } else ++it;
        }
        if ((size_t)(first_slot + count) > ctx->patterns.size()) ctx->patterns.resize(first_slot + count);
        for (int s = first_slot; s < first_slot + count; ++s)
            if (ctx->patterns[s].block) { (void)hipFree(ctx->patterns[s].block); ctx->patterns[s] = PatternHost{}; }
        PatternPool pl{};
        pl.first_slot = first_slot; pl.count = count; pl.Pcap = Pmax;
        pl.o_pil = up((size_t)G * 2, 16);
        pl.o_plan = up(pl.o_pil + (size_t)Pmax * 4, 256);
        pl.stride = up(pl.o_plan + (size_t)G * sizeof(PlanRec), 256);
        if (hipMalloc((void**)&pl.block, pl.stride * (size_t)count) != hipSuccess) { set_error("chest_patterns_build: out of device memory (%zu B)", pl.stride * (size_t)count); return CHEST_ENOMEM; }
        if (hipMalloc((void**)&pl.d_meta, up((size_t)count * sizeof(int32_t), 8) + (size_t)count * sizeof(uint64_t)) != hipSuccess) { (void)hipFree(pl.block); set_error("out of device memory"); return CHEST_ENOMEM; }
        ctx->pools.push_back(pl);
        pool = &ctx->pools.back();
        // make sure the device table covers the range
        const int last = first_slot + count - 1;
        if (last >= ctx->pattern_cap) {
            int cap = ctx->pattern_cap ? ctx->pattern_cap : 16;
            while (cap <= last) cap *= 2;
            PatternDev* nd = nullptr;
            CHEST_HIP(hipMalloc((void**)&nd, (size_t)cap * sizeof(PatternDev)));
            CHEST_HIP(hipMemset(nd, 0, (size_t)cap * sizeof(PatternDev)));
            if (ctx->d_patterns && ctx->pattern_cap) CHEST_HIP(hipMemcpy(nd, ctx->d_patterns, (size_t)ctx->pattern_cap * sizeof(PatternDev), hipMemcpyDeviceToDevice));
            CHEST_HIP(hipDeviceSynchronize());
            if (ctx->d_patterns) CHEST_HIP(hipFree(ctx->d_patterns));
            ctx->d_patterns = nd;
            ctx->pattern_cap = cap;
        }
    }
    // per-pattern pilot counts and ids to the device (small; ordered on the launch stream)
    int32_t* d_np = (int32_t*)pool->d_meta;
    uint64_t* d_ids = (uint64_t*)((unsigned char*)pool->d_meta + up((size_t)pool->count * sizeof(int32_t), 8));
    CHEST_HIP(hipMemcpyAsync(d_np, num_pilots, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (pattern_ids) CHEST_HIP(hipMemcpyAsync(d_ids, pattern_ids, (size_t)count * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    PatBuildArgs a{};
    a.table = ctx->d_patterns; a.pool = (unsigned char*)pool->block; a.stride = pool->stride; a.o_pil = pool->o_pil; a.o_plan = pool->o_plan;
    a.num_pilots = d_np; a.given_idx = pilot_idx; a.pattern_ids = d_ids; a.seed = seed; a.idx_stride = idx_stride;
    a.first_slot = first_slot; a.S = ctx->fp.S; a.nsc = ctx->fp.nsc; a.method = method; a.Pcap = pool->Pcap;
    a.simp_out = simplices_out; a.nbr_out = simplices_out ? neighbors_out : nullptr; a.nsimp_out = num_simplices_out; a.status = ctx->d_status;
    if (simplices_out && pool->Pcap != Pmax) { set_error("chest_patterns_build: simplices_out needs a fresh slot range (row stride is 2*Pmax*3)"); return CHEST_EINVAL; }
    const size_t lds_run = pattern_build_lds_bytes(G, pool->Pcap);
    CHEST_LDS_OPT_IN(ctx, LDS_DELAUNAY, pattern_build_kernel, lds_run);
    const int sp = prof_open(ctx, st, CHEST_K_PLAN);
    hipLaunchKernelGGL(pattern_build_kernel, dim3((unsigned)count), dim3(PB_THREADS), lds_run, st, a);
    prof_close(ctx, st, sp);
    CHEST_HIP(hipGetLastError());
    // host mirror of the table entries
    for (int i = 0; i < count; ++i) {
        PatternHost& ph = ctx->patterns[first_slot + i];
        unsigned char* base = (unsigned char*)pool->block + (size_t)i * pool->stride;
        ph.block = nullptr;                                      // owned by the pool
        ph.dev.slot = (const uint16_t*)base;
        ph.dev.plan = method >= 0 ? (const void*)(base + pool->o_plan) : nullptr;
        ph.dev.pilot_idx = (const int32_t*)(base + pool->o_pil);
        ph.dev.P = num_pilots[i];
    }
    return CHEST_OK;
}

/* pilot list of a slot back to the host (sorted flat indices); synchronises on the device */
extern "C" int chest_pattern_get_pilots(chest_ctx* ctx, int32_t slot, int32_t* pilot_idx_out) {
    if (!ctx || !pilot_idx_out || slot < 0 || (size_t)slot >= ctx->patterns.size() || ctx->patterns[slot].dev.slot == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    CHEST_GUARD(ctx);
    CHEST_HIP(hipDeviceSynchronize());
    CHEST_HIP(hipMemcpy(pilot_idx_out, ctx->patterns[slot].dev.pilot_idx, (size_t)ctx->patterns[slot].dev.P * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CHEST_OK;
}
