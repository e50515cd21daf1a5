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
//   2. pilot bit table: per 32 resource elements their pilot mask and the number of pilots before them (the draw slot of an
//      element follows from a popcount: chest_device.h draw_slot).
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
#include <algorithm>

#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

#ifndef CHEST_PB_THREADS
#define CHEST_PB_THREADS 256
#endif
constexpr int PB_THREADS_DEFAULT = CHEST_PB_THREADS;      // >= 256 (the radix selection keeps its 256 bins on the first 256 threads)
constexpr int PB_THREADS_NEAREST = 512;                   // 'nearest' plans: no flips, no serial stage -- eight waves per pattern (3.9 -> 5.3 M patterns/s)
constexpr uint32_t TNONE = 0xFFFFu;
constexpr int SEL_SHIFT = 20, SEL_BINS = 1 << (32 - SEL_SHIFT);      // pilot selection: histogram over the keys' top 12 bits
constexpr int SEL_LIST = 256;                                        // capacity of the threshold bin's element list (mean G / 4096 <= 16)
constexpr int PB_MAX_ROWS = CHEST_MAX_SYMBOLS;

struct PatBuildArgs {
    PatternDev* table;            // device pattern table; entries first_slot + j are written by the kernel
    unsigned char* pool;          // count blocks of `stride` bytes: pilot bit table uint2[G / 32 + 1] | pilot_idx i32[Pcap] | plan PlanRec[G]
    size_t stride, o_pil, o_plan, o_tri;   // o_tri: compact plans only (method 2): u16 triangle id per element behind the record table
    const int32_t* num_pilots;    // [count]
    const int32_t* given_idx;     // [count][idx_stride] sorted flat indices, or nullptr: draw them
    const uint64_t* pattern_ids;  // [count]
    uint64_t seed;
    int32_t idx_stride, first_slot, S, nsc, method, Pcap;
    int32_t* simp_out;            // optional [count][2*Pcap][3]
    int32_t* nbr_out;             // optional [count][2*Pcap][3]: triangle across the edge opposite each vertex, -1 on the hull
    int32_t* nsimp_out;           // optional [count]
    int32_t* status;
    int32_t flip_cap;             // flip rounds allowed per stage (0: 4 * NT + 64, never reached); chest_tune "pattern_flip_cap"
    int32_t solo_at;              // active triangles at which wave 0 takes the flips over (0: the default)
    int32_t stop_after;           // diagnostic (chest_tune "pattern_stop_after"): leave after phase k = 1 selection, 2 slot table, 3 strips,
                                  // 4 pockets, 5 flips (0: run to the end) -- instruction counts per phase from counter differences
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
// x[i mod 3] for i in 0 .. 4 without a run-time subscript
__device__ __forceinline__ uint32_t sel3(const uint32_t (&x)[3], int i) {
    return (i == 0 || i == 3) ? x[0] : ((i == 1 || i == 4) ? x[1] : x[2]);
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

// number of j in [0, n - 1) with col(a[j]) + col(a[j + 1]) < key / <= key (a: one row's pilots, ascending columns)
__device__ __forceinline__ int count_pairsum_less(const uint32_t* a, int n, int key) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (py_of(a[mid]) + py_of(a[mid + 1]) < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int count_pairsum_leq(const uint32_t* a, int n, int key) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (py_of(a[mid]) + py_of(a[mid + 1]) <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

// exclusive scan of one value per thread over the workgroup (wave shuffles + one LDS hop); `scr` >= 8 ints of LDS;
// returns (exclusive prefix, total)
template <int PB_THREADS>
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


// position of the j-th (0-based) set bit of b (j < popcount(b)); b is wave-uniform, j per lane: vector ALU only
__device__ __forceinline__ int nth_set_bit(uint64_t b, int j) {
    int pos = 0;
    uint32_t w = (uint32_t)b;
    int c = __popc(w);
    if (j >= c) { j -= c; pos = 32; w = (uint32_t)(b >> 32); }
    c = __popc(w & 0xFFFFu); if (j >= c) { j -= c; pos += 16; w >>= 16; }
    c = __popc(w & 0xFFu);   if (j >= c) { j -= c; pos += 8;  w >>= 8; }
    c = __popc(w & 0xFu);    if (j >= c) { j -= c; pos += 4;  w >>= 4; }
    c = __popc(w & 0x3u);    if (j >= c) { j -= c; pos += 2;  w >>= 2; }
    if (j >= (int)(w & 1u)) pos += 1;
    return pos;
}

// Work items are bits of per-lane masks (item = (owner lane, bit)).  The masks of a wave are uneven -- in a flip round the
// busiest lane holds two to three times the average -- and a loop over the lane's own bits runs as long as the busiest lane,
// each pass a chain of dependent LDS reads.  Here level p of the masks (the p-th item of every lane that has one; a ballot)
// is packed together with the following levels while they fit into the 64 lanes of one pass: lane l of a pass takes the
// (l - offset)-th lane of its level's ballot (vector ALU on the uniform ballot word) and reads that lane's mask with one
// ds_bpermute.  fn(owner_lane, bit) runs on about ceil(items / 64) passes instead of max-items-per-lane passes.  All control
// flow here is wave-uniform; no barrier or cross-lane operation may occur inside fn.
template <class F>
__device__ __forceinline__ void wave_balanced(uint32_t mask, int lane, F&& fn) {
    const int cnt = __popc(mask);
    if (__ballot(cnt > 1) == 0) {                               // nothing to balance (most rounds after the first few)
        if (mask != 0) fn(lane, __ffs(mask) - 1);
        return;
    }
    int p = 0;
    uint64_t Bp = __ballot(cnt > 0);
    while (Bp != 0) {
        int filled = 0, owner = -1, level = 0;
        do {
            const int np = __popcll(Bp);
            if (filled + np > 64) break;                        // (the first level of a pass always fits)
            const int j = lane - filled;
            if (j >= 0 && j < np) { owner = nth_set_bit(Bp, j); level = p; }
            filled += np;
            ++p;
            Bp = __ballot(cnt > p);
        } while (Bp != 0);
        uint32_t m = __shfl(mask, owner < 0 ? lane : owner, 64);
        for (int q = 0; q < level; ++q) m &= m - 1;
        if (owner >= 0) fn(owner, __ffs(m) - 1);
    }
}

size_t pattern_build_lds_bytes(int G, int Pcap, int PB_THREADS = PB_THREADS_DEFAULT) {
    const size_t tri = (size_t)4 * Pcap + 16 + (size_t)2 * Pcap * (sizeof(uint16_t) * 6 + 4);            // pts | triangle records
    const size_t keys = (size_t)4 * (SEL_BINS + 2 * SEL_LIST);                                        // selection: histogram + threshold-bin list
    (void)G;
    return (tri > keys ? tri : keys) + (size_t)(PB_THREADS + 16 + 3 * (PB_MAX_ROWS + 2)) * sizeof(int) + 64;
}

#ifdef CHEST_STAMPS
__device__ unsigned long long* g_pat_stamp = nullptr;           // [count][16]: cycles of wave 0 per phase, then detail slots (diagnostic build only)
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                       if (tid == 0 && g_pat_stamp) g_pat_stamp[(size_t)blockIdx.x * 16 + (i)] = now_ - pst_; pst_ = now_; \
                       __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

// COMPACT: method 2 (compact 'linear' plan); a template parameter so that the full-plan kernel carries none of its code
// Launch bounds: eight 256-thread workgroups per CU, i.e. at most 64 VGPRs and 80 SGPRs (with more SGPRs the hardware admits
// fewer workgroups than the LDS would, MI355X_MICROARCH.md "Residency"; the scalar spills this costs go to VGPR lanes).  Sparse
// patterns, whose LDS footprint lets more than five workgroups share a CU, gain 10 % (P = 419: 1.48 -> 1.33 ms per 10 000,
// P = 83: 1.05 -> 0.90); at P = 838 the LDS holds five either way and the kernel is bound by its vector instructions (r03 PMC:
// 94 k VALU instructions per pattern, vector ALU 82 % busy) -- a 12-B packed record format that fits a sixth workgroup was
// built, measured at the same 1.85-1.9 ms and dropped (profiles/r03_pattern_ab_formats.txt).
template <bool COMPACT, int PB_THREADS>
__global__ void __launch_bounds__(PB_THREADS, 8)
pattern_build_kernel(const PatBuildArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, j = blockIdx.x;
#ifdef CHEST_STAMPS
    unsigned long long pst_ = __builtin_amdgcn_s_memtime();
#endif
    const int S = A.S, nsc = A.nsc, G = S * nsc;
    const int P = A.num_pilots[j];
    unsigned char* base = A.pool + (size_t)j * A.stride;
    uint2* __restrict__ bitsp = (uint2*)base;
    int32_t* __restrict__ pil = (int32_t*)(base + A.o_pil);
    PlanRec* __restrict__ plan = (PlanRec*)(base + A.o_plan);
    if (tid == 0) {
        PatternDev pd;
        pd.bits = bitsp; pd.plan = A.method >= 0 ? (const void*)plan : nullptr; pd.pilot_idx = pil; pd.P = P;
        pd.pad = COMPACT ? (int32_t)(A.o_tri - A.o_plan) : 0;       // compact plan: bytes from the triangle records to the per-element ids
        A.table[A.first_slot + j] = pd;
        if (A.nsimp_out) A.nsimp_out[j] = 0;
    }
    if (P < 1 || P > A.Pcap || P > G) { if (tid == 0) atomicExch(A.status, 1); return; }

    // ---- LDS carve-up: the key array of step 1 shares its space with the triangle arrays of steps 3-5 ----------------
    const int NT = 2 * A.Pcap;
    int* misc = (int*)lds_raw;                                   // [PB_THREADS + 16 + 3 * (rows + 2)]
    int* scan = misc;                                            // PB_THREADS + 1: workgroup scans of the pilot selection, then
    int* cnt_s = misc + PB_THREADS + 1;                          // a few scalars ([6]: active triangles of the workgroup, flip rounds)
    int* rowstart = misc + PB_THREADS + 16;                      // [rows + 1] first pilot of every row
    int* occ = rowstart + PB_MAX_ROWS + 2;                       // occupied rows
    int* toff = occ + PB_MAX_ROWS + 2;                           // first triangle of every strip
    static_assert(PB_MAX_ROWS <= 32, "one lane of a wave per row (row tables, pocket scan)");
    uint32_t* pmask = (uint32_t*)scan;                           // ... and in the flip rounds [PB_THREADS]: my triangles that proposed a flip
    // (byte offsets from the LDS base, never pointer -> integer -> pointer: such a round trip loses the LDS address space and
    // every access below becomes a FLAT instruction)
    const unsigned off_big = ((unsigned)(PB_THREADS + 16 + 3 * (PB_MAX_ROWS + 2)) * 4u + 15u) & ~15u;
    uint32_t* pts = (uint32_t*)(lds_raw + off_big);
    // one 16-B record per triangle: vertices h[0..3), neighbours across the edge opposite each vertex h[3..6), claim word
    const unsigned off_tr = (off_big + 4u * (unsigned)A.Pcap + 15u) & ~15u;
    TriRec* TR = (TriRec*)(lds_raw + off_tr);

    // ================= 1. pilots =================================================================================
    const int per = (G + PB_THREADS - 1) / PB_THREADS;           // contiguous element block of this thread
    const int g_lo = min(G, tid * per), g_hi = min(G, g_lo + per);
    if (A.given_idx != nullptr) {
        for (int i = tid; i < P; i += PB_THREADS) pil[i] = A.given_idx[(size_t)j * A.idx_stride + i];
    } else {
        const uint64_t pid = A.pattern_ids[j];
        const RngKey key{(uint32_t)A.seed, (uint32_t)(A.seed >> 32), (uint32_t)pid, (uint32_t)(pid >> 32)};
        // The P smallest keys by ONE histogram over the keys' top 12 bits: everything in a bin below the bin that holds the
        // P-th smallest key is a pilot, and that bin's few elements (G / 4096 on average) are ranked exactly by (key, element).
        // Keys are recomputed in the second pass (a Philox call keys four elements; 9 calls per thread and pass at 14 x 599) instead
        // of being kept in LDS: the selection then needs 18 KB whatever the grid size (a key array is 4 G bytes = 33.5 KB
        // at 14 x 599, which capped sparse patterns at four workgroups per CU).
        uint32_t* bins = (uint32_t*)(lds_raw + off_big);                       // [SEL_BINS]
        uint32_t* lst = bins + SEL_BINS;                                       // [SEL_LIST][2]: (key, element) of the threshold bin
        const int nq = (G + 3) >> 2;
        const int qper = (nq + PB_THREADS - 1) / PB_THREADS;
        const int q_lo = min(nq, tid * qper), q_hi = min(nq, q_lo + qper);     // contiguous elements 4 q_lo ... 4 q_hi - 1
        auto for_my_keys = [&](auto&& f) {                                     // f(key, element) in element order
            for (int q = q_lo; q < q_hi; ++q) {
                const u32x4 w = draw(key, STREAM_PATTERN, (uint32_t)q);
                const int g = 4 * q;
                f(w.x, g);
                if (g + 1 < G) f(w.y, g + 1);
                if (g + 2 < G) f(w.z, g + 2);
                if (g + 3 < G) f(w.w, g + 3);
            }
        };
        for (int b = tid; b < SEL_BINS; b += PB_THREADS) bins[b] = 0;
        if (tid == 0) cnt_s[5] = 0;
        __syncthreads();
        for_my_keys([&](uint32_t k, int) { atomicAdd(&bins[k >> SEL_SHIFT], 1u); });
        __syncthreads();
        constexpr int BPT = SEL_BINS / 256;                                    // bins per scanning thread
        int h = 0;
        if (tid < 256) for (int i = 0; i < BPT; ++i) h += (int)bins[tid * BPT + i];
        const int2 ex = block_exscan<PB_THREADS>(h, scan);                                 // keys in lower bin groups
        if (tid < 256 && ex.x < P && P <= ex.x + h) {                          // exactly one group qualifies
            int acc = ex.x;
            for (int i = 0; i < BPT; ++i) {
                const int c = (int)bins[tid * BPT + i];
                if (P <= acc + c) { cnt_s[0] = tid * BPT + i; cnt_s[4] = P - acc; break; }
                acc += c;
            }
        }
        __syncthreads();
        const uint32_t tb = (uint32_t)cnt_s[0];                                // threshold bin
        const int need = cnt_s[4];                                             // pilots to take from it (>= 1)
        // second pass: count my elements below the threshold bin, list those inside it; where a thread's block has at most 64
        // elements (grids up to 16 384 elements) their classification is kept as two bit masks and the third pass needs no keys
        const bool masked = 4 * qper <= 64;
        int nl = 0;
        uint64_t m_lt = 0, m_eq = 0;
        for_my_keys([&](uint32_t k, int g) {
            const uint32_t b = k >> SEL_SHIFT;
            nl += b < tb ? 1 : 0;
            const uint64_t bit = 1ull << ((g - 4 * q_lo) & 63);
            if (b < tb) m_lt |= bit;
            if (b == tb) {
                m_eq |= bit;
                const int at = atomicAdd(&cnt_s[5], 1);
                if (at < SEL_LIST) { lst[2 * at] = k; lst[2 * at + 1] = (uint32_t)g; }
            }
        });
        __syncthreads();
        const int nlist = cnt_s[5];
        if (nlist > SEL_LIST) { if (tid == 0) atomicExch(A.status, 1); return; }   // not reached with uniform keys (uniform exit)
        // an element of the threshold bin is a pilot iff fewer than `need` of the bin's elements precede it in (key, element) order
        auto taken = [&](uint32_t k, int g) -> bool {
            int before = 0;
            for (int e = 0; e < nlist; ++e) {
                const uint32_t ke = lst[2 * e], ge = lst[2 * e + 1];
                before += (ke < k || (ke == k && ge < (uint32_t)g)) ? 1 : 0;
            }
            return before < need;
        };
        int mine = nl;
        uint64_t m_take = m_lt;
        for (int e = 0; e < nlist; ++e) {
            const int ge = (int)lst[2 * e + 1];
            if (ge >= 4 * q_lo && ge < 4 * q_hi && taken(lst[2 * e], ge)) { ++mine; m_take |= 1ull << ((ge - 4 * q_lo) & 63); }
        }
        (void)m_eq;
        const int2 sp = block_exscan<PB_THREADS>(mine, scan);
        int pos = sp.x;
        if (masked) {
            for (uint64_t mm = m_take; mm != 0; mm &= mm - 1) pil[pos++] = 4 * q_lo + (__ffsll((long long)mm) - 1);
        } else {
            for_my_keys([&](uint32_t k, int g) {
                const uint32_t b = k >> SEL_SHIFT;
                if (b < tb || (b == tb && taken(k, g))) pil[pos++] = g;
            });
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
    if (A.stop_after == 1) return;
    // ================= 2. pilot bit table ==========================================================================
    // {mask of 32 elements, pilots before them} per word (PatternDev::bits): every pilot sets its bit in LDS (the triangle area is
    // still free), one thread per word scans the popcounts, 8-B stores.  (Rounds 1-4 wrote a u16 draw slot per element here: 16.8 KB
    // of a 14 x 599 pattern's 64 KB, 3 000 vector instructions of its 94 000.)
    {
        const int W = pattern_bit_words(G);                      // <= 2 049 words (G <= 65 535)
        uint32_t* __restrict__ wm = (uint32_t*)(lds_raw + off_tr);
        for (int w = tid; w < W; w += PB_THREADS) wm[w] = 0;
        __syncthreads();
        for (int i = tid; i < P; i += PB_THREADS) { const uint32_t g = (uint32_t)pil[i]; atomicOr(&wm[g >> 5], 1u << (g & 31u)); }
        __syncthreads();
        const int wper = (W + PB_THREADS - 1) / PB_THREADS;
        const int w_lo = min(W, tid * wper), w_hi = min(W, w_lo + wper);
        int mine = 0;
        for (int w = w_lo; w < w_hi; ++w) mine += __popc(wm[w]);
        const int2 ex = block_exscan<PB_THREADS>(mine, scan);
        int before = ex.x;
        for (int w = w_lo; w < w_hi; ++w) { const uint32_t m = wm[w]; bitsp[w] = make_uint2(m, (uint32_t)before); before += __popc(m); }
        __syncthreads();                                         // (the LDS words are free again for the triangle records)
    }
    PSTAMP(1);                                                   // slot table
    if (A.stop_after == 2) return;
    if (A.method < 0) return;                                    // simulation-only pattern: no plan

    // per-row tables
    for (int r = tid; r <= S; r += PB_THREADS) rowstart[r] = count_less(pts, P, (uint32_t)r << 16);
    __syncthreads();
    if (tid < 64) {
        // occupied rows and the first triangle of every strip: lane r looks at row r (S <= 32 rows)
        const int n_r = tid < S ? rowstart[tid + 1] - rowstart[tid] : 0;
        const uint64_t occm = __ballot(n_r > 0);
        const int q = __popcll(occm & ((1ull << tid) - 1ull));   // my row's rank among the occupied rows
        const int mm = __popcll(occm);
        if (n_r > 0) occ[q] = tid;
        // strip q joins occupied rows q and q + 1: n_q + n_{q+1} - 2 triangles; lane q of the COMPACTED order holds n of row occ[q]
        const int src = tid < mm ? nth_set_bit(occm, tid) : 0;
        const int n_q = __shfl(n_r, src, 64);
        const int n_q1 = __shfl(n_q, tid + 1 < 64 ? tid + 1 : 63, 64);
        const int sz = tid + 1 < mm ? n_q + n_q1 - 2 : 0;
        int incl = sz;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { const int u = __shfl_up(incl, off, 64); if (tid >= off) incl += u; }
        const int total = __shfl(incl, 31, 64);
        if (tid < mm) toff[tid] = incl - sz;                      // (strip mm - 1 does not exist: its entry is the total)
        if (tid == 0) { cnt_s[1] = mm; cnt_s[2] = total; }       // occupied rows (>= 1: P >= 1), strip triangles
    }
    __syncthreads();
    const int m = cnt_s[1];

    if (A.method == 1) {
        // ================= 'nearest' plan ========================================================================
        int g = g_lo;
        int s = g / nsc, k = g - s * nsc;
        // occupied rows are visited outwards from the element's own row and the search stops once a row is further away than
        // the best pilot found (rows are 1 apart, pilots of a row ~10): a handful of rows instead of all of them.  Ties go to
        // the smaller pilot ordinal whatever the visiting order.
        int qlo = 0;                                             // occupied rows below index qlo lie above row s
        while (qlo < m && occ[qlo] < s) ++qlo;
        for (; g < g_hi; ++g) {
            int best = 0x7fffffff, bi = 0;
            auto visit = [&](int q) {
                const int r = occ[q], a = rowstart[r], n = rowstart[r + 1] - a;
                const int c = count_less(pts + a, n, ((uint32_t)r << 16) | (uint32_t)k);      // first column >= k
                const int dr = (r - s) * (r - s);
                if (c > 0) { const int dc = k - py_of(pts[a + c - 1]); const int d2 = dr + dc * dc; if (d2 < best || (d2 == best && a + c - 1 < bi)) { best = d2; bi = a + c - 1; } }
                if (c < n) { const int dc = py_of(pts[a + c]) - k; const int d2 = dr + dc * dc; if (d2 < best || (d2 == best && a + c < bi)) { best = d2; bi = a + c; } }
            };
            int up = qlo - 1, dn = qlo;                          // next candidates above / at-or-below row s
            while (up >= 0 || dn < m) {
                const int du = up >= 0 ? s - occ[up] : 0x7fff, dd = dn < m ? occ[dn] - s : 0x7fff;
                const int dmin = du < dd ? du : dd;
                if (dmin * dmin > best) break;                   // every remaining row is further away than the best pilot
                if (du <= dd) visit(up--); else visit(dn++);
            }
            plan[g] = PlanRec{1.0f, 0.0f, (uint32_t)bi | ((uint32_t)bi << 16), (uint32_t)bi | (1u << 16)};
            if (++k == nsc) { k = 0; ++s; while (qlo < m && occ[qlo] < s) ++qlo; }
        }
        return;
    }

    // ================= 3. initial triangulation ====================================================================
    PSTAMP(2);                                                   // row tables
    const int nstrip = cnt_s[2];
    // every pilot emits the triangle that advances from it to its right-hand neighbour in its row: once as a point of the
    // upper row of the strip below it (A-step) and once as a point of the lower row of the strip above it (B-step).
    // Merge rule = the Delaunay triangulation of the two rows alone: the apex of a row edge is the pilot of the other row
    // nearest to the edge's midpoint, i.e. an A-step over (a_i, a_i+1) comes after exactly the B-steps whose edge midpoint lies
    // to its left (ties: the upper row first).  Compared with merging by column this start is 17 % of the flips and a
    // quarter of the flip rounds closer to the Delaunay triangulation (tools/delaunay_model.py).
    for (int p = tid; p < P; p += PB_THREADS) {
        const int r = px_of(pts[p]);
        int q = 0;
        while (occ[q] != r) ++q;                                 // occupied-row index (<= S steps)
        const int a0 = rowstart[r], na = rowstart[r + 1] - a0, i = p - a0;
        if (i + 1 >= na) continue;                               // last point of its row: no step starts here
        const int esum = py_of(pts[p]) + py_of(pts[p + 1]);      // twice the midpoint column of the edge (p, p + 1)
        int jb = 0, ia = 0;
        if (q + 1 < m) {                                         // B-steps of strip q before the A-step over this edge
            const int rb = occ[q + 1], b0 = rowstart[rb], nb = rowstart[rb + 1] - b0;
            jb = count_pairsum_less(pts + b0, nb, esum);
        }
        if (q > 0) {                                             // A-steps of strip q - 1 before the B-step over this edge
            const int ru = occ[q - 1], u0 = rowstart[ru], nu = rowstart[ru + 1] - u0;
            ia = count_pairsum_leq(pts + u0, nu, esum);
        }
        if (q + 1 < m) {
            // A-step of strip q: triangle (a_i, b_j, a_{i+1}), j = B-steps taken before
            const int rb = occ[q + 1], b0 = rowstart[rb], nb = rowstart[rb + 1] - b0;
            const int k = i + jb, t = toff[q] + k, nq = na + nb - 2;
            st_tri(TR, (uint32_t)t, (uint32_t)p, (uint32_t)(b0 + jb), (uint32_t)(p + 1),
                   (uint32_t)(k + 1 < nq ? t + 1 : TNONE),                      // rung after the step
                   q > 0 ? (uint32_t)(toff[q - 1] + i + ia) : TNONE,            // across the row edge: the B-step of the strip above over the same edge
                   (uint32_t)(k > 0 ? t - 1 : TNONE), 0u);                      // rung before the step
        }
        if (q > 0) {
            // B-step of strip q-1: triangle (b_j, b_{j+1}, a_i) with j = i here
            const int ru = occ[q - 1], u0 = rowstart[ru], nu = rowstart[ru + 1] - u0;
            const int k = i + ia, t = toff[q - 1] + k, nq = nu + na - 2;
            st_tri(TR, (uint32_t)t, (uint32_t)p, (uint32_t)(p + 1), (uint32_t)(u0 + ia),
                   (uint32_t)(k + 1 < nq ? t + 1 : TNONE),                      // rung after
                   (uint32_t)(k > 0 ? t - 1 : TNONE),                           // rung before
                   q + 1 < m ? (uint32_t)(toff[q] + i + jb) : TNONE, 0u);       // across the row edge: the A-step of the strip below
        }
    }
    __syncthreads();
    PSTAMP(3);                                                   // strips
    if (A.stop_after == 3) return;
    // pockets between the end chains and the convex hull (<= 2 (m - 2) triangles).  A monotone-chain scan over the first (left) /
    // last (right) pilot of every occupied row; every boundary edge of the scan carries its owner (triangle, slot), so each
    // pocket triangle is linked to its neighbours in O(1).  The scan itself is serial, but nothing in it waits for the LDS:
    // lane q of wave 0 first fetches everything row q contributes (its two chain points, their positions, the strip triangle
    // that owns the chain edge towards the previous row and the slot of that edge), the chain stacks live in lane registers
    // (entry i in lane i: written by a select on the lane number, read by v_readlane with the uniform stack pointer), and the
    // whole wave runs the scan with uniform control flow; lane 0 issues the stores (r02: one thread's chain of dependent LDS
    // reads, 47 k cycles = a tenth of the kernel).
    if (tid < 64) {
        auto rd = [](uint32_t v, int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(i)); };
        auto back_link = [&](uint32_t own, int t) {               // the owner of an edge learns its new neighbour
            if (own != 0xFFFFFFFFu) TR[own >> 2].h[3 + ((int)(own & 3u))] = (uint16_t)t;
        };
        auto slot_of_edge = [&](int t, uint32_t x, uint32_t y) -> int {     // slot of t whose edge joins x and y
            const TriU Te = ld_tri(TR, (uint32_t)t);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const uint32_t e0 = Te.v[(d + 1) % 3], e1 = Te.v[(d + 2) % 3];
                if ((e0 == x && e1 == y) || (e0 == y && e1 == x)) return d;
            }
            return 0;
        };
        const int lq = tid;
        uint32_t P_side[2] = {0u, 0u}, PP_side[2] = {0u, 0u}, CUR_side[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        uint32_t EMP = 0u;                                       // strip lq - 1 has no triangle (both rows hold one pilot)
        if (lq < m) {
            const int r = occ[lq];
            P_side[0] = (uint32_t)rowstart[r]; P_side[1] = (uint32_t)(rowstart[r + 1] - 1);
            PP_side[0] = pts[P_side[0]]; PP_side[1] = pts[P_side[1]];
            if (lq > 0) {
                const int t0 = toff[lq - 1], nq = toff[lq] - t0;
                if (nq > 0) {
                    const int rp = occ[lq - 1];
                    CUR_side[0] = ((uint32_t)t0 << 2) | (uint32_t)slot_of_edge(t0, (uint32_t)rowstart[rp], P_side[0]);
                    CUR_side[1] = ((uint32_t)(t0 + nq - 1) << 2) | (uint32_t)slot_of_edge(t0 + nq - 1, (uint32_t)(rowstart[rp + 1] - 1), P_side[1]);
                } else EMP = 1u;
            }
        }
        uint32_t RUNG = 0xFFFFFFFFu;                              // [lane q] who took the bare chain edge of strip q on the left pass
        int nt = nstrip;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            uint32_t STACK = 0u, SPOS = 0u, EOWN = 0xFFFFFFFFu, EQ = 0xFFFFFFFFu;    // [lane i] chain stack: pilot, position, edge owner, bare strip
            int sp = 0;
            uint32_t top = 0, sec = 0, ptop = 0, psec = 0;       // stack[sp - 1], stack[sp - 2] and their positions
            for (int q = 0; q < m; ++q) {
                const uint32_t p = rd(P_side[side], q), pp = rd(PP_side[side], q);
                uint32_t cur = rd(CUR_side[side], q);            // owner of the chain edge (previous chain point, p)
                uint32_t cur_q = 0xFFFFFFFFu;                    // cur is still the bare chain edge of that (triangle-less) strip
                if (rd(EMP, q) != 0u) { cur = rd(RUNG, q - 1); cur_q = (uint32_t)(q - 1); }    // the edge belongs to both chains
                while (sp >= 2) {
                    const int o = orient2(psec, ptop, pp);
                    // the top point lies in a pocket if it is on the inner side of sec -> p (left chain: o < 0, right chain: o > 0)
                    if (!(side == 0 ? o < 0 : o > 0)) break;
                    // counter-clockwise: (a, p, b) on the left chain, (a, b, p) on the right one; a = sec, b = top
                    const int t = nt++;
                    const uint32_t own_ab = rd(EOWN, sp - 1);
                    const uint32_t n_ab = own_ab != 0xFFFFFFFFu ? own_ab >> 2 : TNONE, n_bp = cur != 0xFFFFFFFFu ? cur >> 2 : TNONE;
                    // edge (b, p) is opposite a (slot 0); edge (a, b) is opposite p: slot 1 on the left chain, 2 on the right one;
                    // edge (a, p), opposite b, is the new chain edge: its neighbour comes later
                    const int s_ab = side == 0 ? 1 : 2, s_ap = side == 0 ? 2 : 1;
                    if (lq == 0) {
                        st_tri(TR, (uint32_t)t, sec, side == 0 ? p : top, side == 0 ? top : p,
                               n_bp, side == 0 ? n_ab : TNONE, side == 0 ? TNONE : n_ab, 0u);
                        back_link(own_ab, t);
                        back_link(cur, t);
                    }
                    // a chain edge shared by both chains (strip without triangles): remember who took it on the left pass
                    if (side == 0) {
                        const uint32_t e = rd(EQ, sp - 1);
                        if (e != 0xFFFFFFFFu && (uint32_t)lq == e) RUNG = ((uint32_t)t << 2) | (uint32_t)s_ab;
                        if (cur_q != 0xFFFFFFFFu && (uint32_t)lq == cur_q) RUNG = ((uint32_t)t << 2) | 0u;
                    }
                    cur_q = 0xFFFFFFFFu;
                    cur = ((uint32_t)t << 2) | (uint32_t)s_ap;
                    --sp;
                    top = sec; ptop = psec;
                    if (sp >= 2) { sec = rd(STACK, sp - 2); psec = rd(SPOS, sp - 2); }
                }
                if (lq == sp) { EOWN = cur; EQ = cur_q; STACK = p; SPOS = pp; }
                ++sp;
                sec = top; psec = ptop; top = p; ptop = pp;
            }
        }
        if (lq == 0) cnt_s[2] = nt;
    }
    __syncthreads();
    const int ntri = cnt_s[2];
    PSTAMP(4);                                                   // pockets
    if (A.stop_after == 4) return;

    // ================= 4. Lawson flips ==============================================================================
    // Rounds of independent flips.  A triangle is ACTIVE (its claim word starts the round as CLAIM_DIRTY) if it changed or lost a
    // claim in the previous round; everything is active at first.  Only active triangles test their edges -- an edge between
    // two active triangles from the smaller index -- so once the bulk of the flips is done a round costs one cheap pass.
    // claim word: CLAIM_NONE / CLAIM_DIRTY (inactive / active, unclaimed), the smallest key (4 t + edge < 2^18) of the edges that
    // claimed the triangle, or that key | CLAIM_WON once the edge has flipped: the two triangles of a flip find each other
    // through it (the owner t keeps its partner in neighbour slot 1, the partner keeps t in slot 2)
    //
    // A round is three phases that must not overlap (test + claim | rewrite | relink), and the kernel is bound by the
    // instructions it issues: a phase costs a wave its ~200-350 instructions whether 64 of its lanes have a triangle or two.
    // The active set shrinks fast (P = 838: 1 545, 623, 705, 598, 554, 376, 277, 167, 92, 57, 27, 16, 2 triangles per round,
    // tools/delaunay_model.py), so the rounds run in two stages:
    //   stage 0  all four waves, every thread the triangles t = tid + 256 i, phases separated by workgroup barriers -- while
    //            the workgroup's active set is worth four waves;
    //   stage 1  from FLIP_SOLO_AT active triangles on (and at most 2 048 triangles in all: 32 mask bits per lane), wave 0 alone
    //            takes every triangle (t = lane + 64 i) and finishes WITHOUT barriers -- the LDS executes a wave's operations in
    //            issue order, so program order separates the phases -- while the other three waves wait at the stage's end
    //            and issue nothing (r03 stamps: 8 of 14 rounds; a third of the kernel's instructions).
    // Both stages flip by the same rules, so the triangulation does not depend on where the switch happens.
    constexpr uint32_t CLAIM_NONE = 0xFFFFFFFFu, CLAIM_DIRTY = 0xFFFFFFFEu, CLAIM_WON = 0x40000000u;
    constexpr uint32_t CLAIM_KEEP = 0xFFFFFFFDu;                 // unclaimed but kept active (first round); any real key is smaller
    const int FLIP_SOLO_AT = A.solo_at > 0 ? A.solo_at : 192;      // (chest_tune "pattern_solo_at": experiments)
    const int lane = tid & 63, wave = tid >> 6, wbase = tid & ~63;
    const int round_cap = A.flip_cap > 0 ? A.flip_cap : 4 * NT + 64;
    const bool can_solo = ntri <= 2048 && PB_THREADS >= 256;
    // [64] hand-over to stage 1 and its per-round set of claimed triangles (bit t >> 6 of word t & 63) -- in the row tables'
    // place, which are dead once the pockets are filled (the LDS granule is 1 280 B: 25 granules = 32 000 B per workgroup is
    // what lets five workgroups share a CU)
    uint32_t* gmask = (uint32_t*)rowstart;
    static_assert(2 * (PB_MAX_ROWS + 2) >= 64, "gmask takes the place of rowstart + occ");
    pmask[tid] = 0;
    if (tid < 64) gmask[tid] = 0;
    if (tid == 0) { cnt_s[3] = 0; cnt_s[6] = 0; }
#ifdef CHEST_STAMPS
    unsigned dbg_rounds = 0, dbg_cross = 0;
    unsigned long long dbg_ph[4] = {0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#define DBG_PH(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg_ph[i] += n_ - dbg_t; dbg_t = n_; } while (0)
    // inside the test phase of the one-wave stage (slots 12..15 of the record): record read | neighbours + pilots read | opposite
    // pilots read + in-circle tests + claims | everything drained; each stamp behind a forced wait for the LDS
    unsigned long long dbg_p0[4] = {0, 0, 0, 0}, dbg_u = 0;
    unsigned dbg_solo_rounds = 0;
#define DBG_P0(i) do { if (solo) { __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg_p0[i] += n_ - dbg_u; dbg_u = n_; } } while (0)
#else
#define DBG_PH(i) do { } while (0)
#define DBG_P0(i) do { } while (0)
#endif
    uint32_t active = 0;
    for (int i = 0, t = tid; t < ntri; t += PB_THREADS, ++i) { TR[t].claim = CLAIM_DIRTY; active |= 1u << i; }
    __syncthreads();
    int round = 0;
    bool done = false;                                            // no edge left to flip
    for (int stage = 0; stage < 2 && !done; ++stage) {
        const bool solo = stage == 1;
        const int istride = solo ? 64 : PB_THREADS, ilane = solo ? lane : tid, ibase = solo ? 0 : wbase;
        auto sync = [&]() {                                       // phase boundary
            if (solo) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // program order = LDS order inside a wave
            else __syncthreads();
        };
        if (solo) {
            // hand-over: everybody's active triangles into wave 0's masks
            for (uint32_t mm = active; mm != 0; mm &= mm - 1) {
                const uint32_t t = (uint32_t)(tid + PB_THREADS * (__ffs(mm) - 1));
                atomicOr(&gmask[t & 63u], 1u << (t >> 6));
            }
            __syncthreads();
            if (wave != 0) break;
            active = gmask[lane];
            gmask[lane] = 0;                                      // from here on: the round's claimed triangles
            pmask[lane] = 0;
            sync();
        }
        for (; round < round_cap; ++round) {
#ifdef CHEST_STAMPS
            ++dbg_rounds;
#endif
            // phase 0: active triangles test their edges; an edge that fails the in-circle test claims both of its triangles.
            // One 16-B read brings a triangle; the three neighbours and then the three opposite pilots are fetched together.
#ifdef CHEST_STAMPS
            if (solo) { ++dbg_solo_rounds; dbg_u = __builtin_amdgcn_s_memtime(); }
#endif
            // The FIRST round knows more than the later ones: every strip is the Delaunay triangulation of its two rows, so the
            // only edges that can fail are the ROW edges (shared by the strips above and below a row) -- one in-circle test per
            // strip triangle instead of three (r03 PMC: the first round was 9.9 k of the flips' 53.7 k vector instructions).  Pocket
            // triangles and the strip triangles next to them carry unchecked edges: they take no test here and stay active
            // (CLAIM_KEEP: unclaimed, but alive for the next round, which tests all their edges).
            if (round == 0 && !solo) {
                wave_balanced(active, lane, [&](int owner, int bit) {
                    const uint32_t t = (uint32_t)(ibase + owner + istride * bit);
                    const TriU T0 = ld_tri(TR, t);
                    const bool pocketish = t >= (uint32_t)nstrip || (T0.n[0] != TNONE && T0.n[0] >= (uint32_t)nstrip) ||
                                           (T0.n[1] != TNONE && T0.n[1] >= (uint32_t)nstrip) || (T0.n[2] != TNONE && T0.n[2] >= (uint32_t)nstrip);
                    const uint32_t p0 = pts[T0.v[0]], p1 = pts[T0.v[1]], p2 = pts[T0.v[2]];
                    // A-step (p, apex, p + 1): the row edge lies opposite vertex 1; B-step (p, p + 1, apex): opposite vertex 2
                    const bool a_step = px_of(p0) == px_of(p2);
                    const uint32_t u = a_step ? T0.n[1] : T0.n[2];
                    const bool need = !pocketish && u != TNONE && t < u;                 // (both sides are active: the smaller one tests)
                    const TriU U0 = ld_tri(TR, need ? u : t);
                    const uint32_t b = a_step ? T0.v[2] : T0.v[0], c = a_step ? T0.v[0] : T0.v[1];   // the edge (b, c), counter-clockwise after the apex
                    const uint32_t d = (U0.v[0] != b && U0.v[0] != c) ? U0.v[0] : ((U0.v[1] != b && U0.v[1] != c) ? U0.v[1] : U0.v[2]);
                    const uint32_t pd = pts[need ? d : T0.v[0]];
                    const uint32_t pa = a_step ? p1 : p2, pb = a_step ? p2 : p0, pc = a_step ? p0 : p1;
                    bool mark = pocketish;
                    if (pocketish) atomicMin(&TR[t].claim, CLAIM_KEEP);
                    if (need && incircle(pa, pb, pc, pd) > 0) {
                        const uint32_t key = t * 4u + (a_step ? 1u : 2u);
                        atomicMin(&TR[t].claim, key);
                        atomicMin(&TR[u].claim, key);
                        atomicOr(&pmask[ibase + owner], 1u << bit);
                        mark = true;
                    }
                    if (mark) cnt_s[3] = 1;
                });
            } else
            wave_balanced(active, lane, [&](int owner, int bit) {
                const uint32_t t = (uint32_t)(ibase + owner + istride * bit);
                const TriU T0 = ld_tri(TR, t);
                DBG_P0(0);
                const uint32_t pv[3] = {pts[T0.v[0]], pts[T0.v[1]], pts[T0.v[2]]};
                TriU U[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) U[i] = ld_tri(TR, T0.n[i] == TNONE ? t : T0.n[i]);
                DBG_P0(1);
                uint32_t pd[3];
                bool need[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const uint32_t u = T0.n[i];
                    need[i] = !(u == TNONE || (u < t && U[i].claim != CLAIM_NONE));       // else the other (active) side tests it
                    const uint32_t b = T0.v[(i + 1) % 3], c = T0.v[(i + 2) % 3];
                    const uint32_t d = (U[i].v[0] != b && U[i].v[0] != c) ? U[i].v[0] : ((U[i].v[1] != b && U[i].v[1] != c) ? U[i].v[1] : U[i].v[2]);
                    pd[i] = pts[need[i] ? d : T0.v[i]];
                }
                bool proposed = false;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (need[i] && incircle(pv[i], pv[(i + 1) % 3], pv[(i + 2) % 3], pd[i]) > 0) {
                        const uint32_t key = t * 4u + (uint32_t)i;
                        atomicMin(&TR[t].claim, key);
                        atomicMin(&TR[T0.n[i]].claim, key);
                        proposed = true;
                        // stage 1 keeps the round's CLAIMED triangles in gmask so that the end of the round visits them instead
                        // of scanning all the triangles the one wave owns
                        if (solo) atomicOr(&gmask[T0.n[i] & 63u], 1u << (T0.n[i] >> 6));
                    }
                }
                if (proposed) {
                    atomicOr(&pmask[ibase + owner], 1u << bit);
                    if (solo) atomicOr(&gmask[t & 63u], 1u << (t >> 6)); else cnt_s[3] = 1;
                }
                DBG_P0(2);
            });
            sync();
            DBG_P0(3);
            DBG_PH(0);
            const uint32_t proposed = pmask[ilane];
            if (solo ? __ballot(proposed != 0) == 0 : cnt_s[3] == 0) { done = true; break; }
            // phase 1: winners (an edge that holds the claim on both of its triangles) rewrite the two triangles; outer neighbours
            // keep their OLD ids for now.  Only triangles that proposed can own a winning edge.
            wave_balanced(proposed, lane, [&](int owner, int bit) {
                const uint32_t t = (uint32_t)(ibase + owner + istride * bit);
                const TriU T0 = ld_tri(TR, t);
                const uint32_t key = T0.claim;
                if ((key >> 2) != t) return;                     // the winning edge of t was tested from another triangle
                // (run-time subscripts into the register copies of a record would put them in scratch memory: selects instead)
                const int i = (int)(key & 3u);
                const uint32_t u = sel3(T0.n, i);
                const TriU U0 = ld_tri(TR, u);
                if (U0.claim != key) return;
                const uint32_t a = sel3(T0.v, i), b = sel3(T0.v, i + 1), c = sel3(T0.v, i + 2);
                const uint32_t n_ab = sel3(T0.n, i + 2), n_ca = sel3(T0.n, i + 1);
                const int jd = (U0.v[0] != b && U0.v[0] != c) ? 0 : ((U0.v[1] != b && U0.v[1] != c) ? 1 : 2);
                const uint32_t d = sel3(U0.v, jd);
                // u = (d, c, b) rotated so that u[jd] = d: the edge opposite c is (b, d), the edge opposite b is (d, c)
                const uint32_t n_db = sel3(U0.n, jd + 1), n_cd = sel3(U0.n, jd + 2);
                st_tri(TR, t, a, b, d, n_db, u, n_ab, key | CLAIM_WON);
                st_tri(TR, u, a, d, c, n_cd, n_ca, t, key | CLAIM_WON);
            });
            if (tid == 0 && !solo) cnt_s[6] = 0;                 // (read after the round's last barrier, added to after this one)
            sync();
            DBG_PH(1);
            // phase 2: the owner of every flip resolves the outer neighbours of its two new triangles (the old neighbour, or its
            // partner if that one flipped too) and writes the back-pointer of neighbours that did not flip
            uint32_t won = 0;                                    // my triangles that own a flip of this round (claim = own key | CLAIM_WON)
            for (uint32_t mm = proposed; mm != 0; mm &= mm - 1) {
                const int bit = __ffs(mm) - 1;
                const uint32_t t = (uint32_t)(ilane + istride * bit);
                const uint32_t c = TR[t].claim;
                if ((c & 0xC0000000u) == CLAIM_WON && ((c & ~CLAIM_WON) >> 2) == t) won |= 1u << bit;
            }
            wave_balanced(won, lane, [&](int owner, int bit) {
                const uint32_t t = (uint32_t)(ibase + owner + istride * bit);
                const TriU Tt = ld_tri(TR, t);
                const uint32_t pT = Tt.n[1];
                const TriU Tp = ld_tri(TR, pT);
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const uint32_t X = side == 0 ? t : pT;
                    const uint32_t pX = side == 0 ? pT : t;
                    const TriU TX = side == 0 ? Tt : Tp;
                    TriU TY[3];
                    uint32_t pY[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t Y = (TX.n[k] == TNONE || TX.n[k] == pX) ? X : TX.n[k];
                        TY[k] = ld_tri(TR, Y);
                        const uint32_t cy = TY[k].claim;         // flipped in this round: its partner, else TNONE
                        pY[k] = (cy & 0xC0000000u) == CLAIM_WON ? ((((cy & ~CLAIM_WON) >> 2) == Y) ? TY[k].n[1] : TY[k].n[2]) : TNONE;
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
            });
            sync();
            DBG_PH(2);
            // claimed triangles (flipped, or candidates that lost) are the next round's active set; marks that drew no claim clear
            if (solo) {
                uint32_t next = 0;
                for (uint32_t mm = active | gmask[lane]; mm != 0; mm &= mm - 1) {       // this round's active and claimed triangles
                    const int i = __ffs(mm) - 1;
                    const uint32_t t = (uint32_t)(lane + 64 * i);
                    const bool act = TR[t].claim < CLAIM_DIRTY;
                    TR[t].claim = act ? CLAIM_DIRTY : CLAIM_NONE;
                    if (act) next |= 1u << i;
                }
                gmask[lane] = 0;
                active = next;
            } else {
                active = 0;
                for (int i = 0, t = tid; t < ntri; t += PB_THREADS, ++i) {
                    const uint32_t c = TR[t].claim;
                    if (c == CLAIM_NONE) continue;
                    const bool act = c < CLAIM_DIRTY;
                    TR[t].claim = act ? CLAIM_DIRTY : CLAIM_NONE;
                    if (act) active |= 1u << i;
                }
                if (can_solo) {                                   // the workgroup's active triangles
                    int c = __popc(active);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
                    if (lane == 0) atomicAdd(&cnt_s[6], c);
                }
                if (tid == 0) cnt_s[3] = 0;
            }
            pmask[ilane] = 0;
            sync();
            DBG_PH(3);
            if (!solo && can_solo && cnt_s[6] <= FLIP_SOLO_AT) { ++round; break; }       // few enough for one wave: stage 1
        }
        if (!solo && !can_solo) break;
    }
    if (!done && round >= round_cap && (tid & 63) == 0) atomicExch(A.status, 3);      // ran into the round limit: not a finished triangulation
    __syncthreads();

    PSTAMP(5);                                                   // flips
    if (A.stop_after == 5) return;
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

    // compact plan (method 2): one 16-B record per triangle -- the three pilot ordinals, the three vertex positions
    // (symbol << 11 | subcarrier) and 1 / (twice the area) -- and one u16 triangle id per resource element; the frame kernel
    // evaluates the two edge functions itself (chest_device.h: plan_rec).  Record 2 Pcap - 1 (never a triangle: at most
    // 2 P - 5 exist) stands for "outside the convex hull".
    constexpr bool compact = COMPACT;
    uint16_t* __restrict__ triof = (uint16_t*)(base + A.o_tri);
    const uint32_t t_outside = (uint32_t)(2 * A.Pcap - 1);
    if (compact) {
        uint4* __restrict__ tab = (uint4*)plan;
        for (int t = tid; t < ntri; t += PB_THREADS) {
            const TriU T0 = ld_tri(TR, (uint32_t)t);
            const uint32_t a = pts[T0.v[0]], b = pts[T0.v[1]], c = pts[T0.v[2]];
            const uint32_t pa = ((uint32_t)px_of(a) << 11) | (uint32_t)py_of(a), pb = ((uint32_t)px_of(b) << 11) | (uint32_t)py_of(b),
                           pc = ((uint32_t)px_of(c) << 11) | (uint32_t)py_of(c);
            tab[t] = make_uint4(T0.v[0] | (T0.v[1] << 16), T0.v[2] | (pa << 16), pb | (pc << 16), __float_as_uint(1.0f / (float)orient2(a, b, c)));
        }
        if (tid == 0) tab[t_outside] = make_uint4(0u, 0xFFFF0000u, 0u, 0u);
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
        // (kept in the claim word of record i for pilot i: the claims are dead after the flips)
        for (int i = tid; i < P; i += PB_THREADS) TR[i].claim = 0xFFFFFFFFu;
        __syncthreads();
        for (int t = tid; t < ntri; t += PB_THREADS) { atomicMin(&TR[TR[t].h[0]].claim, (uint32_t)t); atomicMin(&TR[TR[t].h[1]].claim, (uint32_t)t); atomicMin(&TR[TR[t].h[2]].claim, (uint32_t)t); }
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
            return (int)TR[ip].claim;
        };
        int g = g_lo;
        int s = g / nsc, k = g - s * nsc;
        int t = g < g_hi ? near_tri(s, k) : 0;
        uint32_t i0 = 0, i1 = 0, i2 = 0, n0 = TNONE, n1 = TNONE, n2 = TNONE;
        int o0 = 0, o1 = 0, o2 = 0, d0 = 0, d1 = 0, d2 = 0;
        uint64_t id_lo = 0, id_hi = 0;                            // compact plan: the last eight triangle ids, waiting for their 16-B store
        bool fresh = true;                                        // triangle t not loaded yet
        int since = 0;                                            // crossings since the last emitted element
        while (g < g_hi) {
            if (fresh) {
                const TriU Tc = ld_tri(TR, (uint32_t)t);          // vertices and neighbours in one read: a crossing needs no second one
                i0 = Tc.v[0]; i1 = Tc.v[1]; i2 = Tc.v[2]; n0 = Tc.n[0]; n1 = Tc.n[1]; n2 = Tc.n[2];
                const uint32_t a = pts[i0], b = pts[i1], c = pts[i2], q = ((uint32_t)s << 16) | (uint32_t)k;
                o0 = orient2(b, c, q); o1 = orient2(c, a, q); o2 = orient2(a, b, q);
                d0 = px_of(c) - px_of(b); d1 = px_of(a) - px_of(c); d2 = px_of(b) - px_of(a);   // d(orient(u, v, q)) / d(column of q) = v.x - u.x
                fresh = false;
            }
            uint32_t nx = TNONE;
            bool cross = false;
            if (o0 < 0) { nx = n0; cross = true; }
            else if (o1 < 0) { nx = n1; cross = true; }
            else if (o2 < 0) { nx = n2; cross = true; }
#ifdef CHEST_STAMPS
            if (cross && nx != TNONE) ++dbg_cross;
#endif
            if (cross && nx != TNONE && since <= ntri + 8) { t = (int)nx; fresh = true; ++since; continue; }
            PlanRec rec{0.f, 0.f, 0u, 0u};                        // leaving through a hull edge: outside the convex hull
            uint32_t t_hit = t_outside;
            if (!cross) {
                t_hit = (uint32_t)t;
                if (!compact) {                                   // (a compact plan leaves the weights to its reader)
                    const float den = (float)(o0 + o1 + o2);
                    rec.w0 = (float)o0 / den; rec.w1 = (float)o1 / den;
                    rec.i01 = i0 | (i1 << 16); rec.i2v = i2 | (1u << 16);
                }
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
                        t_hit = (uint32_t)u;
                        break;
                    }
                }
            }
            if (!compact) plan[g] = rec;
            else {
                // ids leave eight at a time (one aligned 16-B store per group of elements 8 j ... 8 j + 7): they are shifted through a
                // 128-bit register, newest on top; a group cut by the ends of the run goes out as single u16 stores
                id_lo = (id_lo >> 16) | (id_hi << 48);
                id_hi = (id_hi >> 16) | ((uint64_t)t_hit << 48);
                if ((g & 7) == 7 || g + 1 == g_hi) {
                    const int n_in = min(g - g_lo + 1, (g & 7) + 1);              // elements of the group that are mine: the newest n_in
                    if (n_in == 8) *(uint4*)&triof[g - 7] = make_uint4((uint32_t)id_lo, (uint32_t)(id_lo >> 32), (uint32_t)id_hi, (uint32_t)(id_hi >> 32));
                    else
                        for (int j = 0; j < n_in; ++j) {                             // element g - j sits in 16-bit slot 7 - j
                            const int sl = 7 - j;
                            triof[g - j] = (uint16_t)(sl >= 4 ? (id_hi >> (16 * (sl - 4))) : (id_lo >> (16 * sl)));
                        }
                }
            }
            ++g; since = 0;
            if (++k == nsc) {                                     // next row: far from here, start again from the layout guess
                k = 0; ++s;
                if (g < g_hi) t = near_tri(s, k);
                fresh = true;
            } else { o0 += d0; o1 += d1; o2 += d2; }
        }
    } else {
        for (int g = g_lo; g < g_hi; ++g) { if (compact) triof[g] = (uint16_t)t_outside; else plan[g] = PlanRec{0.f, 0.f, 0u, 0u}; }
    }
    PSTAMP(6);                                                   // plan walk
#ifdef CHEST_STAMPS
    if (g_pat_stamp) {                                           // slot 7: flip rounds | largest crossing count of a lane of wave 0 << 32
        for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(dbg_cross, off, 64); dbg_cross = o > dbg_cross ? o : dbg_cross; }
        if (tid == 0) {
            unsigned long long* d_ = g_pat_stamp + (size_t)blockIdx.x * 16;
            d_[7] = (unsigned long long)dbg_rounds | ((unsigned long long)dbg_cross << 32);
            for (int i_ = 0; i_ < 4; ++i_) d_[8 + i_] = dbg_ph[i_];
            for (int i_ = 0; i_ < 4; ++i_) d_[12 + i_] = dbg_p0[i_];
            d_[7] |= (unsigned long long)dbg_solo_rounds << 56;
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
    if (method < -1 || method > 2) { set_error("chest_patterns_build: method must be -1 (no plan), 0 ('linear'), 1 ('nearest') or 2 ('linear', compact plan)"); return CHEST_EINVAL; }
    if (method == 2 && (ctx->fp.S > 32 || ctx->fp.nsc > 2047)) { set_error("chest_patterns_build: compact plans pack positions as 5 + 11 bits"); return CHEST_EINVAL; }
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
    if (2 * Pmax > 32 * PB_THREADS_DEFAULT) { set_error("chest_patterns_build: at most %d pilots per pattern", 16 * PB_THREADS_DEFAULT); return CHEST_EINVAL; }
    const size_t lds = pattern_build_lds_bytes(G, Pmax, method == 1 ? PB_THREADS_NEAREST : PB_THREADS_DEFAULT);
    if (lds > 160 * 1024) {
        set_error("chest_patterns_build: %d pilots need %zu B of LDS (> 160 KiB); use chest_pattern_set_triangulation for such patterns", Pmax, lds);
        return CHEST_EINVAL;
    }
    auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    // pool reuse: the slot range is served by one pool whose blocks are large enough -> build in place (stream ordered)
    PatternPool* pool = nullptr;
    for (auto& pl : ctx->pools)
        if (pl.first_slot == first_slot && pl.count >= count && pl.Pcap >= Pmax && pl.compact == (method == 2)) pool = &pl;
    if (pool == nullptr) {
        CHEST_HIP(hipDeviceSynchronize());
        // release whatever owned these slots before
        for (auto it = ctx->pools.begin(); it != ctx->pools.end();) {
            const bool overlap = it->first_slot < first_slot + count && first_slot < it->first_slot + it->count;
            if (overlap) {
                // every slot of the old pool loses its storage: host entries AND the device table entries (a descriptor naming a
                // released slot must be flagged CHEST_FRAME_BAD_DESC, not read freed memory); slots that were re-set through
                // chest_pattern_set since own their block and stay
                const int s_end = std::min(it->first_slot + it->count, (int)ctx->patterns.size());
                for (int s = it->first_slot; s < s_end;) {
                    if (ctx->patterns[s].block != nullptr) { ++s; continue; }
                    int e = s;                                   // run of released slots [s, e)
                    while (e < s_end && ctx->patterns[e].block == nullptr) ctx->patterns[e++] = PatternHost{};
                    const int dev_e = std::min(e, ctx->pattern_cap);
                    if (s < dev_e) CHEST_HIP(hipMemset(ctx->d_patterns + s, 0, (size_t)(dev_e - s) * sizeof(PatternDev)));
                    s = e;
                }
                (void)hipFree(it->block); (void)hipFree(it->d_meta);
                it = ctx->pools.erase(it);
            } else ++it;
        }
        if ((size_t)(first_slot + count) > ctx->patterns.size()) ctx->patterns.resize(first_slot + count);
        for (int s = first_slot; s < first_slot + count; ++s)
            if (ctx->patterns[s].block) { (void)hipFree(ctx->patterns[s].block); ctx->patterns[s] = PatternHost{}; }
        PatternPool pl{};
        pl.first_slot = first_slot; pl.count = count; pl.Pcap = Pmax;
        pl.o_pil = up((size_t)pattern_bit_words(G) * sizeof(uint2), 16);
        pl.o_plan = up(pl.o_pil + (size_t)Pmax * 4, 256);
        pl.compact = method == 2;
        if (pl.compact) {                                       // triangle records [2 Pmax] | u16 triangle id per element
            pl.o_tri = up(pl.o_plan + (size_t)2 * Pmax * sizeof(PlanRec), 256);
            pl.stride = up(pl.o_tri + (size_t)G * 2, 256);
        } else {
            pl.o_tri = 0;
            pl.stride = up(pl.o_plan + (size_t)G * sizeof(PlanRec), 256);
        }
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
    a.table = ctx->d_patterns; a.pool = (unsigned char*)pool->block; a.stride = pool->stride; a.o_pil = pool->o_pil; a.o_plan = pool->o_plan; a.o_tri = pool->o_tri;
    a.num_pilots = d_np; a.given_idx = pilot_idx; a.pattern_ids = d_ids; a.seed = seed; a.idx_stride = idx_stride;
    a.first_slot = first_slot; a.S = ctx->fp.S; a.nsc = ctx->fp.nsc; a.method = method; a.Pcap = pool->Pcap;
    a.simp_out = simplices_out; a.nbr_out = simplices_out ? neighbors_out : nullptr; a.nsimp_out = num_simplices_out; a.status = ctx->d_status;
    a.flip_cap = ctx->tune_pattern_flip_cap; a.stop_after = ctx->tune_pattern_stop_after; a.solo_at = ctx->tune_pattern_solo_at;
    if (simplices_out && pool->Pcap != Pmax) { set_error("chest_patterns_build: simplices_out needs a fresh slot range (row stride is 2*Pmax*3)"); return CHEST_EINVAL; }
    // 'nearest' plans have no flips and no one-wave stage: their workgroups take eight waves (2.59 -> 1.90 ms per 10 000 patterns at
    // P = 838; the Delaunay path measures 2.05 against 1.89 ms with eight: it keeps four)
    const int nthr = method == 1 ? PB_THREADS_NEAREST : PB_THREADS_DEFAULT;
    const size_t lds_run = pattern_build_lds_bytes(G, pool->Pcap, nthr) + (size_t)ctx->tune_pattern_extra_lds;
    const int sp = prof_open(ctx, st, CHEST_K_PLAN);
    if (method == 2) {
        CHEST_LDS_OPT_IN(ctx, LDS_DELAUNAY_CP, (pattern_build_kernel<true, PB_THREADS_DEFAULT>), lds_run);
        hipLaunchKernelGGL((pattern_build_kernel<true, PB_THREADS_DEFAULT>), dim3((unsigned)count), dim3(nthr), lds_run, st, a);
    } else if (method == 1) {
        CHEST_LDS_OPT_IN(ctx, LDS_DELAUNAY_NEAREST, (pattern_build_kernel<false, PB_THREADS_NEAREST>), lds_run);
        hipLaunchKernelGGL((pattern_build_kernel<false, PB_THREADS_NEAREST>), dim3((unsigned)count), dim3(nthr), lds_run, st, a);
    } else {
        CHEST_LDS_OPT_IN(ctx, LDS_DELAUNAY, (pattern_build_kernel<false, PB_THREADS_DEFAULT>), lds_run);
        hipLaunchKernelGGL((pattern_build_kernel<false, PB_THREADS_DEFAULT>), dim3((unsigned)count), dim3(nthr), lds_run, st, a);
    }
    prof_close(ctx, st, sp);
    CHEST_HIP(hipGetLastError());
    // host mirror of the table entries
    for (int i = 0; i < count; ++i) {
        PatternHost& ph = ctx->patterns[first_slot + i];
        unsigned char* base = (unsigned char*)pool->block + (size_t)i * pool->stride;
        if (ph.block != nullptr) {                               // set through chest_pattern_set since the pool was made: the pool takes
            CHEST_HIP(hipStreamSynchronize(st));                 // the slot back (frames already launched may still read the old block)
            (void)hipFree(ph.block);
        }
        ph.block = nullptr;                                      // owned by the pool
        ph.dev.bits = (const uint2*)base;
        ph.dev.plan = method >= 0 ? (const void*)(base + pool->o_plan) : nullptr;
        ph.dev.pad = method == 2 ? (int32_t)(pool->o_tri - pool->o_plan) : 0;
        ph.dev.pilot_idx = (const int32_t*)(base + pool->o_pil);
        ph.dev.P = num_pilots[i];
    }
    return CHEST_OK;
}

/* pilot list of a slot back to the host (sorted flat indices); synchronises on the device */
extern "C" int chest_pattern_get_pilots(chest_ctx* ctx, int32_t slot, int32_t* pilot_idx_out) {
    if (!ctx || !pilot_idx_out || slot < 0 || (size_t)slot >= ctx->patterns.size() || ctx->patterns[slot].dev.bits == nullptr) {
        set_error("pattern slot %d is not set", slot);
        return CHEST_ESTATE;
    }
    CHEST_GUARD(ctx);
    CHEST_HIP(hipDeviceSynchronize());
    CHEST_HIP(hipMemcpy(pilot_idx_out, ctx->patterns[slot].dev.pilot_idx, (size_t)ctx->patterns[slot].dev.P * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CHEST_OK;
}
