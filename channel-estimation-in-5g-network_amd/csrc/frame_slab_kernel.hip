// K2 of the fused frame pipeline (see frame_kernel.hip for K1 and the reference lines replaced).
//
// Write path: everything a frame produces is, per tensor, ONE contiguous span of HBM ([S][T][nsc], [S][R][nsc],
// [S][R][T][nsc] blocks of the reference's layouts), so the kernel treats its outputs as flat byte ranges:
//
//   * thread <-> subcarrier pair, FIXED for the whole frame (thread i owns k = 2i, 2i+1; ceil(nsc/128) waves per
//     workgroup).  The tap twiddles exp(-2 pi j kappa_k d / N) of a thread therefore never change: they live in
//     registers (no LDS table), and every wave of the workgroup works on the same (symbol, rx, tx) row at a time, so the
//     tap gains stay wave-uniform scalar loads.
//   * one "unit" = (OFDM symbol, group of RG rx antennas).  Its rows -- tx[s] (T rows), H[s][r0..] (RG*T rows),
//     rx[s][r0..] (RG rows); in the estimate phase H_ls and H_mmse [s][r0..] (RG*T rows each, tx-replicated, quirk Q9) --
//     are assembled in an LDS slab whose segments mirror the global spans byte for byte (same offset mod 16).
//   * store-out: the workgroup copies each segment to HBM as consecutive, 128-B-line-aligned, 1 KiB wave stores
//     (ds_read_b128 -> global_store_dwordx4 nt): every 128-B line inside a segment is written whole by one instruction,
//     only a segment's first and last line can be partial, and a unit's ~10-40 KB leave the CU as three bursts of
//     consecutive kilobytes instead of eight interleaved 1 008-B row streams per wave.
//
// Random streams, arithmetic and results are those of the previous kernel (chest_device.h; oracle/philox.py).
#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

// Output tensors are written once and never read back: nontemporal stores (tools/archive/micro/store_flavour.hip).
__device__ __forceinline__ void st16(float2* p, float4 v) {
    const v4f_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (v4f_t*)p);
}
__device__ __forceinline__ void st8(float2* p, float2 a) {
    const v2f_t t = {a.x, a.y};
    __builtin_nontemporal_store(t, (v2f_t*)p);
}

// Copy LDS segment l[0..n) to global g[0..n) (float2 elements; g and l wave-uniform with equal address mod 16).
// Store q covers the 1 KiB window of 128-B lines [q*128 - head, q*128 - head + 128) in element units, head = elements
// between the enclosing line start and g.  Stores q0, q0 + qstep, ... are this wave's (0 <= q0 < qstep).  Returns the
// number of stores of the segment.
__device__ __forceinline__ void seg_store_edge(float2* __restrict__ g, const float2* __restrict__ l, int n, int e) {
    if (e >= 0 && e + 1 < n) st16(g + e, *(const float4*)(l + e));
    else if (e == -1) st8(g, l[0]);                           // odd head: element 0 sits alone in its 16-B slot
    else if (e >= 0 && e == n - 1) st8(g + e, l[e]);
}
// The lane number is recomputed here (two instructions, opaque to the optimiser) instead of being passed in: under the
// 96-register budget the compiler otherwise keeps it in SCRATCH across the compute stretch, and a scratch reload is a vector
// memory operation whose s_waitcnt vmcnt(0) also waits for every store the wave still has in flight -- the store-out of one
// segment would drain before the next one starts.
__device__ __forceinline__ int lane_id_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ int seg_store(float2* __restrict__ g, const float2* __restrict__ l, int n, int q0, int qstep) {
    const int lane = lane_id_fresh();
    const int head = (int)(((uintptr_t)g & 127u) >> 3);
    const int nst = (head + n + 127) >> 7;
    const int off = 2 * lane - head;
    if (q0 == 0) seg_store_edge(g, l, n, off);                                    // first window: may start before g
    if (nst > 1 && (nst - 1 - q0) % qstep == 0 && nst - 1 >= q0) seg_store_edge(g, l, n, ((nst - 1) << 7) + off);   // last window: may end after g + n
    // interior windows are whole: no per-lane conditions; two in flight per iteration (both LDS reads before the stores)
    int q = q0 == 0 ? qstep : q0;
    for (; q + qstep < nst - 1; q += 2 * qstep) {
        const int e0 = (q << 7) + off, e1 = ((q + qstep) << 7) + off;
        const float4 a = *(const float4*)(l + e0), b = *(const float4*)(l + e1);
        st16(g + e0, a);
        st16(g + e1, b);
    }
    if (q < nst - 1) { const int e0 = (q << 7) + off; st16(g + e0, *(const float4*)(l + e0)); }
    return nst;
}

// Diagnostic build only (-DCHEST_STAMPS, tools/stamps.py): per-wave cycle sums of the unit's segments, written to a buffer of
// their own; no stamp executes in the shipped library.
#ifdef CHEST_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;       // [B][16 waves][16 slots]
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DECL unsigned long long st_a = 0, st_b = 0, st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_ADD(i) do { STAMP(st_b); st_sum[i] += st_b - st_a; st_a = st_b; } while (0)
#else
#define STAMP(t) do { } while (0)
#define STAMP_DECL do { } while (0)
#define STAMP_ADD(i) do { } while (0)
#endif

typedef float f16v_t __attribute__((ext_vector_type(16)));

struct SlabArgs {
    const FrameParams* fp;
    const PatternDev* pat;
    const TapProfile* prof;
    const float* rec;
    const float2* c_g;
    const float2* csum_g;
    const chest_injected* inj;
    const chest_frame_outputs* out;
    float2* hp;        // LDS [Pmax*R] LS estimates at the pilots; nullptr: no estimates in this launch -- the pilot values go straight
                       // to out.h_ls_pilots (if asked for) and the slab has the LDS to itself
    float* red;
    float* alpha;
    float2* slab;      // LDS, 16-B aligned
    RngKey key;
    int slab2;         // float2 elements from `slab` to a SECOND slab (0: none): synthesis units alternate between the two (see slab_body)
    int b, RG, nst, nsx;    // nsx: dedicated store waves at the end of the workgroup; nst: store-out crew override (chest_tune "frame_store_waves", experiments), 0 = by the rule in slab_body
};

// LDS slab segment capacities in float2 elements (even, +2 so that a parity shift of one element fits)
__host__ __device__ inline int slab_seg(int rows, int nsc) { return ((rows * nsc + 1) & ~1) + 2; }
// rows of the tx segment: every tx antenna sends the same grid (quirk Q5), and rows two apart start at the same 16-B phase
// (2 nsc elements = 16 nsc bytes), so two rows serve any number of antennas -- the store-out writes them T / 2 times
__host__ __device__ inline int tx_rows(int T, int SS) { return (SS == 1 && T > 2) ? 2 : SS * T; }
__host__ __device__ inline int slab_elems(int T, int RG, int nsc, int SS, bool with_estimates = true) {
    const int w = slab_seg(tx_rows(T, SS), nsc) + slab_seg(SS * RG * T, nsc) + slab_seg(SS * RG, nsc);
    const int e = with_estimates ? 2 * slab_seg(SS * RG * T, nsc) : 0;
    return w > e ? w : e;
}

// TT / RR / RGT > 0: compile-time antenna counts and rx-group size.  Loops unroll; a unit's rows are accumulated in
// registers TAP-OUTER (one scalar load brings a tap's gains for all antenna pairs of the unit, so coefficient SGPRs are
// streamed instead of hoisted row by row), the odd subcarrier's twiddle is the even one's times the tap's uniform step
// exp(-2 pi j delay/N) (needs an even number of used bins below DC: no pair straddles it; the launcher checks), and both
// barriers of a unit sit back to back behind the long compute stretch.  0 = run-time values: full twiddle registers,
// rows written to the slab as they are produced.
// CP: the patterns carry compact plans (triangle records + per-element triangle ids, chest_device.h: plan_rec).
// MF (4x4, one rx antenna and one symbol per unit, device RNG): the tap sum H[r][t][k] = sum_d c_d[r][t] w_d(k) of a symbol -- a complex
// (16 x NT) x (NT x subcarriers) product -- runs on the MATRIX pipe as v_mfma_f32_32x32x2_f32 (full fp32, same peak as the packed vector
// FMAs it replaces, but a pipe of its own: the vector ALU keeps the Philox / Box-Muller / symbol work).  Rows of a tile are (rx, re | im, tx),
// columns 32 subcarriers, one K step per tap (its real and imaginary twiddle row); four tiles -- even / odd subcarriers of the wave's
// lower / upper 32 pairs -- and a v_permlane32_swap per result register bring every lane the complex values of its own pair for all
// sixteen antenna pairs.  Computed once per symbol (first rx unit), consumed by the symbol's four units from registers.
template <int NT, int TT, int RR, int RGT, int SS, bool INJ, bool CP, bool MF = false>
__device__ __forceinline__ void slab_body(const SlabArgs& A) {
    const FrameParams& fp = *A.fp;
    const PatternDev& pat = *A.pat;
    const chest_injected& inj = *A.inj;
    const chest_frame_outputs& out = *A.out;
    constexpr bool STAGED = TT > 0 && RR > 0 && RGT > 0;
    const int S = fp.S, G = fp.G, P = pat.P, b = A.b;
    const int T = TT > 0 ? TT : fp.T, R = RR > 0 ? RR : fp.R, RG = RGT > 0 ? RGT : A.RG;
    const int nsc = fp.nsc;
    const int tid = threadIdx.x, lane = tid & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // subcarrier pairs are dealt evenly to the waves (599 subcarriers: 60 pairs on each of 5 waves) so that no wave of
    // the workgroup is the straggler at the unit barriers
    // Dedicated store waves (A.nsx > 0): the LAST nsx waves of the workgroup own no subcarriers -- they take no part in the
    // synthesis and do nothing but stream finished slabs to HBM, while the compute waves go from the "slab complete" barrier
    // straight into the next unit's synthesis: the store-out of unit u runs beside the compute stretch of unit u + 1 for EVERY
    // compute wave (the excused-wave scheme below gives that head start to one or two waves only).
    const int nsx = A.nsx;
    const int ncw = nwave - nsx;                                 // compute waves
    const bool cw = wave < ncw;                                  // wave-uniform
    const int npair = (nsc + 1) >> 1;
    const int ppw = (npair + ncw - 1) / ncw;
    const int pr = wave * ppw + lane;
    const bool act = cw && lane < ppw && pr < npair;
    const int k0 = 2 * pr, k1 = k0 + 1;
    const bool v1 = act && k1 < nsc;
    const bool tailw = __builtin_amdgcn_readfirstlane((int)((nsc & 1) && wave == (npair - 1) / ppw));   // the wave that owns the odd last subcarrier
    const int kc0 = act ? k0 : nsc - 1, kc1 = v1 ? k1 : nsc - 1;
    const float noise_std = A.rec[ST_NOISE_STD];
    float2* __restrict__ hp = A.hp;
    float2* __restrict__ slab = A.slab;
    // hp == nullptr: pilot estimates leave as 8-B stores into this frame's [R][stride] block (a symbol's pilots have consecutive
    // ordinals, so the lanes of a wave that hold pilots write neighbouring addresses); no destination: nothing is kept
    float2* __restrict__ hl_direct = (hp == nullptr && out.h_ls_pilots != nullptr) ? (float2*)out.h_ls_pilots + (size_t)b * R * out.h_ls_stride : nullptr;
    const bool keep_pilots = hp != nullptr || hl_direct != nullptr;
    auto put_pilot = [&](unsigned ordinal, int r, float2 v) {
        if (hp != nullptr) hp[ordinal * R + r] = v;
        else hl_direct[(size_t)r * out.h_ls_stride + ordinal] = v;
    };
    float2* __restrict__ txb = out.tx ? (float2*)out.tx + (size_t)b * S * T * nsc : nullptr;
    float2* __restrict__ rxb = out.rx ? (float2*)out.rx + (size_t)b * S * R * nsc : nullptr;
    float2* __restrict__ Hb = out.H ? (float2*)out.H + (size_t)b * S * R * T * nsc : nullptr;
    // INJ == false: the device-RNG instantiation carries no injected-input pointers or branches at all
    const float* __restrict__ usym = (INJ && inj.u_sym) ? inj.u_sym + (size_t)b * G : nullptr;
    const float* __restrict__ nre = (INJ && inj.n_re) ? inj.n_re + (size_t)b * S * R * nsc : nullptr;
    const float* __restrict__ nim = (INJ && inj.n_im) ? inj.n_im + (size_t)b * S * R * nsc : nullptr;
    const gptr_bits pbits = (gptr_bits)pat.bits;
    // parity (element offset & 1) of the frame's first element in each tensor; 16-B-aligned tensor bases (host-checked)
    const int par_tx = (int)(((size_t)b * S * T * nsc) & 1), par_rx = (int)(((size_t)b * S * R * nsc) & 1),
              par_h = (int)(((size_t)b * S * R * T * nsc) & 1);
    // one row of a slab segment: both subcarriers of the pair; only the wave holding the odd last subcarrier needs a
    // per-lane condition (callers are inside `if (act)`)
    auto put = [&](float2* __restrict__ row, float2 a, float2 b2) {
        row[k0] = a;
        if (!tailw) row[k1] = b2;
        else if (v1) row[k1] = b2;
    };

    // ---- tap twiddles exp(-2 pi j kappa_k d / N) of this thread's subcarriers: registers for the whole frame ---------
    constexpr int NW = STAGED ? 1 : 2;                           // STAGED keeps the even subcarrier's only
    static_assert(!MF || (STAGED && TT == 4 && RR == 4 && RGT == 1 && SS == 1 && !INJ), "MF: 4x4, one rx antenna and one symbol per unit");
    float2 w[MF ? 1 : NT][NW];
    const float invN_tw = 1.0f / (float)fp.N;
    auto bin_of = [&](int kc) -> int { return (kc < fp.half) ? (kc - fp.half) : (kc - fp.half + 1); };   // signed bin of a used subcarrier (DC removed)
    // MF: B operands of the even-subcarrier tiles, [tap][lower | upper 32 pairs of the wave]: lanes 0-31 hold Re w, lanes 32-63 Im w of
    // subcarrier 2 (wave pair base + (lane & 31) + 32 x)
    float bq[MF ? NT : 1][2];
    if constexpr (MF) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int kx = min(2 * (wave * ppw + (lane & 31) + 32 * x), nsc - 1);
            const int kax = bin_of(kx);
#pragma unroll
            for (int d = 0; d < NT; ++d) {
                const float2 e = cis_turns((float)(kax * A.prof->delay[d]) * invN_tw);
                bq[d][x] = lane < 32 ? e.x : -e.y;               // w = conj(exp(2 pi j kappa d / N))
            }
        }
    }
    if constexpr (!MF) {
        const float invN = 1.0f / (float)fp.N;
        const int half = fp.half;
        const int ka = (kc0 < half) ? (kc0 - half) : (kc0 - half + 1), kb = (kc1 < half) ? (kc1 - half) : (kc1 - half + 1);
#pragma unroll
        for (int d = 0; d < NT; ++d) {
            const int dl = A.prof->delay[d];
            const float2 e0 = cis_turns((float)(ka * dl) * invN);                // |kappa*delay| < 2^24: exact
            w[d][0] = make_float2(e0.x, -e0.y);
            if constexpr (!STAGED) { const float2 e1 = cis_turns((float)(kb * dl) * invN); w[d][NW - 1] = make_float2(e1.x, -e1.y); }
        }
    }
    auto w_even = [&](int d) -> float2 {                         // twiddle of the pair's even subcarrier
        if constexpr (MF) { const float2 e = cis_turns((float)(bin_of(kc0) * A.prof->delay[d]) * invN_tw); return make_float2(e.x, -e.y); }   // (estimate phase only)
        else return w[d][0];
    };
    auto w_odd = [&](int d) -> float2 {                          // twiddle of the pair's odd subcarrier
        if constexpr (STAGED) {
            float2 we = w_even(d);
            asm volatile("" : "+v"(we.x), "+v"(we.y));           // keeps the product inside the loop: hoisted, it would cost the
            return cmul(we, make_float2(A.prof->step[d][0], A.prof->step[d][1]));   // 2*NT registers this form exists to save
        } else return w[MF ? 0 : d][NW - 1];
    };

    // SSN consecutive symbols form one unit (STAGED instantiations with RG == R only: a unit's spans stay contiguous across
    // symbols); 1 everywhere else.
    constexpr int SSN = STAGED ? SS : 1;
    const int txr = tx_rows(T, SSN);                             // rows of tx kept in the slab (2 when T > 2: written T / 2 times)
    const int seg_tx = 0, seg_h = slab_seg(txr, nsc), seg_rx = seg_h + slab_seg(SSN * RG * T, nsc);
    const int ngroups = R / RG;
    bool first_unit = true;
    // Two slabs (launches whose LDS has room: the 2048-FFT 4x4 frames without estimates, 2 x 67 KB) with dedicated store waves: unit
    // u + 1 is synthesised and written into the OTHER slab while the store waves stream unit u -- the barrier in front of the slab
    // writes ("the previous store-out has read the slab") goes away, one barrier per unit is left, and the workgroup's store stream
    // no longer pauses for the synthesis (one workgroup per CU: nothing else on the CU fills that pause).
    const bool two_slabs = A.slab2 != 0 && nsx > 0;
    int unit_no = 0;
    // Store-out crew.  A workgroup of 4n+1 waves puts two of its waves on one SIMD; the younger of them loses the issue
    // arbitration and is the straggler of every compute stretch (stamps: +40 % on the 5th wave).  It is excused from the
    // store-out: after the "slab complete" barrier it goes straight to the next unit's compute while the others stream the
    // slab, which hands it a head start the length of the store-out.  4n + 2 waves -- the 10-wave workgroups of the 2048-FFT
    // grids -- leave TWO SIMDs with an extra wave: both are excused.  Measured per crew size (tools/ab_store_waves.py,
    // profiles/r04_store_crew.txt): 2048-FFT 4x4 2.09 ms with all ten waves storing, 2.01-2.04 with four to eight; 2x2 2.07 -> 1.98;
    // SISO (two symbols per unit since round 2) 0.83 -> 0.76: the rule holds for every staged shape.  Units of one or two rows
    // (the generic instantiations at T = R = 1) keep every wave on the stores.
    const int xw = nwave & 3;
    // (ten waves and one workgroup per CU: the store-out of a unit has nothing else to hide behind; four storing waves, six in the
    // next unit's synthesis measure 1-4 % below eight / two)
    const int nst_rule = (nwave > 4 && (xw == 1 || xw == 2) && RG * T * SSN >= 2) ? (nwave >= 10 ? 4 : nwave - xw) : nwave;
    const int nstw = nsx > 0 ? nsx : (A.nst > 0 ? min(A.nst, nwave) : nst_rule);
    const bool storer = nsx > 0 ? !cw : wave < nstw;
    const int swave = nsx > 0 ? wave - ncw : wave;               // my index in the store crew
    STAMP_DECL;
    STAMP(st_a);

    // ---- phase W: per unit (symbols, rx group) assemble tx / H / rx rows in the slab, then stream them out -----------
    // Slot entries (which resource elements are pilots, and their ordinals).  Injected draws need them at the top of a unit,
    // so that path fetches them one unit ahead; the device-RNG path needs them only when the unit's rows go into the slab:
    // it issues the loads at the top of the unit and lets the synthesis cover them.  (A value prefetched across the store-out
    // is consumed behind an s_waitcnt vmcnt(0) that also waits for every store of that store-out.)
    uint32_t sln[SSN][2];
    if constexpr (INJ) {
#pragma unroll
        for (int ss = 0; ss < SSN; ++ss) {
            const int sy = ss < S ? ss : S - 1;
            sln[ss][0] = draw_slot(pbits, sy * nsc + kc0, P); sln[ss][1] = draw_slot(pbits, sy * nsc + kc1, P);
        }
    }
    f16v_t hacc[MF ? 4 : 1];                                      // MF: a symbol's H, all antenna pairs, this lane's subcarrier pair
    auto mf_burst = [&](int s) {
        if constexpr (MF) {
            // A operand of tap d: lane (m = lane & 31, k = lane >> 5) holds A[m][k], row m = (rx = m >> 3, part = (m >> 2) & 1, tx = m & 3):
            // part re: (Re c, -Im c), part im: (Im c, Re c) against the twiddle rows (Re w, Im w)
            const int m = lane & 31, kk = lane >> 5, part = (m >> 2) & 1;
            const float* __restrict__ ca = (const float*)(A.c_g + (size_t)s * KTS * R * T) + (((m >> 3) * T + (m & 3)) * 2 + (part ^ kk));
            const uint32_t negm = (part == 0 && kk == 1) ? 0x80000000u : 0u;
            float av[NT];
#pragma unroll
            for (int d = 0; d < NT; ++d) av[d] = __uint_as_float(__float_as_uint(ca[d * R * T * 2]) ^ negm);
#pragma unroll
            for (int d = 0; d < NT; ++d) {
                const float sr = A.prof->step[d][0], si = A.prof->step[d][1];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const float own = bq[d][x];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(own), __float_as_uint(own), false, false);
                    const float other = __uint_as_float(lane < 32 ? sw[1] : sw[0]);     // the same subcarrier's other part
                    const float odd = fmaf(own, sr, other * (lane < 32 ? -si : si));     // (w step): Re on lanes 0-31, Im on 32-63
                    if (d == 0) {
                        const f16v_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        hacc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], own, zero, 0, 0, 0);
                        hacc[2 + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], odd, zero, 0, 0, 0);
                    } else {
                        hacc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], own, hacc[x], 0, 0, 0);
                        hacc[2 + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], odd, hacc[2 + x], 0, 0, 0);
                    }
                }
            }
            // result register v of a tile: row (rx = v >> 2, part = lane >> 5, tx = v & 3), column lane & 31.  Swap the upper half of the
            // lower-pairs tile with the lower half of the upper-pairs tile: every lane then holds Re (first of the two tiles) and Im
            // (second) of ITS pair's subcarrier
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const auto e = __builtin_amdgcn_permlane32_swap(__float_as_uint(hacc[0][v]), __float_as_uint(hacc[1][v]), false, false);
                hacc[0][v] = __uint_as_float(e[0]); hacc[1][v] = __uint_as_float(e[1]);
                const auto o = __builtin_amdgcn_permlane32_swap(__float_as_uint(hacc[2][v]), __float_as_uint(hacc[3][v]), false, false);
                hacc[2][v] = __uint_as_float(o[0]); hacc[3][v] = __uint_as_float(o[1]);
            }
        }
    };
    for (int sb = 0; sb < S; sb += SSN) {
        const int ns = SSN == 1 ? 1 : min(SSN, S - sb);                          // symbols of this unit (ragged last unit)
        unsigned sc[SSN][2];
        float2 xs[SSN][2];
        bool pp[SSN][2];
#pragma unroll
        for (int ss = 0; ss < SSN; ++ss) { sc[ss][0] = sc[ss][1] = 0; xs[ss][0] = xs[ss][1] = make_float2(0.f, 0.f); pp[ss][0] = pp[ss][1] = false; }
        if (cw) {
#pragma unroll
        for (int ss = 0; ss < SSN; ++ss) {
            if constexpr (INJ) {
                sc[ss][0] = sln[ss][0]; sc[ss][1] = sln[ss][1];
                const int sn = sb + SSN + ss;
                if (sn < S) { sln[ss][0] = draw_slot(pbits, sn * nsc + kc0, P); sln[ss][1] = draw_slot(pbits, sn * nsc + kc1, P); }
            } else {
                const int sy = sb + ss < S ? sb + ss : S - 1;
                sc[ss][0] = draw_slot(pbits, sy * nsc + kc0, P); sc[ss][1] = draw_slot(pbits, sy * nsc + kc1, P);
            }
            float u0, u1;
            if (usym != nullptr) {
                u0 = usym[sc[ss][0]]; u1 = usym[sc[ss][1]];
            } else {
                const u32x4 z = draw(A.key, STREAM_SYM, (uint32_t)((sb + ss) * npair + pr));
                u0 = u01(z.x); u1 = u01(z.y);
            }
            xs[ss][0] = cis_unit(__builtin_amdgcn_fractf(u0));                   // exp(j 2pi u) (:395,:399)
            xs[ss][1] = cis_unit(__builtin_amdgcn_fractf(u1));
            pp[ss][0] = keep_pilots && act && ss < ns && sc[ss][0] < (unsigned)P; pp[ss][1] = keep_pilots && v1 && ss < ns && sc[ss][1] < (unsigned)P;
        }
        }
        for (int rg = 0; rg < ngroups; ++rg) {
            const int r0 = rg * RG;
            // global spans of this unit (element offsets inside the frame) and their parities
            const unsigned e_tx = (unsigned)(sb * T) * (unsigned)nsc, e_h = (unsigned)((sb * R + r0) * T) * (unsigned)nsc,
                           e_rx = (unsigned)(sb * R + r0) * (unsigned)nsc;
            float2* __restrict__ slab_u = slab + ((two_slabs && (unit_no & 1)) ? A.slab2 : 0);
            ++unit_no;
            float2* __restrict__ l_tx = slab_u + seg_tx + ((par_tx + e_tx) & 1);
            float2* __restrict__ l_h = slab_u + seg_h + ((par_h + e_h) & 1);
            float2* __restrict__ l_rx = slab_u + seg_rx + ((par_rx + e_rx) & 1);
            if constexpr (STAGED) {
                // compute the whole unit into registers first: nothing here touches the slab, so this stretch overlaps the
                // other waves' store-out of the previous unit
                float2 hv[SSN][RGT * TT][2], qv[SSN][RGT][2];
                if (cw) {
#pragma unroll
                for (int ss = 0; ss < SSN; ++ss) {
                    if (ss >= ns) continue;
                    const int s = sb + ss;
                    const float2* __restrict__ cs = A.c_g + (size_t)s * KTS * R * T;   // this symbol's gains [tap][r][t]
                    if constexpr (MF) {
                        if (rg == 0) mf_burst(s);                                 // the symbol's H for all four rx antennas, into registers
#define CHEST_MF_ROW(RX)                                                                                                           \
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                             \
                            hv[ss][t][0] = make_float2(hacc[0][4 * (RX) + t], hacc[1][4 * (RX) + t]);                               \
                            hv[ss][t][1] = make_float2(hacc[2][4 * (RX) + t], hacc[3][4 * (RX) + t]);                               \
                        }
                        switch (r0) {                                             // (uniform; static register indices in every arm)
                            case 0: CHEST_MF_ROW(0) break;
                            case 1: CHEST_MF_ROW(1) break;
                            case 2: CHEST_MF_ROW(2) break;
                            default: CHEST_MF_ROW(3) break;
                        }
#undef CHEST_MF_ROW
                    }
                    // the first tap of every profile sits at delay 0 (the launcher checks): its twiddle is 1, the accumulators start from its gains
                    if constexpr (!MF) {
                        const float2* __restrict__ cd = cs + r0 * T;
#pragma unroll
                        for (int i = 0; i < RGT * TT; ++i) hv[ss][i][0] = hv[ss][i][1] = cd[i];
                    }
#pragma unroll
                    for (int d = 1; d < (MF ? 0 : NT); ++d) {
                        const float2 wa = w[MF ? 0 : d][0], wb = w_odd(d);
                        const float2* __restrict__ cd = cs + (d * R + r0) * T;   // [rr][t] contiguous: one scalar load per tap
#pragma unroll
                        for (int i = 0; i < RGT * TT; ++i) {
                            const float2 c = cd[i];
                            cfma(hv[ss][i][0], c, wa);
                            cfma(hv[ss][i][1], c, wb);
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < RGT; ++rr) {
                        const int r = r0 + rr;
                        // sum_t H_t is accumulated from the per-tx rows, whether or not H is stored
                        float2 hs0 = make_float2(0.f, 0.f), hs1 = make_float2(0.f, 0.f);
#pragma unroll
                        for (int t = 0; t < TT; ++t) {
                            hs0.x += hv[ss][rr * TT + t][0].x; hs0.y += hv[ss][rr * TT + t][0].y;
                            hs1.x += hv[ss][rr * TT + t][1].x; hs1.y += hv[ss][rr * TT + t][1].y;
                        }
                        const float2 y0 = cmul(hs0, xs[ss][0]), y1 = cmul(hs1, xs[ss][1]);   // y = sum_t H x (:330-334)
                        float2 n0v, n1v;
                        if (nre != nullptr) {
                            const unsigned row = (unsigned)(s * R + r) * (unsigned)nsc;
                            n0v = make_float2((nre + row)[kc0], (nim + row)[kc0]);
                            n1v = make_float2((nre + row)[kc1], (nim + row)[kc1]);
                        } else {
                            const u32x4 z = draw(A.key, STREAM_NOISE, (uint32_t)((s * R + r) * npair + pr));
                            n0v = gauss_pair(z.x, z.y);
                            n1v = gauss_pair(z.z, z.w);
                        }
                        qv[ss][rr][0] = make_float2(fmaf(noise_std, n0v.x, y0.x), fmaf(noise_std, n0v.y, y0.y));
                        qv[ss][rr][1] = make_float2(fmaf(noise_std, n1v.x, y1.x), fmaf(noise_std, n1v.y, y1.y));
                    }
                }
                }
                STAMP_ADD(0);                                                    // W compute
                if (!first_unit && !two_slabs) __syncthreads();                  // previous unit's store-out has read the slab
                first_unit = false;
                STAMP_ADD(1);                                                    // W barrier B
                if (act) {
#pragma unroll
                    for (int ss = 0; ss < SSN; ++ss) {
                        if (ss >= ns) continue;
                        if (rg == 0 && txb != nullptr) {
#pragma unroll
                            for (int t = 0; t < (SSN == 1 && TT > 2 ? 2 : TT); ++t) put(l_tx + (ss * TT + t) * nsc, xs[ss][0], xs[ss][1]);
                        }
#pragma unroll
                        for (int rr = 0; rr < RGT; ++rr) {
                            const int r = r0 + rr;
                            if (Hb != nullptr) {
#pragma unroll
                                for (int t = 0; t < TT; ++t) put(l_h + ((ss * RGT + rr) * TT + t) * nsc, hv[ss][rr * TT + t][0], hv[ss][rr * TT + t][1]);
                            }
                            if (rxb != nullptr) put(l_rx + (ss * RGT + rr) * nsc, qv[ss][rr][0], qv[ss][rr][1]);
                            if (pp[ss][0]) put_pilot(sc[ss][0], r, pilot_divide(qv[ss][rr][0], xs[ss][0]));    // (src/baseline_estimators.py:110)
                            if (pp[ss][1]) put_pilot(sc[ss][1], r, pilot_divide(qv[ss][rr][1], xs[ss][1]));
                        }
                    }
                }
            } else {
                const int s = sb;
                const float2 x0 = xs[0][0], x1 = xs[0][1];
                const unsigned s0 = sc[0][0], s1 = sc[0][1];
                const bool p0 = pp[0][0], p1 = pp[0][1];
                const float2* __restrict__ cs = A.c_g + (size_t)s * KTS * R * T;     // this symbol's gains [tap][r][t]
                if (!first_unit && !two_slabs) __syncthreads();                  // previous unit's store-out has read the slab
                first_unit = false;
                if (act && rg == 0 && txb != nullptr) {
                    for (int t = 0; t < txr; ++t) put(l_tx + t * nsc, x0, x1);
                }
                for (int rr = 0; cw && rr < RG; ++rr) {
                    const int r = r0 + rr;
                    float2 hs0 = make_float2(0.f, 0.f), hs1 = make_float2(0.f, 0.f);
                    for (int t = 0; t < T; ++t) {
                        const float2* __restrict__ coef = cs + r * T + t;        // tap d at stride R*T
                        float2 h0 = make_float2(0.f, 0.f), h1 = make_float2(0.f, 0.f);
#pragma unroll
                        for (int d = 0; d < NT; ++d) {
                            const float2 c = coef[d * R * T];
                            cfma(h0, c, w[d][0]);
                            cfma(h1, c, w[d][NW - 1]);
                        }
                        if (act && Hb != nullptr) put(l_h + (rr * T + t) * nsc, h0, h1);
                        hs0.x += h0.x; hs0.y += h0.y; hs1.x += h1.x; hs1.y += h1.y;
                    }
                    const float2 y0 = cmul(hs0, x0), y1 = cmul(hs1, x1);         // y = sum_t H x (:330-334)
                    float2 n0v, n1v;
                    if (nre != nullptr) {
                        const unsigned row = (unsigned)(s * R + r) * (unsigned)nsc;
                        n0v = make_float2((nre + row)[kc0], (nim + row)[kc0]);
                        n1v = make_float2((nre + row)[kc1], (nim + row)[kc1]);
                    } else {
                        const u32x4 z = draw(A.key, STREAM_NOISE, (uint32_t)((s * R + r) * npair + pr));
                        n0v = gauss_pair(z.x, z.y);
                        n1v = gauss_pair(z.z, z.w);
                    }
                    const float2 q0 = make_float2(fmaf(noise_std, n0v.x, y0.x), fmaf(noise_std, n0v.y, y0.y));
                    const float2 q1 = make_float2(fmaf(noise_std, n1v.x, y1.x), fmaf(noise_std, n1v.y, y1.y));
                    if (act && rxb != nullptr) put(l_rx + rr * nsc, q0, q1);
                    if (p0) put_pilot(s0, r, pilot_divide(q0, x0));              // (src/baseline_estimators.py:110)
                    if (p1) put_pilot(s1, r, pilot_divide(q1, x1));
                }
            }
            STAMP_ADD(2);                                                        // W slab writes
            __syncthreads();                                                     // slab complete
            STAMP_ADD(3);                                                        // W barrier A
            if (storer) {
                int ph = swave;                                                  // rotate the stores of consecutive segments over the waves
                if (rg == 0 && txb != nullptr) {
                    if (txr == ns * T) { const int n = seg_store(txb + e_tx, l_tx, ns * T * nsc, ph, nstw); ph = (ph + nstw - n % nstw) % nstw; }
                    else                                                         // two slab rows -> rows (0, 1), (2, 3), ... of the symbol
                        for (int t0 = 0; t0 < T; t0 += 2) {
                            const int n = seg_store(txb + e_tx + t0 * nsc, l_tx, min(2, T - t0) * nsc, ph, nstw); ph = (ph + nstw - n % nstw) % nstw;
                        }
                }
                if (Hb != nullptr) { const int n = seg_store(Hb + e_h, l_h, ns * RG * T * nsc, ph, nstw); ph = (ph + nstw - n % nstw) % nstw; }
                if (rxb != nullptr) seg_store(rxb + e_rx, l_rx, ns * RG * nsc, ph, nstw);
            }
            STAMP_ADD(4);                                                        // W store-out
        }
    }
    float2* __restrict__ Lb = out.H_ls ? (float2*)out.H_ls + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ Mb = out.H_mmse ? (float2*)out.H_mmse + (size_t)b * S * R * T * nsc : nullptr;
    const bool want_metrics = out.metrics != nullptr;
    const bool want_est = (Lb != nullptr) || (Mb != nullptr) || want_metrics;
    __syncthreads();                                                             // hp complete; last store-out has read the slab
    STAMP_ADD(5);                                                                // barrier after phase W

    // ---- pilot symbols in pilot order: regenerated from their own draws (identical arithmetic) -------------------------
    if (out.pilot_symbols != nullptr) {
        float2* __restrict__ psb = (float2*)out.pilot_symbols + (size_t)b * out.pilot_stride;
        if constexpr (CP) {
            // A pattern of its own per frame: no read of its pilot index list (3.4 KB from HBM per frame).  One word of the pilot bit
            // table per thread: its pilots have consecutive ordinals (.y, .y + 1, ...), the lanes of a wave write one ascending run.
            const int W = (G + 31) >> 5;
            for (int wi = tid; wi < W; wi += (int)blockDim.x) {
                const v2u_t wv = pbits[wi];
                uint32_t ord = wv.y;
                for (uint32_t mm = wv.x; mm != 0; mm &= mm - 1) {
                    const int g = (wi << 5) + (__ffs(mm) - 1);
                    const int sy = g / nsc, k = g - sy * nsc;
                    float u;
                    if (usym != nullptr) u = usym[ord];                          // first P draws are the pilot phases (:395)
                    else { const u32x4 z = draw(A.key, STREAM_SYM, (uint32_t)(sy * npair + (k >> 1))); u = u01((k & 1) ? z.y : z.x); }
                    st8(psb + ord, cis_unit(__builtin_amdgcn_fractf(u)));
                    ++ord;
                }
            }
        } else {
            const bool al = (((uintptr_t)psb) & 15) == 0;
            for (int i = 2 * tid; i < P; i += 2 * (int)blockDim.x) {             // shared patterns (tables in L2): linear 16-B stores
                float2 xs2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int p = i + j < P ? i + j : P - 1;
                    float u;
                    if (usym != nullptr) {
                        u = usym[p];                                             // first P draws are the pilot phases (:395)
                    } else {
                        const int g = ((gptr_i32)pat.pilot_idx)[p];
                        const int sy = g / nsc, k = g - sy * nsc;
                        const u32x4 z = draw(A.key, STREAM_SYM, (uint32_t)(sy * npair + (k >> 1)));
                        u = u01((k & 1) ? z.y : z.x);
                    }
                    xs2[j] = cis_unit(__builtin_amdgcn_fractf(u));
                }
                if (al && i + 1 < P) st16(psb + i, make_float4(xs2[0].x, xs2[0].y, xs2[1].x, xs2[1].y));
                else { st8(psb + i, xs2[0]); if (i + 1 < P) st8(psb + i + 1, xs2[1]); }
            }
        }
    }
    if (out.h_ls_pilots != nullptr) {                // LS estimates at the pilots, [R][stride] per frame, zero padded
        float2* __restrict__ hl = (float2*)out.h_ls_pilots + (size_t)b * R * out.h_ls_stride;
        if (hp != nullptr) {
            for (int i = tid; i < R * out.h_ls_stride; i += blockDim.x) {
                const int r = i / out.h_ls_stride, p = i - r * out.h_ls_stride;
                hl[i] = p < P ? hp[p * R + r] : make_float2(0.f, 0.f);
            }
        } else {                                                                 // the values are there already: only the zero padding
            const int npad = out.h_ls_stride - P;
            for (int i = tid; i < R * npad; i += blockDim.x) hl[(size_t)(i / npad) * out.h_ls_stride + P + i % npad] = make_float2(0.f, 0.f);
        }
    }
#ifdef CHEST_STAMPS
    if (!want_est && g_stamp_buf != nullptr && lane == 0) {      // (a launch without estimates leaves here: its stamps too)
        unsigned long long* dst = g_stamp_buf + ((size_t)b * 16 + wave) * 16;
        for (int i = 0; i < 12; ++i) dst[i] = st_sum[i];
    }
#endif
    if (!want_est) return;

    // ---- MMSE shrinkage per rx: p = mean|h_ls|^2, alpha = p/(p+sigma^2) (:177-190, scalar form) ---
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
        for (int i = tid; i < P; i += blockDim.x) a += cabs2(hp[i * R + r]);
        a = block_sum(a, A.red);
        if (tid == 0) {
            const double p = (double)a / (double)P;
            A.alpha[r] = (float)(p / (p + (double)A.rec[ST_SIGMA2]));
        }
    }
    __syncthreads();

    // ---- phase E: plan-driven interpolation (griddata linear/nearest as a 3-point gather), tx-replicated rows (quirk Q9)
    //      assembled in the slab and streamed out; optional error sums. -----------------------------------------------
    const gptr_plan plan = (gptr_plan)pat.plan;
    const float invT = 1.0f / (float)T;
    const int seg_l = 0, seg_m = slab_seg(SSN * RG * T, nsc);
    float m_ls = 0.f, m_mm = 0.f;
    PlanRec cn[SSN][2];                                                          // plan records, one unit ahead
    // compact plans: a record is found through the element's triangle id, so the ids run TWO units ahead of their use and the
    // records one unit ahead; cn then holds the raw record words, decoded where they are used (plan_rec)
    const gptr_u16 triof = (gptr_u16)((const CHEST_GLOBAL unsigned char*)pat.plan + (CP ? pat.pad : 0));
    uint32_t idn[SSN][2];
#pragma unroll
    for (int ss = 0; ss < SSN; ++ss) { cn[ss][0] = cn[ss][1] = PlanRec{0.f, 0.f, 0u, 0u}; idn[ss][0] = idn[ss][1] = 0; }
    if (cw) {
#pragma unroll
    for (int ss = 0; ss < SSN; ++ss) {
        const int sy = ss < S ? ss : S - 1;
        if constexpr (CP) {
            cn[ss][0] = ld_plan(plan, triof[sy * nsc + kc0]); cn[ss][1] = ld_plan(plan, triof[sy * nsc + kc1]);
            const int s2 = SSN + ss < S ? SSN + ss : S - 1;
            idn[ss][0] = triof[s2 * nsc + kc0]; idn[ss][1] = triof[s2 * nsc + kc1];
        } else {
            cn[ss][0] = ld_plan(plan, sy * nsc + kc0); cn[ss][1] = ld_plan(plan, sy * nsc + kc1);
            idn[ss][0] = idn[ss][1] = 0;
        }
    }
    }
    first_unit = true;
    // STAGED: a unit's estimates are gathered into registers one unit AHEAD, between the "slab complete" barrier of the unit
    // before and that unit's store-out -- the plan records they need were requested a barrier pair earlier and no store sits
    // between request and use (a value fetched across a store-out is consumed behind an s_waitcnt vmcnt(0) that drains the
    // wave's stores).
    constexpr int EVR = STAGED ? RGT : 1;
    float2 ev[SSN][EVR][2];
#pragma unroll
    for (int ss = 0; ss < SSN; ++ss)
#pragma unroll
        for (int rr = 0; rr < EVR; ++rr) ev[ss][rr][0] = ev[ss][rr][1] = make_float2(0.f, 0.f);
    auto gather = [&](const PlanRec (&c)[SSN][2], int s0, int r0, float2 (&e)[SSN][EVR][2]) {   // c: records of symbols s0, s0 + 1, ...
        if (!cw) return;                                                         // (store waves hold no elements)
#pragma unroll
        for (int ss = 0; ss < SSN; ++ss) {
            const PlanRec c0 = plan_rec<CP>(c[ss][0], s0 + ss, kc0), c1 = plan_rec<CP>(c[ss][1], s0 + ss, kc1);
#pragma unroll
            for (int rr = 0; rr < EVR; ++rr) { e[ss][rr][0] = interp(c0, hp, R, r0 + rr); e[ss][rr][1] = interp(c1, hp, R, r0 + rr); }
        }
    };
    if constexpr (STAGED) gather(cn, 0, 0, ev);
    STAMP_ADD(6);                                                                // pilot outputs + alpha
    for (int sb = 0; sb < S; sb += SSN) {
        const int ns = SSN == 1 ? 1 : min(SSN, S - sb);
        PlanRec cc[SSN][2];
#pragma unroll
        for (int ss = 0; ss < SSN; ++ss) {
            cc[ss][0] = cn[ss][0]; cc[ss][1] = cn[ss][1];
            const int sn = sb + SSN + ss;
            if (cw && sn < S) {
                if constexpr (CP) {
                    cn[ss][0] = ld_plan(plan, idn[ss][0]); cn[ss][1] = ld_plan(plan, idn[ss][1]);
                    const int s2 = sn + SSN;
                    if (s2 < S) { idn[ss][0] = triof[s2 * nsc + kc0]; idn[ss][1] = triof[s2 * nsc + kc1]; }
                } else { cn[ss][0] = ld_plan(plan, sn * nsc + kc0); cn[ss][1] = ld_plan(plan, sn * nsc + kc1); }
            }
        }
        for (int rg = 0; rg < ngroups; ++rg) {
            const int r0 = rg * RG;
            const unsigned e_h = (unsigned)((sb * R + r0) * T) * (unsigned)nsc;
            float2* __restrict__ l_l = slab + seg_l + ((par_h + e_h) & 1);
            float2* __restrict__ l_m = slab + seg_m + ((par_h + e_h) & 1);
            const bool more = rg + 1 < ngroups || sb + SSN < S;                  // another unit follows (uniform)
            if (want_metrics && cw) {
                // sum_t |H_t - v|^2 = sum_t |H_t - Hbar|^2 + T |Hbar - v|^2 ; the first term is
                // estimate-independent and comes analytically from K1 (ST_SUM_VAR).
                // (STAGED: the tap twiddles are RECOMPUTED here, once per unit, instead of staying in 2 NT registers through the whole
                //  estimate phase, which was the kernel's register peak)
                float2 we_m[STAGED && !MF ? NT : 1];
                if constexpr (STAGED && !MF) {
#pragma unroll
                    for (int d = 1; d < NT; ++d) { const float2 e = cis_turns((float)(bin_of(kc0) * A.prof->delay[d]) * invN_tw); we_m[d] = make_float2(e.x, -e.y); }
                    we_m[0] = make_float2(1.f, 0.f);                       // (delay 0)
                }
#pragma unroll
                for (int ss = 0; ss < SSN; ++ss) {
                    if (ss >= ns) continue;
                    for (int rr = 0; rr < RG; ++rr) {
                        const int r = r0 + rr;
                        const float al = A.alpha[r];
                        float2 e0, e1;
                        if constexpr (STAGED) { e0 = ev[ss][rr][0]; e1 = ev[ss][rr][1]; }
                        else { e0 = interp(plan_rec<CP>(cc[ss][0], sb + ss, kc0), hp, R, r); e1 = interp(plan_rec<CP>(cc[ss][1], sb + ss, kc1), hp, R, r); }
                        const float2* __restrict__ coef = A.csum_g + (size_t)(sb + ss) * KTS * R + r;      // tap d at stride R
                        float2 hb0 = make_float2(0.f, 0.f), hb1 = make_float2(0.f, 0.f);
#pragma unroll
                        for (int d = 0; d < NT; ++d) {
                            const float2 c = coef[d * R];
                            if constexpr (STAGED && !MF) {
                                cfma(hb0, c, we_m[d]);
                                cfma(hb1, c, cmul(we_m[d], make_float2(A.prof->step[d][0], A.prof->step[d][1])));
                            } else {
                                cfma(hb0, c, w_even(d));
                                cfma(hb1, c, w_odd(d));
                            }
                        }
                        if (act) {
                            const float2 m = make_float2(hb0.x * invT, hb0.y * invT), f = make_float2(al * e0.x, al * e0.y);
                            m_ls += cabs2(make_float2(m.x - e0.x, m.y - e0.y));
                            m_mm += cabs2(make_float2(m.x - f.x, m.y - f.y));
                        }
                        if (v1) {
                            const float2 m = make_float2(hb1.x * invT, hb1.y * invT), f = make_float2(al * e1.x, al * e1.y);
                            m_ls += cabs2(make_float2(m.x - e1.x, m.y - e1.y));
                            m_mm += cabs2(make_float2(m.x - f.x, m.y - f.y));
                        }
                    }
                }
            }
            if (Lb == nullptr && Mb == nullptr) {                                // metrics only: nothing to stream (uniform)
                if constexpr (STAGED) { if (more) { if (rg + 1 < ngroups) gather(cc, sb, r0 + RG, ev); else gather(cn, sb + SSN, 0, ev); } }
                continue;
            }
            STAMP_ADD(7);                                                        // E compute (metrics)
            if (!first_unit) __syncthreads();
            first_unit = false;
            STAMP_ADD(8);                                                        // E barrier B
            if (act) {
#pragma unroll
                for (int ss = 0; ss < SSN; ++ss) {
                    if (ss >= ns) continue;
                    for (int rr = 0; rr < RG; ++rr) {
                        const int r = r0 + rr;
                        const float al = A.alpha[r];
                        float2 e0, e1;
                        if constexpr (STAGED) { e0 = ev[ss][rr][0]; e1 = ev[ss][rr][1]; }
                        else { e0 = interp(plan_rec<CP>(cc[ss][0], sb + ss, kc0), hp, R, r); e1 = interp(plan_rec<CP>(cc[ss][1], sb + ss, kc1), hp, R, r); }
                        const float2 f0 = make_float2(al * e0.x, al * e0.y), f1 = make_float2(al * e1.x, al * e1.y);
                        for (int t = 0; t < T; ++t) {
                            const int row = ((ss * RG + rr) * T + t) * nsc;
                            if (Lb != nullptr) put(l_l + row, e0, e1);
                            if (Mb != nullptr) put(l_m + row, f0, f1);
                        }
                    }
                }
            }
            STAMP_ADD(9);                                                        // E slab writes
            __syncthreads();
            STAMP_ADD(10);                                                       // E barrier A
            if constexpr (STAGED) {                                              // the next unit's estimates, ahead of this unit's stores
                if (more) { if (rg + 1 < ngroups) gather(cc, sb, r0 + RG, ev); else gather(cn, sb + SSN, 0, ev); }
            }
            if (storer) {
                int ph = swave;
                if (Lb != nullptr) { const int n = seg_store(Lb + e_h, l_l, ns * RG * T * nsc, ph, nstw); ph = (ph + nstw - n % nstw) % nstw; }
                if (Mb != nullptr) seg_store(Mb + e_h, l_m, ns * RG * T * nsc, ph, nstw);
            }
            STAMP_ADD(11);                                                       // E store-out
        }
    }
#ifdef CHEST_STAMPS
    if (g_stamp_buf != nullptr && lane == 0) {
        unsigned long long* dst = g_stamp_buf + ((size_t)b * 16 + wave) * 16;
        for (int i = 0; i < 12; ++i) dst[i] = st_sum[i];
    }
#endif
    if (want_metrics) {
        __syncthreads();
        m_ls = block_sum(m_ls, A.red);
        m_mm = block_sum(m_mm, A.red);
        if (tid == 0) {
            float* m = out.metrics + (size_t)b * CHEST_NUM_METRICS;
            m[CHEST_M_SUM_H2] = A.rec[ST_SUM_H2];
            m[CHEST_M_SUM_ERR_LS] = A.rec[ST_SUM_VAR] + (float)T * m_ls;
            m[CHEST_M_SUM_ERR_MMSE] = A.rec[ST_SUM_VAR] + (float)T * m_mm;
            m[CHEST_M_SIGNAL_POWER] = A.rec[ST_SIGNAL_POWER]; m[CHEST_M_NOISE_STD] = noise_std;
            m[CHEST_M_ALPHA0] = A.alpha[0]; m[CHEST_M_COUNT] = (float)(S * R * T) * (float)fp.nsc;
            m[CHEST_M_RESERVED] = 0.f;
        }
    }
}

// MAXTHR: largest workgroup this instantiation is launched with (448 = 5 compute waves for nsc <= 640, the 1024-FFT grids, and up
// to two store waves; 768 = 10 + 2 for the 2048-FFT grids).
template <int MAXTHR, int MINW, int TT, int RR, int RGT, int SS, bool INJ, bool CP, bool MF = false>
__global__ void __launch_bounds__(MAXTHR, MINW)
frame_slab_kernel(const FrameParams* __restrict__ fpp, const chest_frame_desc* __restrict__ descs,
                  const PatternDev* __restrict__ patterns, int Pmax, uint64_t seed, uint64_t frame_offset,
                  const chest_injected inj, const chest_frame_outputs out, const float* __restrict__ state,
                  int RG, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const FrameParams& fp = *fpp;
    const int tid = threadIdx.x;
    int b = blockIdx.x;
#ifdef CHEST_STAMPS
    const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_mt0 = __builtin_amdgcn_s_memtime();
#endif
    {
        // Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share one).  Give each XCD a
        // contiguous frame range: its L2 then holds one pattern group's tables and its writes are one sequential region.
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);        // bijective for any grid size
    }
    const int S = fp.S, T = fp.T, R = fp.R, G = fp.G;
    const float* rec = state + (size_t)b * frame_state_floats(S, R, T);
    const chest_frame_desc fd = descs[b];
    // A failed frame leaves its outputs untouched (chest.h) -- except its rows of h_ls_pilots, which are written as zeros: the
    // full-covariance filter reads that tensor as ONE GEMM operand (scaled by its largest magnitude), so whatever an uninitialised
    // allocation held there would reach the other frames' results.
    auto fail_frame = [&]() {
        if (out.h_ls_pilots != nullptr) {
            float2* hl = (float2*)out.h_ls_pilots + (size_t)b * R * out.h_ls_stride;
            for (int i = tid; i < R * out.h_ls_stride; i += blockDim.x) hl[i] = make_float2(0.f, 0.f);
        }
    };
    if (rec[ST_VALID] == 0.f) { fail_frame(); return; }          // K1 flagged the descriptor (uniform exit)
    const PatternDev pat = patterns[fd.pattern_slot];
    if (pat.bits == nullptr) {
        if (tid == 0) { atomicExch(status, 1); if (out.frame_status) out.frame_status[b] = 1; }
        fail_frame();
        return;
    }
    const bool want_est = out.H_ls != nullptr || out.H_mmse != nullptr || out.metrics != nullptr;
    if (want_est && (pat.plan == nullptr || (pat.pad != 0) != CP)) {       // no plan, or not in the format this launch reads
        if (tid == 0) { atomicExch(status, 2); if (out.frame_status) out.frame_status[b] = 2; }
        fail_frame();
        return;
    }
    // Compact plans belong to ONE frame and come from HBM.  A read that misses the caches while the chip is saturated with
    // this kernel's stores waits microseconds, and the estimate phase would meet one such wait per unit (2.94 ms per 10 000
    // frames, against 2.0 ms when the tables are cached).  The tables are small enough -- 44 KB at 14 x 599 -- to be pulled in
    // whole right here, one dword per 128-B line, one or two loads per thread, while the mask below is written: 2.47 ms.
    uint32_t touch0 = 0, touch1 = 0;
    if constexpr (CP) {
        if (want_est) {
            const CHEST_GLOBAL uint32_t* pw = (const CHEST_GLOBAL uint32_t*)pat.plan;
            const int nlines = (pat.pad + 2 * G + 127) >> 7;     // triangle records | triangle id of every element
            touch0 = pw[min(tid, nlines - 1) * 32];
            touch1 = pw[min(tid + (int)blockDim.x, nlines - 1) * 32];
            for (int i = tid + 2 * (int)blockDim.x; i < nlines; i += blockDim.x) touch1 += pw[i * 32];
        }
    }

    // ---- pilot mask: 16 resource elements per thread and one 16-B store, after peeling the row to 16-B alignment
    //      (G is rarely a multiple of 16, so most rows start misaligned) ---------------------------------------------
    const gptr_bits gbits = (gptr_bits)pat.bits;
    auto is_pilot = [&](int g) -> uint32_t { return (gbits[g >> 5].x >> (g & 31)) & 1u; };
    if (out.pilot_mask != nullptr) {
        uint8_t* mrow = out.pilot_mask + (size_t)b * G;
        const int lead = min(G, (int)((16 - ((uintptr_t)mrow & 15)) & 15));
        const int nvec = (G - lead) >> 4, tail0 = lead + (nvec << 4);
        if (tid < lead) mrow[tid] = (uint8_t)is_pilot(tid);
        for (int v = tid; v < nvec; v += blockDim.x) {
            const uint32_t m16 = mask_bits16(gbits, lead + (v << 4));
            uint32_t wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t n = (m16 >> (4 * q)) & 0xFu;                   // four mask bits -> four 0/1 bytes
                wq[q] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
            }
            *(uint4*)(mrow + lead + (v << 4)) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
        if (tid < G - tail0) mrow[tail0 + tid] = (uint8_t)is_pilot(tail0 + tid);
    }
    // the same mask as float32 0/1 (the dataset schema of run_phase3_dataset_generation.py:139): 4 elements per 16-B store
    if (out.pilot_mask_f32 != nullptr) {
        float* mrow = out.pilot_mask_f32 + (size_t)b * G;
        const int lead = min(G, (int)(((16 - ((uintptr_t)mrow & 15)) & 15) >> 2));
        const int nvec = (G - lead) >> 2, tail0 = lead + (nvec << 2);
        if (tid < lead) mrow[tid] = is_pilot(tid) ? 1.f : 0.f;
        for (int v = tid; v < nvec; v += blockDim.x) {
            const uint32_t m4 = mask_bits16(gbits, lead + (v << 2));
            const v4f_t m = {(m4 & 1u) ? 1.f : 0.f, (m4 & 2u) ? 1.f : 0.f, (m4 & 4u) ? 1.f : 0.f, (m4 & 8u) ? 1.f : 0.f};
            __builtin_nontemporal_store(m, (v4f_t*)(mrow + lead + (v << 2)));
        }
        if (tid < G - tail0) mrow[tail0 + tid] = is_pilot(tail0 + tid) ? 1.f : 0.f;
    }

    if constexpr (CP) asm volatile("" :: "v"(touch0), "v"(touch1));          // the touches have landed (nothing else uses their values)
    SlabArgs A;
    A.fp = &fp; A.pat = &pat; A.prof = &fp.prof[fd.channel_type]; A.rec = rec;
    A.c_g = (const float2*)(rec + ST_SCAL);
    A.csum_g = A.c_g + (size_t)S * R * T * KTS;
    A.inj = &inj; A.out = &out;
    // (a launch that produces no estimate -- the full-covariance route reads h_ls_pilots instead -- keeps no pilot table in LDS:
    //  53.7 KB at 4x4 / P = 1 678, which is what lets two such workgroups share a CU)
    const size_t hp_elems = want_est ? (((size_t)Pmax * R + 1) & ~(size_t)1) : 0;
    A.hp = want_est ? (float2*)lds_raw : nullptr;                // [Pmax*R rounded up to even]
    A.red = (float*)((float2*)lds_raw + hp_elems);               // [32]
    A.alpha = A.red + 32;                                        // [8]
    A.slab = (float2*)(A.alpha + 8);                             // 16-B aligned: hp span is a multiple of 16 B, 40 floats = 160 B
    const uint64_t frame = frame_offset + (uint64_t)b;
    A.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};
    A.b = b; A.RG = RG & 0xFF; A.nst = (RG >> 8) & 0xFF; A.nsx = (RG >> 16) & 0xFF;
    // a second slab behind the first when the launch was given the LDS for it (bit 24 of the crew argument) and produces no estimates
    A.slab2 = ((RG >> 24) & 1) && !want_est ? (slab_elems(T, RG & 0xFF, fp.nsc, SS, false) + 1) & ~1 : 0;
    switch (A.prof->ntaps) {                                     // workgroup-uniform dispatch on the profile
        case 5: slab_body<5, TT, RR, RGT, SS, INJ, CP, MF>(A); break;     // EPA at 15.36 MHz
        case 8: slab_body<8, TT, RR, RGT, SS, INJ, CP, MF>(A); break;     // EVA
        default: slab_body<KT, TT, RR, RGT, SS, INJ, CP, MF>(A); break;   // ETU / anything else (taps padded with zeros)
    }
#ifdef CHEST_STAMPS
    if (g_stamp_buf != nullptr && (tid & 63) == 0) {
        unsigned long long* dst = g_stamp_buf + ((size_t)b * 16 + (tid >> 6)) * 16;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dst[12] = st_rt0; dst[13] = __builtin_amdgcn_s_memrealtime(); dst[14] = __builtin_amdgcn_s_memtime() - st_mt0;
        dst[15] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

size_t slab_lds_bytes(int nsc, int T, int R, int RG, int Pmax, int SS, bool with_estimates = true) {
    const size_t hp = with_estimates ? (((size_t)Pmax * R + 1) & ~(size_t)1) : 0;
    return (hp + (size_t)slab_elems(T, RG, nsc, SS, with_estimates)) * sizeof(float2) + 40 * sizeof(float);
}

// Rx antennas per unit: the largest divisor of R whose slab still lets two workgroups share a CU's 160 KiB of LDS (long
// contiguous bursts and few barriers matter more than a third or fourth resident workgroup: 2x2 frames measure 2.64 ms
// with whole-symbol units against 3.21 ms with one rx antenna per unit); if none does, the largest that fits at all.
// ctx->tune_rx_group overrides (chest_tune, tools only).
static int choose_rx_group(const chest_ctx* ctx, int Pmax) {
    const FrameParams& fp = ctx->fp;
    if (ctx->tune_rx_group > 0 && fp.R % ctx->tune_rx_group == 0) return ctx->tune_rx_group;
    int best2 = 0, best1 = 1;
    for (int rg = 1; rg <= fp.R; ++rg) {
        if (fp.R % rg) continue;
        const size_t lds = slab_lds_bytes(fp.nsc, fp.T, fp.R, rg, Pmax, 1);
        if (lds <= 80 * 1024) best2 = rg;
        if (lds <= 160 * 1024) best1 = rg;
    }
    return best2 ? best2 : best1;
}

template <int MAXTHR, int MINW, int TT, int RR, int RGT, int SS, bool INJ, bool CP = false, bool MF = false>
static int launch_one(chest_ctx* ctx, int which, hipStream_t st, dim3 grid, dim3 block, size_t lds, const chest_frame_desc* descs, int Pmax,
                      uint64_t seed, uint64_t frame_offset, const chest_injected& inj, const chest_frame_outputs& out,
                      const float* workspace, int RG) {
    if ((int)block.x > MAXTHR) { set_error("frame kernel: a %u-thread workgroup for an instantiation built for %d", block.x, MAXTHR); return CHEST_EINVAL; }
    CHEST_LDS_OPT_IN(ctx, which, (frame_slab_kernel<MAXTHR, MINW, TT, RR, RGT, SS, INJ, CP, MF>), lds);
    hipLaunchKernelGGL((frame_slab_kernel<MAXTHR, MINW, TT, RR, RGT, SS, INJ, CP, MF>), grid, block, lds, st, ctx->d_fp, descs, ctx->d_patterns, Pmax,
                       seed, frame_offset, inj, out, workspace, RG, ctx->d_status);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int launch_k2_slab(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, int Pmax, uint64_t seed,
                   uint64_t frame_offset, const chest_injected& inj, const chest_frame_outputs& out, const float* workspace, bool compact) {
    const FrameParams& fp = ctx->fp;
    const int npair = (fp.nsc + 1) / 2, nwave = (npair + 63) / 64;
    if (nwave > 16) { set_error("frame kernel handles at most 2047 used subcarriers (got %d)", fp.nsc); return CHEST_EINVAL; }
    const int RG = choose_rx_group(ctx, Pmax);
    // dedicated store waves per workgroup (slab_body): two for the 4x4 frames of the 1024-FFT grids (five compute waves at 128
    // VGPRs, two workgroups per CU: 4.51 -> 4.32 ms per 6 250 frames; one extra wave is always slower -- six or eleven waves sit
    // unevenly on four SIMDs -- and 2x2 frames lose 3 % with two: profiles/r04_store_crew.txt); chest_tune "frame_store_extra"
    // overrides (-1 = none)
    int nsx = ctx->tune_store_extra != 0 ? (ctx->tune_store_extra < 0 ? 0 : ctx->tune_store_extra) : ((fp.T * fp.R >= 16 && nwave <= 5) ? 2 : 0);
    if (nsx > 2) nsx = 2;                                       // (the instantiations' launch bounds leave room for two)
    if (nwave + nsx > 16) nsx = 0;

    const int T = fp.T, R = fp.R;
    const bool injected_in = inj.u_sym || inj.u_angle || inj.n_re;      // parity / replay mode: the generic instantiations
    // SISO frames have three rows per symbol: two symbols per unit halve the per-unit synchronisation (the spans of
    // consecutive symbols are contiguous when a unit holds all rx antennas)
#ifndef CHEST_SS_SISO
#define CHEST_SS_SISO 2
#endif
    constexpr int SS_SISO = CHEST_SS_SISO;
    // the specialised (device-RNG) instantiations start their tap sums from the first tap's gains: every profile's first tap must sit at
    // delay 0 (true of the reference's three tables at any sampling rate); otherwise the generic instantiations run
    bool zero_first = true;
    for (int ty = 0; ty < CHEST_NUM_PROFILES; ++ty) zero_first = zero_first && fp.prof[ty].ntaps > 0 && fp.prof[ty].delay[0] == 0;
    const bool injected = injected_in || !zero_first;
    const bool siso = !injected && !(fp.half & 1) && nwave <= 5 && T == 1 && R == 1 && RG == 1;
    // a launch without estimates (H_ls / H_mmse / metrics) keeps neither the pilot table nor the estimate segments in LDS
    const bool want_est = out.H_ls != nullptr || out.H_mmse != nullptr || out.metrics != nullptr;
    size_t lds = slab_lds_bytes(fp.nsc, fp.T, fp.R, RG, Pmax, siso ? SS_SISO : 1, want_est);
    // One workgroup per CU (more than 80 KB... or 10 waves) and no estimates: a SECOND slab + two dedicated store waves let the store
    // stream of unit u run beside the synthesis of unit u + 1 (slab_body).  chest_tune "frame_two_slabs" -1 switches it off (A/B).
    bool two_slabs = false;
    if (!want_est && !injected && !(fp.half & 1) && nwave == 10 && fp.T == 4 && fp.R == 4 && RG == 1 && ctx->tune_two_slabs >= 0 && ctx->tune_store_extra >= 0) {
        const size_t extra = (size_t)((slab_elems(fp.T, RG, fp.nsc, 1, false) + 1) & ~1) * sizeof(float2);
        if (lds + extra <= 160 * 1024) { lds += extra; two_slabs = true; nsx = ctx->tune_store_extra > 0 ? (ctx->tune_store_extra > 4 ? 4 : ctx->tune_store_extra) : 2; }
    }
    if (lds > 160 * 1024) { set_error("frame kernel needs %zu B of LDS (> 160 KiB): too many antennas, subcarriers or pilots", lds); return CHEST_EINVAL; }
    const int RGarg = RG | (ctx->tune_store_waves << 8) | (nsx << 16) | ((two_slabs ? 1 : 0) << 24);   // (crew parameters ride in the high bits of the kernel argument)
    const dim3 grid((unsigned)B), block((unsigned)(nwave + nsx) * 64);
#define CHEST_SLAB_ARGS st, grid, block, lds, descs, Pmax, seed, frame_offset, inj, out, workspace, RGarg
    const bool w5 = ctx->tune_min_waves == 5;                 // (A/B) the 96-VGPR instantiations of the 5-wave workgroups
    if (compact) {
        // compact plans (chest_patterns_build method 2; every frame of the launch must use one -- the kernel flags a frame whose
        // pattern is in the other format): the device-RNG instantiations of the 1024-FFT grids and the generic one
        if (nwave > 5) { set_error("compact interpolation plans are read by the frame kernels of grids up to 640 used subcarriers (got %d)", fp.nsc); return CHEST_EINVAL; }
        if (!injected && !(fp.half & 1)) {
            // (80-VGPR instantiations for the 5-wave workgroups: see the note on workgroup placement below)
            if (T == 2 && R == 2 && RG == 2 && w5) return launch_one<448, 5, 2, 2, 2, 1, false, true>(ctx, LDS_FRAME_CP_W5, CHEST_SLAB_ARGS);
            if (T == 2 && R == 2 && RG == 2) return launch_one<448, 6, 2, 2, 2, 1, false, true>(ctx, LDS_FRAME_CP, CHEST_SLAB_ARGS);
            if (T == 4 && R == 4 && RG == 1) return launch_one<448, 5, 4, 4, 1, 1, false, true>(ctx, LDS_FRAME_44_CP, CHEST_SLAB_ARGS);
            if (siso && w5) return launch_one<448, 5, 1, 1, 1, SS_SISO, false, true>(ctx, LDS_FRAME_11_CP_W5, CHEST_SLAB_ARGS);
            if (siso) return launch_one<448, 6, 1, 1, 1, SS_SISO, false, true>(ctx, LDS_FRAME_11_CP, CHEST_SLAB_ARGS);
        }
        return launch_one<448, 4, 0, 0, 0, 1, true, true>(ctx, LDS_FRAME_GEN_CP, CHEST_SLAB_ARGS);
    }
    if (nwave <= 5) {                      // the 1024-FFT grids (nsc <= 640): specialised antenna counts on the device-RNG path
        if (!injected) {
            if (fp.half & 1) return launch_one<448, 4, 0, 0, 0, 1, true>(ctx, LDS_FRAME_GEN, CHEST_SLAB_ARGS);   // a pair straddles DC
            // Workgroup placement (round 4, tools/micro/dispatch_rate.hip, profiles/r04_workgroup_placement.txt).  Three 5-wave workgroups
            // per CU are 15 waves; at 96 VGPRs a SIMD has room for five, 20 per CU -- and the dispatcher still leaves a freed slot empty
            // for 35 us on average (a 5-wave workgroup needs TWO wave slots on one SIMD; whether the SIMD its placement starts on has
            // them is luck): 2.38 workgroups resident on average, not 3.  A kernel that only waits shows the same: 2.25-2.44 resident at
            // 96 VGPRs, 2.64 at 80 (six waves per SIMD), 2.74 at 64 or with 4-wave workgroups.  So the 5-wave instantiations are built
            // for 80 VGPRs: 2.86 resident, refill gap 1 us, 1.81 against 1.96 ms per 10 000 2x2 EVA frames.  (Getting there took the
            // tap twiddles out of the estimate phase -- recomputed once per unit there, 96 -> 89 VGPRs --, starting the tap sums from the
            // zero-delay first tap's gains, 89 -> 85, and a handful of spilled registers.)  chest_tune "frame_min_waves" 5 / 4 select the
            // 96- / 128-VGPR instantiations for A/B.
            if (T == 2 && R == 2 && RG == 2 && ctx->tune_min_waves == 4) return launch_one<448, 4, 2, 2, 2, 1, false>(ctx, LDS_FRAME_W5, CHEST_SLAB_ARGS);
            if (T == 2 && R == 2 && RG == 2 && w5) return launch_one<448, 5, 2, 2, 2, 1, false>(ctx, LDS_FRAME_W6, CHEST_SLAB_ARGS);
            if (T == 2 && R == 2 && RG == 2) return launch_one<448, 6, 2, 2, 2, 1, false>(ctx, LDS_FRAME, CHEST_SLAB_ARGS);
            // (4x4: two 7-wave workgroups per CU, 5 compute + 2 store waves.  At the 128-VGPR budget -- four wave slots per SIMD, 16 for 14
            //  waves -- 1.72 of 2 were resident (refill gap mean 51 us of a 333-us lifetime); at 96 VGPRs 1.93 (gap 7 us): 4.41 -> 4.22 ms per
            //  6 250 ETU frames; 80 VGPRs spill 62 registers, 4.67.  "frame_min_waves" 4 = the 128-VGPR instantiation)
            if (T == 4 && R == 4 && RG == 1 && ctx->tune_min_waves == 4) return launch_one<448, 4, 4, 4, 1, 1, false>(ctx, LDS_FRAME_44_W5, CHEST_SLAB_ARGS);
            if (T == 4 && R == 4 && RG == 1) return launch_one<448, 5, 4, 4, 1, 1, false>(ctx, LDS_FRAME_44, CHEST_SLAB_ARGS);
            if (siso && w5) return launch_one<448, 5, 1, 1, 1, SS_SISO, false>(ctx, LDS_FRAME_11_W5, CHEST_SLAB_ARGS);
            // (SISO: 36 KB of LDS, FOUR workgroups fit a CU -- 20 waves: 72 VGPRs, seven wave slots per SIMD, place them best: 0.89 ms per
            //  10 000 frames at the 96-VGPR budget, 0.77 at 80, 0.73 at 72 (18 spilled registers), 0.76 at 64; "frame_min_waves" 6 = 80)
            if (siso && ctx->tune_min_waves == 6) return launch_one<448, 6, 1, 1, 1, SS_SISO, false>(ctx, LDS_FRAME_11_W7, CHEST_SLAB_ARGS);
            if (siso) return launch_one<448, 7, 1, 1, 1, SS_SISO, false>(ctx, LDS_FRAME_11, CHEST_SLAB_ARGS);
        }
        return launch_one<448, 4, 0, 0, 0, 1, true>(ctx, LDS_FRAME_GEN, CHEST_SLAB_ARGS);
    }
    if (nwave <= 10) {                     // the 2048-FFT grids (nsc <= 1280)
        // (4x4 without estimates -- the full-covariance route -- needs 67 KB: two workgroups would fit a CU with a 96-VGPR
        //  instantiation (five waves per SIMD).  Built and measured in round 4: 54 spilled registers, 2.00 against 1.90 ms per
        //  2 304 frames, 4.23 against 4.01 per 4 608 -- the synthesis is bound by the instructions it issues, not by residency;
        //  profiles/r04_cfg5_frame_kernel_isa.txt.  Not kept.)
        // the launch of the full-covariance route (two slabs, 10 + 2 waves): tap sums on the matrix pipe (slab_body, MF): 1.81 against
        // 1.90 ms per 2 304 frames, 3.70 against 4.09 per 4 608 (tools/ab_mfma_synth.py; chest_tune "frame_min_waves" 4 = the vector-ALU
        // instantiation, for A/B).  H and rx agree with the vector-ALU form to 2.5e-7 of the tensor's scale (summation order).
        if (!injected && !(fp.half & 1) && T == 4 && R == 4 && RG == 1 && two_slabs && nwave + nsx <= 12 && ctx->tune_min_waves != 4)
            return launch_one<768, 3, 4, 4, 1, 1, false, false, true>(ctx, LDS_FRAME_640_44_MF, CHEST_SLAB_ARGS);
        if (!injected && !(fp.half & 1) && T == 4 && R == 4 && RG == 1) return launch_one<896, 4, 4, 4, 1, 1, false>(ctx, LDS_FRAME_640_44, CHEST_SLAB_ARGS);   // (10 compute + up to 4 store waves, <= 128 VGPRs)
        return launch_one<768, 3, 0, 0, 0, 1, true>(ctx, LDS_FRAME_640, CHEST_SLAB_ARGS);
    }
    return launch_one<1024, 4, 0, 0, 0, 1, true>(ctx, LDS_FRAME_1024, CHEST_SLAB_ARGS);
#undef CHEST_SLAB_ARGS
}

}  // namespace chest

#ifdef CHEST_STAMPS
extern "C" int chest_debug_set_stamp_buffer(void* dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(chest::g_stamp_buf), &dev_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif
