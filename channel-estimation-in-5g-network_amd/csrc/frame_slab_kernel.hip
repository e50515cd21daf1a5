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

// Output tensors are written once and never read back: nontemporal stores (tools/micro/store_flavour.hip).
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
// between the enclosing line start and g.  Stores q0, q0 + qstep, ... are this wave's.  Returns the number of stores.
__device__ __forceinline__ int seg_store(float2* __restrict__ g, const float2* __restrict__ l, int n, int q0, int qstep, int lane) {
    const int head = (int)(((uintptr_t)g & 127u) >> 3);
    const int nst = (head + n + 127) >> 7;
    for (int q = q0; q < nst; q += qstep) {
        const int e = (q << 7) - head + 2 * lane;
        if (e >= 0 && e + 1 < n) st16(g + e, *(const float4*)(l + e));
        else if (e == -1) st8(g, l[0]);                       // odd head: element 0 sits alone in its 16-B slot
        else if (e >= 0 && e == n - 1) st8(g + e, l[e]);
    }
    return nst;
}

struct SlabArgs {
    const FrameParams* fp;
    const PatternDev* pat;
    const TapProfile* prof;
    const float* rec;
    const float2* c_g;
    const float2* csum_g;
    const chest_injected* inj;
    const chest_frame_outputs* out;
    float2* hp;        // LDS [Pmax*R] LS estimates at the pilots
    float* red;
    float* alpha;
    float2* slab;      // LDS, 16-B aligned
    RngKey key;
    int b, RG;
};

// LDS slab segment capacities in float2 elements (even, +2 so that a parity shift of one element fits)
__host__ __device__ inline int slab_seg(int rows, int nsc) { return ((rows * nsc + 1) & ~1) + 2; }
__host__ __device__ inline int slab_elems(int T, int RG, int nsc) {
    const int w = slab_seg(T, nsc) + slab_seg(RG * T, nsc) + slab_seg(RG, nsc);
    const int e = 2 * slab_seg(RG * T, nsc);
    return w > e ? w : e;
}

template <int NT>
__device__ __forceinline__ void slab_body(const SlabArgs& A) {
    const FrameParams& fp = *A.fp;
    const PatternDev& pat = *A.pat;
    const chest_injected& inj = *A.inj;
    const chest_frame_outputs& out = *A.out;
    const int S = fp.S, G = fp.G, P = pat.P, b = A.b, T = fp.T, R = fp.R, RG = A.RG;
    const int nsc = fp.nsc;
    const int tid = threadIdx.x, lane = tid & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned npair = ((unsigned)nsc + 1) >> 1;
    const int k0 = 2 * tid, k1 = k0 + 1;
    const bool v0 = k0 < nsc, v1 = k1 < nsc;
    const int kc0 = v0 ? k0 : nsc - 1, kc1 = v1 ? k1 : nsc - 1;
    const float noise_std = A.rec[ST_NOISE_STD];
    float2* __restrict__ hp = A.hp;
    float2* __restrict__ slab = A.slab;
    float2* __restrict__ txb = out.tx ? (float2*)out.tx + (size_t)b * S * T * nsc : nullptr;
    float2* __restrict__ rxb = out.rx ? (float2*)out.rx + (size_t)b * S * R * nsc : nullptr;
    float2* __restrict__ Hb = out.H ? (float2*)out.H + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ Lb = out.H_ls ? (float2*)out.H_ls + (size_t)b * S * R * T * nsc : nullptr;
    float2* __restrict__ Mb = out.H_mmse ? (float2*)out.H_mmse + (size_t)b * S * R * T * nsc : nullptr;
    const float* __restrict__ usym = inj.u_sym ? inj.u_sym + (size_t)b * G : nullptr;
    const float* __restrict__ nre = inj.n_re ? inj.n_re + (size_t)b * S * R * nsc : nullptr;
    const float* __restrict__ nim = inj.n_im ? inj.n_im + (size_t)b * S * R * nsc : nullptr;
    const uint16_t* __restrict__ slots = pat.slot;
    // parity (element offset & 1) of the frame's first element in each tensor; 16-B-aligned tensor bases (host-checked)
    const int par_tx = (int)(((size_t)b * S * T * nsc) & 1), par_rx = (int)(((size_t)b * S * R * nsc) & 1),
              par_h = (int)(((size_t)b * S * R * T * nsc) & 1);

    // ---- tap twiddles of this thread's two subcarriers: registers for the whole frame ----------------------------
    float4 w[NT];
    {
        const float invN = 1.0f / (float)fp.N;
        const int half = fp.half;
        const int ka = (kc0 < half) ? (kc0 - half) : (kc0 - half + 1), kb = (kc1 < half) ? (kc1 - half) : (kc1 - half + 1);
#pragma unroll
        for (int d = 0; d < NT; ++d) {
            const int dl = A.prof->delay[d];
            const float2 e0 = cis_turns((float)(ka * dl) * invN), e1 = cis_turns((float)(kb * dl) * invN);   // |kappa*delay| < 2^24: exact
            w[d] = make_float4(e0.x, -e0.y, e1.x, -e1.y);
        }
    }

    const int seg_tx = 0, seg_h = slab_seg(T, nsc), seg_rx = seg_h + slab_seg(RG * T, nsc);
    const int ngroups = R / RG;
    bool first_unit = true;

    // ---- phase W: per unit (symbol, rx group) assemble tx / H / rx rows in the slab, then stream them out ------------
    uint32_t sl0 = slots[kc0], sl1 = slots[kc1];
    for (int s = 0; s < S; ++s) {
        const unsigned s0 = sl0, s1 = sl1;
        if (s + 1 < S) { sl0 = slots[(s + 1) * nsc + kc0]; sl1 = slots[(s + 1) * nsc + kc1]; }      // one symbol ahead
        float u0, u1;
        if (usym != nullptr) {
            u0 = usym[s0]; u1 = usym[s1];
        } else {
            const u32x4 z = draw(A.key, STREAM_SYM, (uint32_t)s * npair + (uint32_t)tid);
            u0 = u01(z.x); u1 = u01(z.y);
        }
        const float2 x0 = cis_unit(__builtin_amdgcn_fractf(u0));                 // exp(j 2pi u) (:395,:399)
        const float2 x1 = cis_unit(__builtin_amdgcn_fractf(u1));
        const bool p0 = v0 && s0 < (unsigned)P, p1 = v1 && s1 < (unsigned)P;
        for (int rg = 0; rg < ngroups; ++rg) {
            const int r0 = rg * RG;
            // global spans of this unit (element offsets inside the frame) and their parities
            const unsigned e_tx = (unsigned)(s * T) * (unsigned)nsc, e_h = (unsigned)((s * R + r0) * T) * (unsigned)nsc,
                           e_rx = (unsigned)(s * R + r0) * (unsigned)nsc;
            float2* __restrict__ l_tx = slab + seg_tx + ((par_tx + e_tx) & 1);
            float2* __restrict__ l_h = slab + seg_h + ((par_h + e_h) & 1);
            float2* __restrict__ l_rx = slab + seg_rx + ((par_rx + e_rx) & 1);
            if (!first_unit) __syncthreads();                                    // previous unit's store-out has read the slab
            first_unit = false;
            if (rg == 0 && txb != nullptr) {
                for (int t = 0; t < T; ++t) {
                    if (v0) l_tx[t * nsc + k0] = x0;
                    if (v1) l_tx[t * nsc + k1] = x1;
                }
            }
            for (int rr = 0; rr < RG; ++rr) {
                const int r = r0 + rr;
                // sum_t H_t is always accumulated from the per-tx rows, whether or not H is stored, so that rx / H_ls /
                // H_mmse are bit-identical for every choice of requested outputs.
                float2 hs0 = make_float2(0.f, 0.f), hs1 = make_float2(0.f, 0.f);
                for (int t = 0; t < T; ++t) {
                    const float2* __restrict__ coef = A.c_g + ((s * R + r) * T + t) * KTS;
                    float2 h0 = make_float2(0.f, 0.f), h1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int d = 0; d < NT; ++d) {
                        const float2 c = coef[d];
                        cfma(h0, c, make_float2(w[d].x, w[d].y));
                        cfma(h1, c, make_float2(w[d].z, w[d].w));
                    }
                    if (Hb != nullptr) {
                        if (v0) l_h[(rr * T + t) * nsc + k0] = h0;
                        if (v1) l_h[(rr * T + t) * nsc + k1] = h1;
                    }
                    hs0.x += h0.x; hs0.y += h0.y; hs1.x += h1.x; hs1.y += h1.y;
                }
                const float2 y0 = cmul(hs0, x0), y1 = cmul(hs1, x1);             // y = sum_t H x (:330-334)
                float2 n0v, n1v;
                if (nre != nullptr) {
                    const unsigned row = (unsigned)(s * R + r) * (unsigned)nsc;
                    n0v = make_float2((nre + row)[kc0], (nim + row)[kc0]);
                    n1v = make_float2((nre + row)[kc1], (nim + row)[kc1]);
                } else {
                    const u32x4 z = draw(A.key, STREAM_NOISE, (uint32_t)(s * R + r) * npair + (uint32_t)tid);
                    n0v = gauss_pair(z.x, z.y);
                    n1v = gauss_pair(z.z, z.w);
                }
                const float2 q0 = make_float2(fmaf(noise_std, n0v.x, y0.x), fmaf(noise_std, n0v.y, y0.y));
                const float2 q1 = make_float2(fmaf(noise_std, n1v.x, y1.x), fmaf(noise_std, n1v.y, y1.y));
                if (rxb != nullptr) {
                    if (v0) l_rx[rr * nsc + k0] = q0;
                    if (v1) l_rx[rr * nsc + k1] = q1;
                }
                if (p0) hp[s0 * R + r] = pilot_divide(q0, x0);                   // (src/baseline_estimators.py:110)
                if (p1) hp[s1 * R + r] = pilot_divide(q1, x1);
            }
            __syncthreads();                                                     // slab complete
            int ph = wave;                                                       // rotate the stores of consecutive segments over the waves
            if (rg == 0 && txb != nullptr) { const int n = seg_store(txb + e_tx, l_tx, T * nsc, ph, nwave, lane); ph = (ph + nwave - n % nwave) % nwave; }
            if (Hb != nullptr) { const int n = seg_store(Hb + e_h, l_h, RG * T * nsc, ph, nwave, lane); ph = (ph + nwave - n % nwave) % nwave; }
            if (rxb != nullptr) seg_store(rxb + e_rx, l_rx, RG * nsc, ph, nwave, lane);
        }
    }
    const bool want_metrics = out.metrics != nullptr;
    const bool want_est = (Lb != nullptr) || (Mb != nullptr) || want_metrics;
    __syncthreads();                                                             // hp complete; last store-out has read the slab

    // ---- pilot symbols in pilot order: regenerated from their own draws (identical arithmetic), linear 16-B stores ----
    if (out.pilot_symbols != nullptr) {
        float2* __restrict__ psb = (float2*)out.pilot_symbols + (size_t)b * out.pilot_stride;
        const bool al = (((uintptr_t)psb) & 15) == 0;
        for (int i = 2 * tid; i < P; i += 2 * (int)blockDim.x) {
            float2 xs[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = i + j < P ? i + j : P - 1;
                float u;
                if (usym != nullptr) {
                    u = usym[p];                                                 // first P draws are the pilot phases (:395)
                } else {
                    const int g = pat.pilot_idx[p];
                    const int s = g / nsc, k = g - s * nsc;
                    const u32x4 z = draw(A.key, STREAM_SYM, (uint32_t)s * npair + ((uint32_t)k >> 1));
                    u = u01((k & 1) ? z.y : z.x);
                }
                xs[j] = cis_unit(__builtin_amdgcn_fractf(u));
            }
            if (al && i + 1 < P) st16(psb + i, make_float4(xs[0].x, xs[0].y, xs[1].x, xs[1].y));
            else { st8(psb + i, xs[0]); if (i + 1 < P) st8(psb + i + 1, xs[1]); }
        }
    }
    if (out.h_ls_pilots != nullptr) {                // LS estimates at the pilots, [R][stride] per frame, zero padded
        float2* __restrict__ hl = (float2*)out.h_ls_pilots + (size_t)b * R * out.h_ls_stride;
        for (int i = tid; i < R * out.h_ls_stride; i += blockDim.x) {
            const int r = i / out.h_ls_stride, p = i - r * out.h_ls_stride;
            hl[i] = p < P ? hp[p * R + r] : make_float2(0.f, 0.f);
        }
    }
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
    const PlanRec* __restrict__ plan = (const PlanRec*)pat.plan;
    const float invT = 1.0f / (float)T;
    const int seg_l = 0, seg_m = slab_seg(RG * T, nsc);
    float m_ls = 0.f, m_mm = 0.f;
    PlanRec c0n = plan[kc0], c1n = plan[kc1];
    first_unit = true;
    for (int s = 0; s < S; ++s) {
        const PlanRec c0 = c0n, c1 = c1n;
        if (s + 1 < S) { c0n = plan[(s + 1) * nsc + kc0]; c1n = plan[(s + 1) * nsc + kc1]; }
        for (int rg = 0; rg < ngroups; ++rg) {
            const int r0 = rg * RG;
            const unsigned e_h = (unsigned)((s * R + r0) * T) * (unsigned)nsc;
            float2* __restrict__ l_l = slab + seg_l + ((par_h + e_h) & 1);
            float2* __restrict__ l_m = slab + seg_m + ((par_h + e_h) & 1);
            if (!first_unit) __syncthreads();
            first_unit = false;
            for (int rr = 0; rr < RG; ++rr) {
                const int r = r0 + rr;
                const float2 e0 = interp(c0, hp, R, r), e1 = interp(c1, hp, R, r);
                const float al = A.alpha[r];
                const float2 f0 = make_float2(al * e0.x, al * e0.y), f1 = make_float2(al * e1.x, al * e1.y);
                for (int t = 0; t < T; ++t) {
                    const int row = (rr * T + t) * nsc;
                    if (Lb != nullptr) { if (v0) l_l[row + k0] = e0; if (v1) l_l[row + k1] = e1; }
                    if (Mb != nullptr) { if (v0) l_m[row + k0] = f0; if (v1) l_m[row + k1] = f1; }
                }
                if (want_metrics) {
                    // sum_t |H_t - v|^2 = sum_t |H_t - Hbar|^2 + T |Hbar - v|^2 ; the first term is
                    // estimate-independent and comes analytically from K1 (ST_SUM_VAR).
                    const float2* __restrict__ coef = A.csum_g + (s * R + r) * KTS;
                    float2 hb0 = make_float2(0.f, 0.f), hb1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int d = 0; d < NT; ++d) {
                        const float2 c = coef[d];
                        cfma(hb0, c, make_float2(w[d].x, w[d].y));
                        cfma(hb1, c, make_float2(w[d].z, w[d].w));
                    }
                    if (v0) {
                        const float2 m = make_float2(hb0.x * invT, hb0.y * invT);
                        m_ls += cabs2(make_float2(m.x - e0.x, m.y - e0.y));
                        m_mm += cabs2(make_float2(m.x - f0.x, m.y - f0.y));
                    }
                    if (v1) {
                        const float2 m = make_float2(hb1.x * invT, hb1.y * invT);
                        m_ls += cabs2(make_float2(m.x - e1.x, m.y - e1.y));
                        m_mm += cabs2(make_float2(m.x - f1.x, m.y - f1.y));
                    }
                }
            }
            if (Lb == nullptr && Mb == nullptr) continue;                        // metrics only: nothing to stream (uniform)
            __syncthreads();
            int ph = wave;
            if (Lb != nullptr) { const int n = seg_store(Lb + e_h, l_l, RG * T * nsc, ph, nwave, lane); ph = (ph + nwave - n % nwave) % nwave; }
            if (Mb != nullptr) seg_store(Mb + e_h, l_m, RG * T * nsc, ph, nwave, lane);
        }
    }
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

// MAXTHR: largest workgroup this instantiation is launched with (320 = 5 waves covers nsc <= 640, the 1024-FFT grids).
template <int MAXTHR, int MINW>
__global__ void __launch_bounds__(MAXTHR, MINW)
frame_slab_kernel(const FrameParams* __restrict__ fpp, const chest_frame_desc* __restrict__ descs,
                  const PatternDev* __restrict__ patterns, int Pmax, uint64_t seed, uint64_t frame_offset,
                  const chest_injected inj, const chest_frame_outputs out, const float* __restrict__ state,
                  int RG, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const FrameParams& fp = *fpp;
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    {
        // Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the blocks that share one).  Give each XCD a
        // contiguous frame range: its L2 then holds one pattern group's tables and its writes are one sequential region.
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);        // bijective for any grid size
    }
    const int S = fp.S, T = fp.T, R = fp.R, G = fp.G;
    const float* rec = state + (size_t)b * frame_state_floats(S, R, T);
    const chest_frame_desc fd = descs[b];
    if (rec[ST_VALID] == 0.f) return;                            // K1 flagged the descriptor (uniform exit)
    const PatternDev pat = patterns[fd.pattern_slot];
    if (pat.slot == nullptr) {
        if (tid == 0) { atomicExch(status, 1); if (out.frame_status) out.frame_status[b] = 1; }
        return;
    }
    const bool want_est = out.H_ls != nullptr || out.H_mmse != nullptr || out.metrics != nullptr;
    if (want_est && pat.plan == nullptr) {
        if (tid == 0) { atomicExch(status, 2); if (out.frame_status) out.frame_status[b] = 2; }
        return;
    }
    const int P = pat.P;

    // ---- pilot mask: 16 resource elements per thread and one 16-B store, after peeling the row to 16-B alignment
    //      (G is rarely a multiple of 16, so most rows start misaligned) ---------------------------------------------
    if (out.pilot_mask != nullptr) {
        uint8_t* mrow = out.pilot_mask + (size_t)b * G;
        const int lead = min(G, (int)((16 - ((uintptr_t)mrow & 15)) & 15));
        const int nvec = (G - lead) >> 4, tail0 = lead + (nvec << 4);
        if (tid < lead) mrow[tid] = pat.slot[tid] < P ? 1 : 0;
        for (int v = tid; v < nvec; v += blockDim.x) {
            const uint16_t* sp = pat.slot + lead + (v << 4);
            uint32_t wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wq[q] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) wq[q] |= (sp[4 * q + j] < P ? 1u : 0u) << (8 * j);
            }
            *(uint4*)(mrow + lead + (v << 4)) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
        if (tid < G - tail0) mrow[tail0 + tid] = pat.slot[tail0 + tid] < P ? 1 : 0;
    }
    // the same mask as float32 0/1 (the dataset schema of run_phase3_dataset_generation.py:139): 4 elements per 16-B store
    if (out.pilot_mask_f32 != nullptr) {
        float* mrow = out.pilot_mask_f32 + (size_t)b * G;
        const int lead = min(G, (int)(((16 - ((uintptr_t)mrow & 15)) & 15) >> 2));
        const int nvec = (G - lead) >> 2, tail0 = lead + (nvec << 2);
        if (tid < lead) mrow[tid] = pat.slot[tid] < P ? 1.f : 0.f;
        for (int v = tid; v < nvec; v += blockDim.x) {
            const uint16_t* sp = pat.slot + lead + (v << 2);
            const v4f_t m = {sp[0] < P ? 1.f : 0.f, sp[1] < P ? 1.f : 0.f, sp[2] < P ? 1.f : 0.f, sp[3] < P ? 1.f : 0.f};
            __builtin_nontemporal_store(m, (v4f_t*)(mrow + lead + (v << 2)));
        }
        if (tid < G - tail0) mrow[tail0 + tid] = pat.slot[tail0 + tid] < P ? 1.f : 0.f;
    }

    SlabArgs A;
    A.fp = &fp; A.pat = &pat; A.prof = &fp.prof[fd.channel_type]; A.rec = rec;
    A.c_g = (const float2*)(rec + ST_SCAL);
    A.csum_g = A.c_g + (size_t)S * R * T * KTS;
    A.inj = &inj; A.out = &out;
    A.hp = (float2*)lds_raw;                                     // [Pmax*R rounded up to even]
    A.red = (float*)(A.hp + (((size_t)Pmax * R + 1) & ~(size_t)1));   // [32]
    A.alpha = A.red + 32;                                        // [8]
    A.slab = (float2*)(A.alpha + 8);                             // 16-B aligned: hp span is a multiple of 16 B, 40 floats = 160 B
    const uint64_t frame = frame_offset + (uint64_t)b;
    A.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)frame, (uint32_t)(frame >> 32)};
    A.b = b; A.RG = RG;
    switch (A.prof->ntaps) {                                     // workgroup-uniform dispatch on the profile
        case 5: slab_body<5>(A); break;                          // EPA at 15.36 MHz
        case 8: slab_body<8>(A); break;                          // EVA
        default: slab_body<KT>(A); break;                        // ETU / anything else (taps padded with zeros)
    }
}

size_t slab_lds_bytes(int nsc, int T, int R, int RG, int Pmax) {
    return ((((size_t)Pmax * R + 1) & ~(size_t)1) + (size_t)slab_elems(T, RG, nsc)) * sizeof(float2) + 40 * sizeof(float);
}

// Rx antennas per unit: the divisor of R that lets the most workgroups share a CU's 160 KiB of LDS (ties: the larger
// group = longer contiguous bursts); ctx->tune_rx_group overrides (chest_tune, tools only).
static int choose_rx_group(const chest_ctx* ctx, int Pmax) {
    const FrameParams& fp = ctx->fp;
    if (ctx->tune_rx_group > 0 && fp.R % ctx->tune_rx_group == 0) return ctx->tune_rx_group;
    int best = 1, best_wg = 0;
    for (int rg = 1; rg <= fp.R; ++rg) {
        if (fp.R % rg) continue;
        const size_t lds = slab_lds_bytes(fp.nsc, fp.T, fp.R, rg, Pmax);
        if (lds > 160 * 1024) continue;
        int wg = (int)((160 * 1024) / lds);
        if (wg > 4) wg = 4;                                      // register file: 4 waves per SIMD at <= 128 VGPRs
        if (wg >= best_wg) { best_wg = wg; best = rg; }
    }
    return best;
}

int launch_k2_slab(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, int Pmax, uint64_t seed,
                   uint64_t frame_offset, const chest_injected& inj, const chest_frame_outputs& out, const float* workspace) {
    const FrameParams& fp = ctx->fp;
    const int npair = (fp.nsc + 1) / 2, nwave = (npair + 63) / 64;
    if (nwave > 16) { set_error("frame kernel handles at most 2047 used subcarriers (got %d)", fp.nsc); return CHEST_EINVAL; }
    const int RG = choose_rx_group(ctx, Pmax);
    const size_t lds = slab_lds_bytes(fp.nsc, fp.T, fp.R, RG, Pmax);
    if (lds > 160 * 1024) { set_error("frame kernel needs %zu B of LDS (> 160 KiB): too many antennas, subcarriers or pilots", lds); return CHEST_EINVAL; }
    const dim3 grid((unsigned)B), block((unsigned)nwave * 64);
    if (nwave <= 5) {
        CHEST_LDS_OPT_IN(ctx, LDS_FRAME, (frame_slab_kernel<320, 4>), lds);
        hipLaunchKernelGGL((frame_slab_kernel<320, 4>), grid, block, lds, st, ctx->d_fp, descs, ctx->d_patterns, Pmax, seed, frame_offset,
                           inj, out, workspace, RG, ctx->d_status);
    } else if (nwave <= 10) {
        CHEST_LDS_OPT_IN(ctx, LDS_FRAME_640, (frame_slab_kernel<640, 3>), lds);
        hipLaunchKernelGGL((frame_slab_kernel<640, 3>), grid, block, lds, st, ctx->d_fp, descs, ctx->d_patterns, Pmax, seed, frame_offset,
                           inj, out, workspace, RG, ctx->d_status);
    } else {
        CHEST_LDS_OPT_IN(ctx, LDS_FRAME_1024, (frame_slab_kernel<1024, 4>), lds);
        hipLaunchKernelGGL((frame_slab_kernel<1024, 4>), grid, block, lds, st, ctx->d_fp, descs, ctx->d_patterns, Pmax, seed, frame_offset,
                           inj, out, workspace, RG, ctx->d_status);
    }
    return CHEST_OK;
}

}  // namespace chest
