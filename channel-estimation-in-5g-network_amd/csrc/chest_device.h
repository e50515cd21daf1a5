// Device-side helpers shared by the gfx950 kernels: Philox4x32-10 counter RNG, complex math,
// turn-based hardware sin/cos, wave64 / workgroup reductions.  CDNA4 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chest.h"

namespace chest {

// ---- random streams --------------------------------------------------------------------------
// counter = (index, stream, frame_lo, frame_hi), key = (seed_lo, seed_hi).  oracle/philox.py mirrors this.
//   STREAM_OSC   index = ((path*T + tx)*R + rx)*20 + n      word x -> angle uniform, y -> phase uniform
//   STREAM_SYM   index = s*npair + (k>>1)                   word (k&1) -> symbol phase uniform of RE (s,k)
//   STREAM_NOISE index = (s*R + r)*npair + (k>>1)           words (x,y) -> k even, (z,w) -> k odd   (npair = ceil(nsc/2))
//   STREAM_PATTERN (frame words = pattern id) index = g>>2, word (g&3) -> selection key of resource element g: the P
//                smallest keys are the pilots of a device-drawn pattern (chest_patterns_build)
enum : uint32_t { STREAM_OSC = 1, STREAM_SYM = 2, STREAM_NOISE = 3, STREAM_PATTERN = 4 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#ifndef CHEST_PHILOX_ROUNDS
#define CHEST_PHILOX_ROUNDS 10
#endif
#ifdef CHEST_FAKE_RNG      // timing experiments only: NOT a usable generator
    const uint32_t h = (c0 * 2654435761u) ^ (c1 * 40503u) ^ c2 ^ k0;
    return {h, h * 1664525u + 1013904223u, h ^ 0x9E3779B9u, h * 22695477u + 1u};
#endif
#pragma unroll
    for (int i = 0; i < CHEST_PHILOX_ROUNDS; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;     // one v_mad_u64_u32 gives hi and lo
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0; c1 = (uint32_t)p1;
        c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

struct RngKey { uint32_t seed_lo, seed_hi, frame_lo, frame_hi; };

__device__ __forceinline__ u32x4 draw(const RngKey& k, uint32_t stream, uint32_t index) {
    return philox4x32_10(index, stream, k.frame_lo, k.frame_hi, k.seed_lo, k.seed_hi);
}
__device__ __forceinline__ uint32_t word(const u32x4& v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
// 24-bit uniforms: [0,1) and (0,1]
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
__device__ __forceinline__ float u01_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f; }

// ---- complex helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ void cfma(float2& acc, float2 a, float2 b) {   // acc += a*b
    acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y);
}
__device__ __forceinline__ float cabs2(float2 a) { return fmaf(a.x, a.x, a.y * a.y); }

// exp(2*pi*j*turns).  v_sin_f32 / v_cos_f32 take their argument in revolutions; on [0,1) their
// absolute error is 1.3e-7 (measured on MI355X over all 2^24 inputs, tools/micro/intrinsic_accuracy.hip),
// i.e. float32 rounding, at 2 instructions instead of ~40 for sincospif.
__device__ __forceinline__ float2 cis_unit(float f) {          // f already in [0,1)
    return make_float2(__builtin_amdgcn_cosf(f), __builtin_amdgcn_sinf(f));
}
__device__ __forceinline__ float2 cis_turns(float turns) {     // any real number of turns
    return cis_unit(__builtin_amdgcn_fractf(turns));
}

// Box-Muller from two uniform words: one complex standard normal (re, im independent N(0,1)).
// v_log_f32 (log2) * ln2: relative error 1.6e-7 on (0,1]; v_sqrt_f32 9e-8 (same measurement).
__device__ __forceinline__ float2 gauss_pair(uint32_t a, uint32_t b) {
    const float l2 = __builtin_amdgcn_logf(u01_open0(a));                  // log2(u) <= 0
    const float mag = __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2);    // sqrt(-2 ln u)
    const float2 e = cis_unit(u01(b));
    return make_float2(mag * e.x, mag * e.y);
}

// h = y / (x + 1e-12)  (reference src/baseline_estimators.py:40, :110, :171)
__device__ __forceinline__ float2 pilot_divide(float2 y, float2 x) {
    const float2 xe = make_float2(x.x + 1e-12f, x.y);
    const float inv = __builtin_amdgcn_rcpf(cabs2(xe));
    return make_float2((y.x * xe.x + y.y * xe.y) * inv, (y.y * xe.x - y.x * xe.y) * inv);
}

// ---- interpolation plan -----------------------------------------------------------------------------
// One 16-B record per resource element: barycentric weights 0/1 (weight 2 = 1 - w0 - w1 as scipy's
// LinearNDInterpolator computes it; 0 when the element lies outside the convex hull), the three pilot
// ordinals, and a "valid" flag.  'nearest' plans are (w0, w1) = (1, 0) with all three ordinals equal.
struct PlanRec { float w0, w1; uint32_t i01, i2v; };

// Pointers read out of the pattern table are generic to the compiler and would be dereferenced with FLAT instructions, which
// count against the LDS / scalar-load counter as well (a wait for an LDS read then also waits for the outstanding prefetch from
// HBM).  They always point to device memory: say so.
#define CHEST_GLOBAL __attribute__((address_space(1)))
typedef const CHEST_GLOBAL uint16_t* gptr_u16;
typedef const CHEST_GLOBAL int32_t* gptr_i32;
typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
typedef const CHEST_GLOBAL v2u_t* gptr_bits;
// draw slot of resource element g from the pattern's pilot bit table (chest_internal.h: PatternDev::bits): < P = pilot ordinal
__device__ __forceinline__ uint32_t draw_slot(gptr_bits bits, int g, int P) {
    const v2u_t w = bits[g >> 5];
    const uint32_t sh = (uint32_t)g & 31u;
    const uint32_t rank = w.y + (uint32_t)__popc(w.x & ((1u << sh) - 1u));
    return ((w.x >> sh) & 1u) ? rank : (uint32_t)P + (uint32_t)g - rank;
}
// bits g .. g + 15 of the pilot mask (g + 15 may run into the next word; the table carries one zero word of padding)
__device__ __forceinline__ uint32_t mask_bits16(gptr_bits bits, int g) {
    const uint64_t ww = (uint64_t)bits[g >> 5].x | ((uint64_t)bits[(g >> 5) + 1].x << 32);
    return (uint32_t)(ww >> (g & 31)) & 0xFFFFu;
}
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
typedef const CHEST_GLOBAL v4u_t* gptr_plan;
__device__ __forceinline__ PlanRec ld_plan(gptr_plan plan, int i) {     // one 16-B global load
    const v4u_t v = plan[i];
    return PlanRec{__uint_as_float(v.x), __uint_as_float(v.y), v.z, v.w};
}

// Compact plans (chest_patterns_build method 2): the plan names a TRIANGLE per resource element (u16) and holds one 16-B
// record per triangle -- {i0 | i1 << 16, i2 | pa << 16, pb | pc << 16, 1 / (2 area)} with p = symbol << 11 | subcarrier of
// the three vertices (pa = 0xFFFF: outside the hull) -- 67 KB per 14 x 599 pattern instead of 151 KB.  The reader evaluates
// the two edge functions at its own (symbol, subcarrier): w0 = orient(b, c, q) / (2 area), w1 = orient(c, a, q) / (2 area),
// exactly the integers the full plan's weights were divided from.  `raw` carries the record's four words in a PlanRec.
template <bool CP>
__device__ __forceinline__ PlanRec plan_rec(const PlanRec& raw, int s, int k) {
    if constexpr (!CP) return raw;
    else {
        const uint32_t d0 = __float_as_uint(raw.w0), d1 = __float_as_uint(raw.w1), d2 = raw.i01;
        const int pa = (int)(d1 >> 16), pb = (int)(d2 & 0xFFFFu), pc = (int)(d2 >> 16);
        const int ax = pa >> 11, ay = pa & 0x7FF, bx = pb >> 11, by = pb & 0x7FF, cx = pc >> 11, cy = pc & 0x7FF;
        const int o0 = (cx - bx) * (k - by) - (cy - by) * (s - bx);
        const int o1 = (ax - cx) * (k - cy) - (ay - cy) * (s - cx);
        const float rden = __uint_as_float(raw.i2v);             // 0 for the "outside" record
        return PlanRec{(float)o0 * rden, (float)o1 * rden, d0, (d1 & 0xFFFFu) | (pa != 0xFFFF ? 1u << 16 : 0u)};
    }
}

__device__ __forceinline__ float2 interp(const PlanRec& p, const float2* __restrict__ hp, int stride, int r) {
    const float w2 = (p.i2v >> 16) ? (1.0f - p.w0 - p.w1) : 0.0f;
    const float2 a0 = hp[(p.i01 & 0xFFFFu) * stride + r], a1 = hp[(p.i01 >> 16) * stride + r],
                 a2 = hp[(p.i2v & 0xFFFFu) * stride + r];
    return make_float2(fmaf(w2, a2.x, fmaf(p.w1, a1.x, p.w0 * a0.x)), fmaf(w2, a2.y, fmaf(p.w1, a1.y, p.w0 * a0.y)));
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the whole workgroup; result broadcast to every thread.  `red` = >= 17 floats of LDS.
// Leaves a barrier behind it, so `red` may be reused immediately.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (wave == 0) {
        float t = lane < nw ? red[lane] : 0.0f;
        t = wave_sum(t);
        if (lane == 0) red[16] = t;
    }
    __syncthreads();
    const float r = red[16];
    __syncthreads();
    return r;
}

}  // namespace chest
