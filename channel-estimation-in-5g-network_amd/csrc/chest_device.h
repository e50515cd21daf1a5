// Device-side helpers shared by the gfx950 kernels: Philox4x32-10 counter RNG, complex math,
// wave64 / workgroup reductions.  CDNA4 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chest.h"

namespace chest {

// ---- random streams --------------------------------------------------------------------------
// counter = (index, stream, frame_lo, frame_hi), key = (seed_lo, seed_hi).  oracle/philox.py mirrors this.
enum : uint32_t { STREAM_OSC = 1, STREAM_SYM = 2, STREAM_NOISE = 3 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

struct RngKey { uint32_t seed_lo, seed_hi, frame_lo, frame_hi; };

__device__ __forceinline__ u32x4 draw(const RngKey& k, uint32_t stream, uint32_t index) {
    return philox4x32_10(index, stream, k.frame_lo, k.frame_hi, k.seed_lo, k.seed_hi);
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

// exp(2*pi*j*turns), argument reduced to [0,1) first so sincospif sees an exact small argument.
__device__ __forceinline__ float2 cis_turns(float turns) {
    const float f = turns - floorf(turns);
    float s, c;
    sincospif(2.0f * f, &s, &c);
    return make_float2(c, s);
}

// Box-Muller from two uniform words: one complex standard normal (re, im independent N(0,1)).
__device__ __forceinline__ float2 gauss_pair(uint32_t a, uint32_t b) {
    const float mag = sqrtf(-2.0f * logf(u01_open0(a)));
    float s, c;
    sincospif(2.0f * u01(b), &s, &c);
    return make_float2(mag * c, mag * s);
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
