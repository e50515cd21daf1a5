// Which aspect of a store pattern costs bandwidth on MI355X?  Same bytes (10 000 2x2 frames, 5 arrays), no compute.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, T = 2, R = 2, NSC = 599;

template <int MODE>
__device__ __forceinline__ void sp(float2* base, unsigned row, int par, int k0, int lane, bool first, float2 v0, float2 v1, float2 n0) {
    // MODE 0: v3 (parity, lane 63 idle, singles)   1: parity forced 0   2: v3 without single-lane stores
    // MODE 3: all 64 lanes store aligned pairs (k0 = 128w + 2 lane), no parity handling
    if (MODE == 1) par = 0;
    const bool storer = (MODE == 3) ? true : lane < 63;
    if (par == 0) {
        if (storer) {
            if (k0 + 1 < NSC) *(float4*)(base + row + k0) = make_float4(v0.x, v0.y, v1.x, v1.y);
            else if (MODE != 2 && k0 < NSC) base[row + k0] = v0;
        }
    } else {
        if (storer) {
            if (k0 + 2 < NSC) *(float4*)(base + row + k0 + 1) = make_float4(v1.x, v1.y, n0.x, n0.y);
            else if (MODE != 2 && k0 + 1 < NSC) base[row + k0 + 1] = v1;
        }
        if (MODE != 2 && first) base[row] = v0;
    }
}
template <int MODE>
__global__ void __launch_bounds__(1024) storeC(float2* tx, float2* rx, float2* H, float2* Hls, float2* Hmm) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    float2* txb = tx + (size_t)b * S * T * NSC; float2* rxb = rx + (size_t)b * S * R * NSC;
    float2* Hb = H + (size_t)b * S * R * T * NSC; float2* Lb = Hls + (size_t)b * S * R * T * NSC; float2* Mb = Hmm + (size_t)b * S * R * T * NSC;
    const int per = (MODE == 3) ? 128 : 126;
    for (int wg = threadIdx.x >> 6; wg < (NSC + per - 1) / per; wg += blockDim.x >> 6) {
        const int k0 = wg * per + 2 * lane; const bool first = (wg == 0 && lane == 0);
        const float2 v0 = make_float2((float)k0, (float)b), v1 = make_float2((float)k0 + 1, (float)b);
        const float2 n0 = make_float2(__shfl_down(v0.x, 1, 64), __shfl_down(v0.y, 1, 64));
        for (int s = 0; s < S; ++s) {
            for (int t = 0; t < T; ++t) { unsigned row = (s * T + t) * NSC; sp<MODE>(txb, row, row & 1, k0, lane, first, v0, v1, n0); }
            for (int r = 0; r < R; ++r) {
                for (int t = 0; t < T; ++t) { unsigned row = ((s * R + r) * T + t) * NSC; sp<MODE>(Hb, row, row & 1, k0, lane, first, v0, v1, n0); }
                unsigned row = (s * R + r) * NSC; sp<MODE>(rxb, row, row & 1, k0, lane, first, v0, v1, n0);
            }
        }
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < T; ++t) { unsigned row = ((s * R + r) * T + t) * NSC; sp<MODE>(Lb, row, row & 1, k0, lane, first, v0, v1, n0); sp<MODE>(Mb, row, row & 1, k0, lane, first, v0, v1, n0); }
    }
}
// V6: per symbol, flat aligned 16-B stores of each array's contiguous symbol slice (what LDS staging would emit)
__global__ void __launch_bounds__(1024) storeSlices(float4* tx, float4* rx, float4* H, float4* Hls, float4* Hmm, int order) {
    const int b = blockIdx.x;
    const int nT = T * NSC / 2, nR = R * NSC / 2, nH = R * T * NSC / 2;      // float4 per symbol slice
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
    float4* txb = tx + (size_t)b * S * nT; float4* rxb = rx + (size_t)b * S * nR;
    float4* Hb = H + (size_t)b * S * nH; float4* Lb = Hls + (size_t)b * S * nH; float4* Mb = Hmm + (size_t)b * S * nH;
    for (int s = 0; s < S; ++s) {
        for (int i = threadIdx.x; i < nT; i += blockDim.x) txb[s * nT + i] = v;
        for (int i = threadIdx.x; i < nH; i += blockDim.x) Hb[s * nH + i] = v;
        for (int i = threadIdx.x; i < nR; i += blockDim.x) rxb[s * nR + i] = v;
        if (order == 0) { for (int i = threadIdx.x; i < nH; i += blockDim.x) { Lb[s * nH + i] = v; Mb[s * nH + i] = v; } }
    }
    if (order == 1)
        for (int s = 0; s < S; ++s)
            for (int i = threadIdx.x; i < nH; i += blockDim.x) { Lb[s * nH + i] = v; Mb[s * nH + i] = v; }
}
// V7: 8-B row-wise (frame_kernel v2 pattern) for reference, one subcarrier per thread
__global__ void __launch_bounds__(1024) storeA(float2* tx, float2* rx, float2* H, float2* Hls, float2* Hmm) {
    const int b = blockIdx.x;
    const size_t fH = (size_t)b * S * R * T * NSC;
    for (int k = threadIdx.x; k < NSC; k += blockDim.x) {
        const float2 v = make_float2((float)k, (float)b);
        for (int s = 0; s < S; ++s) {
            for (int t = 0; t < T; ++t) tx[((size_t)b * S + s) * T * NSC + (size_t)t * NSC + k] = v;
            for (int r = 0; r < R; ++r) {
                for (int t = 0; t < T; ++t) H[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v;
                rx[((size_t)b * S * R + (size_t)s * R + r) * NSC + k] = v;
            }
        }
        for (int s = 0; s < S; ++s) for (int r = 0; r < R; ++r) for (int t = 0; t < T; ++t) {
            Hls[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v; Hmm[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v; }
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 10;
}
int main() {
    const int B = 10000;
    const size_t nH = (size_t)B * S * R * T * NSC, nT = (size_t)B * S * T * NSC, nR = (size_t)B * S * R * NSC;
    float2 *tx, *rx, *H, *Hls, *Hmm;
    CK(hipMalloc(&tx, nT * 8)); CK(hipMalloc(&rx, nR * 8)); CK(hipMalloc(&H, nH * 8)); CK(hipMalloc(&Hls, nH * 8)); CK(hipMalloc(&Hmm, nH * 8));
    const double bytes = (double)(nT + nR + 3 * nH) * 8;
    auto rep = [&](const char* n, int bt, float ms) { printf("%-46s block %4d  %7.3f ms  %7.1f GB/s\n", n, bt, ms, bytes / ms * 1e-6); };
    for (int bt : {320, 640}) {
        rep("A  8-B row-wise, 1 sc/thread", bt, timeit([&] { storeA<<<B, bt>>>(tx, rx, H, Hls, Hmm); }));
        rep("C0 16-B pairs v3 (parity, lane63 idle, singles)", bt, timeit([&] { storeC<0><<<B, bt>>>(tx, rx, H, Hls, Hmm); }));
        rep("C1 16-B pairs, parity forced 0", bt, timeit([&] { storeC<1><<<B, bt>>>(tx, rx, H, Hls, Hmm); }));
        rep("C2 16-B pairs v3 without single-lane stores", bt, timeit([&] { storeC<2><<<B, bt>>>(tx, rx, H, Hls, Hmm); }));
        rep("C3 16-B pairs, all 64 lanes, parity ignored", bt, timeit([&] { storeC<3><<<B, bt>>>(tx, rx, H, Hls, Hmm); }));
        rep("S0 flat symbol slices, all arrays per symbol", bt, timeit([&] { storeSlices<<<B, bt>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)Hls, (float4*)Hmm, 0); }));
        rep("S1 flat symbol slices, estimates afterwards", bt, timeit([&] { storeSlices<<<B, bt>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)Hls, (float4*)Hmm, 1); }));
    }
    return 0;
}
