// Ceiling of the frame kernel's store pattern: same output layout (B frames x [tx, rx, H, H_ls, H_mmse]),
// trivial compute.  Variant A: thread<->subcarrier, 8-B stores row by row (what frame_kernel does).
// Variant B: flat 16-B stores over each frame's arrays.  Variant C: like A but all rows of one (s) written
// back-to-back from registers (store burst).  hipcc -O3 --offload-arch=gfx950 store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int S = 14, T = 2, R = 2, NSC = 599;

__global__ void __launch_bounds__(1024) storeA(float2* tx, float2* rx, float2* H, float2* Hls, float2* Hmm, int passes) {
    const int b = blockIdx.x;
    const size_t fH = (size_t)b * S * R * T * NSC;
    for (int k = threadIdx.x; k < NSC; k += blockDim.x) {
        const float2 v = make_float2((float)k, (float)b);
        if (passes & 1)
        for (int s = 0; s < S; ++s) {
            for (int t = 0; t < T; ++t) tx[((size_t)b * S + s) * T * NSC + (size_t)t * NSC + k] = v;
            for (int r = 0; r < R; ++r) {
                for (int t = 0; t < T; ++t) H[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v;
                rx[((size_t)b * S * R + (size_t)s * R + r) * NSC + k] = v;
            }
        }
        if (passes & 2)
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < T; ++t) {
                    Hls[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v;
                    Hmm[fH + ((size_t)(s * R + r) * T + t) * NSC + k] = v;
                }
    }
}

__global__ void __launch_bounds__(1024) storeB(float4* tx, float4* rx, float4* H, float4* Hls, float4* Hmm) {
    const int b = blockIdx.x;
    const int nH = S * R * T * NSC / 2, nT = S * T * NSC / 2, nR = S * R * NSC / 2;   // float4 = 2 complex
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
    for (int i = threadIdx.x; i < nT; i += blockDim.x) tx[(size_t)b * nT + i] = v;
    for (int i = threadIdx.x; i < nR; i += blockDim.x) rx[(size_t)b * nR + i] = v;
    for (int i = threadIdx.x; i < nH; i += blockDim.x) H[(size_t)b * nH + i] = v;
    for (int i = threadIdx.x; i < nH; i += blockDim.x) Hls[(size_t)b * nH + i] = v;
    for (int i = threadIdx.x; i < nH; i += blockDim.x) Hmm[(size_t)b * nH + i] = v;
}

// Variant C: what frame_kernel v3 does: thread <-> two adjacent subcarriers, wave covers 126 (+2 overlap),
// 16-B stores, rows alternate between aligned and 8-B shifted (shifted rows take k0+2 from lane+1).
__device__ __forceinline__ void store_pair(float2* base, unsigned row, int par, int k0, bool storer, bool first, float2 v0, float2 v1, float2 n0) {
    if (par == 0) {
        if (storer) {
            if (k0 + 1 < NSC) *(float4*)(base + row + k0) = make_float4(v0.x, v0.y, v1.x, v1.y);
            else if (k0 < NSC) base[row + k0] = v0;
        }
    } else {
        if (storer) {
            if (k0 + 2 < NSC) *(float4*)(base + row + k0 + 1) = make_float4(v1.x, v1.y, n0.x, n0.y);
            else if (k0 + 1 < NSC) base[row + k0 + 1] = v1;
        }
        if (first) base[row] = v0;
    }
}
__global__ void __launch_bounds__(1024) storeC(float2* tx, float2* rx, float2* H, float2* Hls, float2* Hmm) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    float2* txb = tx + (size_t)b * S * T * NSC; float2* rxb = rx + (size_t)b * S * R * NSC;
    float2* Hb = H + (size_t)b * S * R * T * NSC; float2* Lb = Hls + (size_t)b * S * R * T * NSC; float2* Mb = Hmm + (size_t)b * S * R * T * NSC;
    for (int wg = threadIdx.x >> 6; wg < (NSC + 125) / 126; wg += blockDim.x >> 6) {
        const int k0 = wg * 126 + 2 * lane; const bool storer = lane < 63, first = (wg == 0 && lane == 0);
        const float2 v0 = make_float2((float)k0, (float)b), v1 = make_float2((float)k0 + 1, (float)b);
        const float2 n0 = make_float2(__shfl_down(v0.x, 1, 64), __shfl_down(v0.y, 1, 64));
        for (int s = 0; s < S; ++s) {
            for (int t = 0; t < T; ++t) { unsigned row = (s * T + t) * NSC; store_pair(txb, row, row & 1, k0, storer, first, v0, v1, n0); }
            for (int r = 0; r < R; ++r) {
                for (int t = 0; t < T; ++t) { unsigned row = ((s * R + r) * T + t) * NSC; store_pair(Hb, row, row & 1, k0, storer, first, v0, v1, n0); }
                unsigned row = (s * R + r) * NSC; store_pair(rxb, row, row & 1, k0, storer, first, v0, v1, n0);
            }
        }
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < R; ++r)
                for (int t = 0; t < T; ++t) { unsigned row = ((s * R + r) * T + t) * NSC; store_pair(Lb, row, row & 1, k0, storer, first, v0, v1, n0); store_pair(Mb, row, row & 1, k0, storer, first, v0, v1, n0); }
    }
}

int main(int argc, char** argv) {
    const int B = 10000;
    const size_t nH = (size_t)B * S * R * T * NSC, nT = (size_t)B * S * T * NSC, nR = (size_t)B * S * R * NSC;
    float2 *tx, *rx, *H, *Hls, *Hmm;
    CK(hipMalloc(&tx, nT * 8)); CK(hipMalloc(&rx, nR * 8)); CK(hipMalloc(&H, nH * 8)); CK(hipMalloc(&Hls, nH * 8)); CK(hipMalloc(&Hmm, nH * 8));
    const double bytes = (double)(nT + nR + 3 * nH) * 8;
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    int blocks[] = {128, 256, 640};
    for (int bi = 0; bi < 3; ++bi) {
        const int bt = blocks[bi];
        for (int variant = 0; variant < 2; ++variant) {
            for (int it = 0; it < 3; ++it) {
                if (variant == 0) storeA<<<B, bt>>>(tx, rx, H, Hls, Hmm, 3);
                else storeB<<<B, bt>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)Hls, (float4*)Hmm);
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int it = 0; it < 10; ++it) {
                if (variant == 0) storeA<<<B, bt>>>(tx, rx, H, Hls, Hmm, 3);
                else storeB<<<B, bt>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)Hls, (float4*)Hmm);
            }
            CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
            float ms; CK(hipEventElapsedTime(&ms, a, c)); ms /= 10;
            printf("block %4d  %s  %7.3f ms  %7.1f GB/s\n", bt, variant == 0 ? "A: 8-B row-wise (thread<->k)" : "B: 16-B flat              ", ms, bytes / ms * 1e-6);
        }
    }
    for (int bt : {64, 128, 320}) {
        for (int it = 0; it < 3; ++it) storeC<<<B, bt>>>(tx, rx, H, Hls, Hmm);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int it = 0; it < 10; ++it) storeC<<<B, bt>>>(tx, rx, H, Hls, Hmm);
        CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
        float ms; CK(hipEventElapsedTime(&ms, a, c)); ms /= 10;
        printf("block %4d  C: 16-B row-wise pairs (v3)    %7.3f ms  %7.1f GB/s\n", bt, ms, bytes / ms * 1e-6);
    }
    return 0;
}
