// Sweep of write schedules for the frame layout: piece = `gs` symbols of one array written contiguously,
// arrays interleaved per piece (order 0) or array-at-a-time per frame (order 1 == gs 14).  Flat aligned 16-B stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, T = 2, R = 2, NSC = 599;
constexpr int nT = T * NSC / 2, nR = R * NSC / 2, nH = R * T * NSC / 2;
__device__ __forceinline__ void fill(float4* p, int n, float4 v) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v; }
__global__ void __launch_bounds__(1024) k(float4* tx, float4* rx, float4* H, float4* L, float4* M, int gs, int fpw) {
    for (int f = 0; f < fpw; ++f) {
        const int b = blockIdx.x * fpw + f;
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
        float4 *txb = tx + (size_t)b * S * nT, *rxb = rx + (size_t)b * S * nR, *Hb = H + (size_t)b * S * nH, *Lb = L + (size_t)b * S * nH, *Mb = M + (size_t)b * S * nH;
        for (int s = 0; s < S; s += gs) {
            const int n = min(gs, S - s);
            fill(txb + s * nT, n * nT, v); fill(Hb + s * nH, n * nH, v); fill(rxb + s * nR, n * nR, v); fill(Lb + s * nH, n * nH, v); fill(Mb + s * nH, n * nH, v);
        }
    }
}
__global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 6; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 6;
}
int main() {
    const int B = 10000;
    const size_t fT = (size_t)B * S * nT, fR = (size_t)B * S * nR, fH = (size_t)B * S * nH;
    float4* all; CK(hipMalloc(&all, (fT + fR + 3 * fH) * 16));
    float4 *tx = all, *rx = tx + fT, *H = rx + fR, *L = H + fH, *M = L + fH;
    const double bytes = (double)(fT + fR + 3 * fH) * 16;
    printf("flat fill 2048x256: %.1f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    printf("%-10s", "gs\\block");
    const int bts[] = {64, 128, 192, 256, 320, 512, 640, 1024};
    for (int bt : bts) printf("%8d", bt);
    printf("   (GB/s)\n");
    for (int gs : {1, 2, 4, 7, 14}) {
        printf("%-10d", gs);
        for (int bt : bts) printf("%8.0f", bytes / timeit([&] { k<<<B, bt>>>(tx, rx, H, L, M, gs, 1); }) * 1e-6);
        printf("\n"); fflush(stdout);
    }
    printf("two frames per WG (grid B/2):\n");
    for (int gs : {1, 14}) {
        printf("%-10d", gs);
        for (int bt : bts) printf("%8.0f", bytes / timeit([&] { k<<<B / 2, bt>>>(tx, rx, H, L, M, gs, 2); }) * 1e-6);
        printf("\n"); fflush(stdout);
    }
    printf("flat fill 2048x256: %.1f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    return 0;
}
