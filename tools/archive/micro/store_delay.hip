// Does spreading a wave's stores out in time (compute between them) cost HBM write bandwidth?
// Wave per symbol (14-wave workgroups, one per frame).  Burst = how many bytes a wave writes back-to-back to
// consecutive addresses before it "computes" (busy FMA loop of `work` iterations).  Same total bytes every time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, T = 2, R = 2, NSC = 599;
constexpr int ROW4 = NSC / 2;                      // float4 per row (the odd element is ignored here)
__device__ __forceinline__ float spin(float x, int n) { for (int i = 0; i < n; ++i) x = fmaf(x, 1.000001f, 1e-7f); return x; }

// order 0: chunk-outer (for each 1-KiB column chunk touch all rows of the symbol), compute after every chunk sweep
// order 1: row-major (whole rows back to back), compute after every row
// order 2: row-major, compute after every array slice (all rows of one array of the symbol)
__global__ void __launch_bounds__(1024) k(float4* tx, float4* rx, float4* H, float4* L, float4* M, int order, int work, float* sink) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float acc = (float)lane;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
    float4* arr[5] = {tx + (size_t)b * S * T * ROW4, H + (size_t)b * S * R * T * ROW4, rx + (size_t)b * S * R * ROW4,
                      L + (size_t)b * S * R * T * ROW4, M + (size_t)b * S * R * T * ROW4};
    const int rows[5] = {T, R * T, R, R * T, R * T};
    for (int s = wave; s < S; s += nw) {
        if (order == 0) {
            for (int phase = 0; phase < 2; ++phase)
                for (int j = 0; j < 5; ++j) {
                    for (int a = phase ? 3 : 0; a < (phase ? 5 : 3); ++a)
                        for (int r = 0; r < rows[a]; ++r) { const int i = j * 64 + lane; if (i < ROW4) arr[a][(s * rows[a] + r) * ROW4 + i] = v; }
                    acc = spin(acc, work * (phase ? 4 : 8));             // work per chunk sweep ~ rows touched
                }
        } else {
            for (int a = 0; a < 5; ++a) {
                for (int r = 0; r < rows[a]; ++r) {
                    for (int i = lane; i < ROW4; i += 64) arr[a][(s * rows[a] + r) * ROW4 + i] = v;
                    if (order == 1) acc = spin(acc, work * 5);
                }
                if (order == 2) acc = spin(acc, work * 5 * rows[a]);
            }
        }
    }
    if (acc == 12345.f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 5;
}
int main() {
    const int B = 10000;
    const size_t fT = (size_t)B * S * T * ROW4, fR = (size_t)B * S * R * ROW4, fH = (size_t)B * S * R * T * ROW4;
    float4* all; CK(hipMalloc(&all, (fT + fR + 3 * fH) * 16));
    float* sink; CK(hipMalloc(&sink, 4));
    float4 *tx = all, *rx = tx + fT, *H = rx + fR, *L = H + fH, *M = L + fH;
    const double bytes = (double)(fT + fR + 3 * fH) * 16;
    printf("flat fill: %.0f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    const char* on[3] = {"chunk-outer, compute per chunk sweep", "row-major, compute per row", "row-major, compute per array slice"};
    for (int bt : {896, 448}) {
        printf("block %d:  work ->", bt);
        const int works[] = {0, 50, 100, 200, 400};
        for (int w : works) printf("%9d", w);
        printf("   (ms per launch)\n");
        for (int order = 0; order < 3; ++order) {
            printf("  %-40s", on[order]);
            for (int w : works) printf("%9.3f", timeit([&] { k<<<B, bt>>>(tx, rx, H, L, M, order, w, sink); }));
            printf("\n"); fflush(stdout);
        }
    }
    printf("flat fill: %.0f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    return 0;
}
