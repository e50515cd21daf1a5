// How does HBM write bandwidth depend on the size of contiguous pieces and on the workgroup->address schedule?
// 10 000 "frames", 5 arrays (tx, rx: 134 176 B/frame; H, H_ls, H_mmse: 268 352 B/frame), flat aligned 16-B stores only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, T = 2, R = 2, NSC = 599;
constexpr int nT = T * NSC / 2, nR = R * NSC / 2, nH = R * T * NSC / 2;     // float4 per symbol slice

__device__ __forceinline__ void fill(float4* p, int n, float4 v) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v; }

// mode 0: whole arrays per frame (5 long streams)   1: per-symbol slices interleaved   2: per-symbol, sim then est
// persistent != 0: grid = persistent workgroups, each walks a contiguous range of frames
__global__ void __launch_bounds__(1024) k(float4* tx, float4* rx, float4* H, float4* L, float4* M, int B, int mode, int persistent) {
    int b0 = blockIdx.x, b1 = blockIdx.x + 1;
    if (persistent) { const int per = (B + gridDim.x - 1) / gridDim.x; b0 = blockIdx.x * per; b1 = min(B, b0 + per); }
    for (int b = b0; b < b1; ++b) {
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
        float4 *txb = tx + (size_t)b * S * nT, *rxb = rx + (size_t)b * S * nR, *Hb = H + (size_t)b * S * nH, *Lb = L + (size_t)b * S * nH, *Mb = M + (size_t)b * S * nH;
        if (mode == 0) { fill(txb, S * nT, v); fill(rxb, S * nR, v); fill(Hb, S * nH, v); fill(Lb, S * nH, v); fill(Mb, S * nH, v); }
        else if (mode == 1) { for (int s = 0; s < S; ++s) { fill(txb + s * nT, nT, v); fill(Hb + s * nH, nH, v); fill(rxb + s * nR, nR, v); fill(Lb + s * nH, nH, v); fill(Mb + s * nH, nH, v); } }
        else { for (int s = 0; s < S; ++s) { fill(txb + s * nT, nT, v); fill(Hb + s * nH, nH, v); fill(rxb + s * nR, nR, v); }
               for (int s = 0; s < S; ++s) { fill(Lb + s * nH, nH, v); fill(Mb + s * nH, nH, v); } }
    }
}
// one big flat fill of a single 10.9 GB region, grid-stride (classic ceiling)
__global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 8; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 8;
}
int main() {
    const int B = 10000;
    const size_t fT = (size_t)B * S * nT, fR = (size_t)B * S * nR, fH = (size_t)B * S * nH;
    float4* all; CK(hipMalloc(&all, (fT + fR + 3 * fH) * 16));
    float4 *tx = all, *rx = tx + fT, *H = rx + fR, *L = H + fH, *M = L + fH;
    const double bytes = (double)(fT + fR + 3 * fH) * 16;
    auto rep = [&](const char* n, float ms) { printf("%-64s %7.3f ms  %7.1f GB/s\n", n, ms, bytes / ms * 1e-6); fflush(stdout); };
    rep("flat grid-stride fill, 2048 x 256", timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }));
    rep("flat grid-stride fill, 8192 x 256", timeit([&] { flat<<<8192, 256>>>(all, fT + fR + 3 * fH); }));
    const char* mn[3] = {"whole arrays per frame", "symbol slices interleaved", "symbol slices, sim then est"};
    for (int mode = 0; mode < 3; ++mode)
        for (int bt : {256, 640}) {
            char buf[128];
            snprintf(buf, sizeof buf, "WG per frame, %s, block %d", mn[mode], bt);
            rep(buf, timeit([&] { k<<<B, bt>>>(tx, rx, H, L, M, B, mode, 0); }));
        }
    for (int grid : {256, 512, 1024, 2048})
        for (int mode : {0, 1}) {
            char buf[128];
            snprintf(buf, sizeof buf, "persistent %d WGs x 640, %s", grid, mn[mode]);
            rep(buf, timeit([&] { k<<<grid, 640>>>(tx, rx, H, L, M, B, mode, 1); }));
        }
    return 0;
}
