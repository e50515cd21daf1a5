// Does the cache policy of the stores change the HBM write rate of the frame kernel's write schedule?
// Flavours: plain, nt (__builtin_nontemporal_store), sc1, sc0 sc1 (write-through, inline asm).
// Patterns: flat grid-stride fill; "frame" = one 448-thread workgroup per frame, wave <-> symbol (w, w+7), 5 chunks per
// symbol, each chunk one 1-KiB store instruction into each of 8 rows (2 tx, 2 rx, 4 H), then the 8 estimate rows per symbol;
// optional VALU work between chunks (delay) to mimic the synthesis.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, ROW4 = 300;                       // row = 300 float4 = 4 800 B (real: 4 792 B)

typedef float v4f __attribute__((ext_vector_type(4)));
template <int F> __device__ __forceinline__ void st(float4* p, float4 v) {
    const v4f t = {v.x, v.y, v.z, v.w};
    if (F == 0) *p = v;
    else if (F == 1) __builtin_nontemporal_store(t, (v4f*)p);
    else if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
}

template <int F> __global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st<F>(p + i, make_float4(1.f, 2.f, 3.f, 4.f));
}

template <int F> __global__ void __launch_bounds__(448) frame(float4* tx, float4* rx, float4* H, float4* L, float4* M, int delay) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = (float)lane;
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        float4* rows[8];
        rows[0] = tx + ((size_t)(b * S + s) * 2 + 0) * ROW4; rows[1] = rows[0] + ROW4;
        rows[2] = rx + ((size_t)(b * S + s) * 2 + 0) * ROW4; rows[3] = rows[2] + ROW4;
        for (int q = 0; q < 4; ++q) rows[4 + q] = H + ((size_t)(b * S + s) * 4 + q) * ROW4;
        for (int j = 0; j < 5; ++j) {
            for (int d = 0; d < delay; ++d) acc = __builtin_amdgcn_sinf(acc) + 0.5f;
            const int i = j * 64 + lane;
            if (i < ROW4) {
                const float4 v = make_float4(acc, 2.f, 3.f, (float)b);
#pragma unroll
                for (int r = 0; r < 8; ++r) st<F>(rows[r] + i, v);
            }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4) {
                const float4 v = make_float4(acc, 5.f, 6.f, (float)b);
#pragma unroll
                for (int q = 0; q < 4; ++q) { st<F>(L + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); st<F>(M + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); }
            }
        }
    }
}

template <typename Fn> float timeit(Fn f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 8; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 8;
}

int main() {
    const int B = 10000;
    const size_t fT = (size_t)B * S * 2 * ROW4, fH = (size_t)B * S * 4 * ROW4, total = 2 * fT + 3 * fH;
    float4* all; CK(hipMalloc(&all, total * 16));
    float4 *tx = all, *rx = tx + fT, *H = rx + fT, *L = H + fH, *M = L + fH;
    const double bytes = (double)total * 16;
    auto rep = [&](const char* n, float ms) { printf("%-58s %7.3f ms  %7.1f GB/s\n", n, ms, bytes / ms * 1e-6); fflush(stdout); };
    const char* fn[4] = {"plain", "nt", "sc1", "sc0 sc1"};
    for (int rep_i = 0; rep_i < 2; ++rep_i) {
        rep("flat fill 8192x256, plain", timeit([&] { flat<0><<<8192, 256>>>(all, total); }));
        rep("flat fill 8192x256, nt", timeit([&] { flat<1><<<8192, 256>>>(all, total); }));
        rep("flat fill 8192x256, sc1", timeit([&] { flat<2><<<8192, 256>>>(all, total); }));
        rep("flat fill 8192x256, sc0 sc1", timeit([&] { flat<3><<<8192, 256>>>(all, total); }));
    }
    for (int delay : {0, 40, 120}) {
        char buf[128];
        snprintf(buf, sizeof buf, "frame schedule, delay %3d, %s", delay, fn[0]); rep(buf, timeit([&] { frame<0><<<B, 448>>>(tx, rx, H, L, M, delay); }));
        snprintf(buf, sizeof buf, "frame schedule, delay %3d, %s", delay, fn[1]); rep(buf, timeit([&] { frame<1><<<B, 448>>>(tx, rx, H, L, M, delay); }));
        snprintf(buf, sizeof buf, "frame schedule, delay %3d, %s", delay, fn[2]); rep(buf, timeit([&] { frame<2><<<B, 448>>>(tx, rx, H, L, M, delay); }));
        snprintf(buf, sizeof buf, "frame schedule, delay %3d, %s", delay, fn[3]); rep(buf, timeit([&] { frame<3><<<B, 448>>>(tx, rx, H, L, M, delay); }));
    }
    return 0;
}
