// Two tensors written concurrently at identical offsets (as H_ls and H_mmse are): how does the write rate depend on the
// distance between their base addresses?  One 24-GiB buffer, A at its start, B at A + 2.75 GiB + j * step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, ROW4 = 300, B = 10000;
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float4* p, float4 v) { const v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4f*)p); }
__global__ void __launch_bounds__(448) pair(float4* L, float4* M) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 v = make_float4((float)lane, 2.f, 3.f, (float)b);
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4)
                for (int q = 0; q < 4; ++q) { st(L + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); st(M + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); }
        }
    }
}
int main(int argc, char** argv) {
    const size_t fH = (size_t)B * S * 4 * ROW4 * 16;           // 2.688 GB
    const size_t total = (size_t)24 << 30;
    char* buf; CK(hipMalloc(&buf, total));
    printf("buffer at %p\n", (void*)buf);
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    auto timeit = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(c); hipEventSynchronize(c); float ms; hipEventElapsedTime(&ms, a, c); return ms / 5; };
    const size_t base_gap = ((size_t)11 << 28);                 // 2.75 GiB
    for (size_t step : {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20}) {
        for (int j = 0; j < 40; ++j) {
            const size_t off = base_gap + j * step;
            if (off + fH > total) break;
            float4 *L = (float4*)buf, *M = (float4*)(buf + off);
            const float t = timeit([&] { pair<<<B, 448>>>(L, M); });
            printf("step %4zu MiB  j %2d  B-A = %#14zx  %6.3f ms  %5.2f TB/s\n", step >> 20, j, off, t, 2.0 * fH / t * 1e-9);
            fflush(stdout);
        }
    }
    return 0;
}
