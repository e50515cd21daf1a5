// Placement lottery vs allocation flavour: five tensors per set, frame-kernel store schedule (see store_placement.hip),
// allocated with hipMalloc or with hipExtMallocWithFlags(hipDeviceMallocContiguous) -- physically contiguous memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int S = 14, ROW4 = 300, B = 10000;
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float4* p, float4 v) { const v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4f*)p); }
__global__ void __launch_bounds__(448) frame(float4* tx, float4* rx, float4* H, float4* L, float4* M) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 v = make_float4((float)lane, 2.f, 3.f, (float)b);
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4) {
                for (int r = 0; r < 2; ++r) st(tx + ((size_t)(b * S + s) * 2 + r) * ROW4 + i, v);
                for (int r = 0; r < 2; ++r) st(rx + ((size_t)(b * S + s) * 2 + r) * ROW4 + i, v);
                for (int q = 0; q < 4; ++q) st(H + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v);
            }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4)
                for (int q = 0; q < 4; ++q) { st(L + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); st(M + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v); }
        }
    }
}
int main() {
    const size_t fT = (size_t)B * S * 2 * ROW4 * 16, fH = (size_t)B * S * 4 * ROW4 * 16;
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    auto timeit = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 6; ++i) f(); hipEventRecord(c); hipEventSynchronize(c); float ms; hipEventElapsedTime(&ms, a, c); return ms / 6; };
    const double bytes = (double)(2 * fT + 3 * fH);
    for (int flavour = 0; flavour < 2; ++flavour)
        for (int k = 0; k < 6; ++k) {
            float4 *p[5]; const size_t sz[5] = {fT, fT, fH, fH, fH};
            for (int i = 0; i < 5; ++i) {
                if (flavour == 0) CK(hipMalloc(&p[i], sz[i]));
                else CK(hipExtMallocWithFlags((void**)&p[i], sz[i], hipDeviceMallocContiguous));
            }
            void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789));
            const float t = timeit([&] { frame<<<B, 448>>>(p[0], p[1], p[2], p[3], p[4]); });
            printf("%s set %d: %6.3f ms  %5.2f TB/s\n", flavour ? "hipDeviceMallocContiguous" : "hipMalloc                ", k, t, bytes / t * 1e-9); fflush(stdout);
        }
    return 0;
}
