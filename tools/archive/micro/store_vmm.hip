// Can placement be made reliably good?  Each of the five tensors is a contiguous VIRTUAL range backed by many small
// physical allocations (HIP virtual-memory API), chunk size swept; compared with plain hipMalloc sets in the same process.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
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
static size_t gran = 0;
static int vmm_alloc(void** out, size_t bytes, size_t chunk, size_t va_align = 0) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    if (!gran) CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    void* va; CK(hipMemAddressReserve(&va, n * chunk, va_align, nullptr, 0));
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)va + i * chunk, chunk, 0, h, 0));
        CK(hipMemRelease(h));
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    *out = va;
    return 0;
}
int main() {
    const size_t fT = (size_t)B * S * 2 * ROW4 * 16, fH = (size_t)B * S * 4 * ROW4 * 16;
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    auto timeit = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 6; ++i) f(); hipEventRecord(c); hipEventSynchronize(c); float ms; hipEventElapsedTime(&ms, a, c); return ms / 6; };
    const double bytes = (double)(2 * fT + 3 * fH);
    for (int k = 0; k < 4; ++k) {
        float4 *tx, *rx, *H, *L, *M;
        CK(hipMalloc(&tx, fT)); CK(hipMalloc(&rx, fT)); CK(hipMalloc(&H, fH)); CK(hipMalloc(&L, fH)); CK(hipMalloc(&M, fH));
        const float t = timeit([&] { frame<<<B, 448>>>(tx, rx, H, L, M); });
        printf("hipMalloc set %d:                 %6.3f ms  %5.2f TB/s\n", k, t, bytes / t * 1e-9); fflush(stdout);
    }
    for (size_t chunk : {(size_t)2 << 20, (size_t)1 << 30})
        for (int k = 0; k < 3; ++k) {
            void *tx, *rx, *H, *L, *M;
            if (vmm_alloc(&tx, fT, chunk) || vmm_alloc(&rx, fT, chunk) || vmm_alloc(&H, fH, chunk) || vmm_alloc(&L, fH, chunk) || vmm_alloc(&M, fH, chunk)) return 1;
            const float t = timeit([&] { frame<<<B, 448>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)L, (float4*)M); });
            printf("VMM chunks of %4zu MiB, set %d (granularity %zu KiB): %6.3f ms  %5.2f TB/s\n", chunk >> 20, k, gran >> 10, t, bytes / t * 1e-9); fflush(stdout);
        }
    for (size_t chunk : {(size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)2 << 30})
        for (int k = 0; k < 4; ++k) {
            void *tx, *rx, *H, *L, *M;
            if (vmm_alloc(&tx, fT, chunk, chunk) || vmm_alloc(&rx, fT, chunk, chunk) || vmm_alloc(&H, fH, chunk, chunk) || vmm_alloc(&L, fH, chunk, chunk) || vmm_alloc(&M, fH, chunk, chunk)) return 1;
            const float t = timeit([&] { frame<<<B, 448>>>((float4*)tx, (float4*)rx, (float4*)H, (float4*)L, (float4*)M); });
            printf("VMM chunks of %4zu MiB, VA aligned to the chunk, set %d: %6.3f ms  %5.2f TB/s   tx@%p\n", chunk >> 20, k, t, bytes / t * 1e-9, tx); fflush(stdout);
        }
    // interleaved creation: chunk i of every tensor is created before chunk i+1 of any (physical neighbours belong to different tensors)
    return 0;
}
