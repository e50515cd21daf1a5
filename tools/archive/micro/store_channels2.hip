// Follow-up to store_channels.hip: what moves a tensor's hot write channel?  K tensors (one hipMalloc each, kept); each is
// written whole, then whole with its base shifted by delta bytes inside the same allocation, then in 8 slices of 1 250 frames.
// Under rocprofv3 --pmc TCC_EA0_WRREQ_STALL (json); tools/micro/store_channels_report.py prints the hot channel per dispatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int B = 10000, S = 14, UNIT = 4 * 599 * 8;
constexpr size_t FRAME = (size_t)S * UNIT;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) tensor_store(char* L, int b0) {
    int b = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7; b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3); }
    b += b0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const v4f v = {(float)lane, 2.f, 3.f, (float)b};
    for (int s = 0; s < S; ++s) {
        char* g = L + (size_t)b * FRAME + (size_t)s * UNIT;
        const size_t lo = (size_t)g, hi = lo + UNIT, a0 = lo & ~(size_t)127;
        for (size_t w = a0 + (size_t)wave * 1024; w < hi; w += 4096) {
            const size_t p = w + (size_t)lane * 16;
            if (p >= lo && p + 16 <= hi) __builtin_nontemporal_store(v, (v4f*)p);
        }
    }
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4;
    const size_t bytes = (size_t)B * FRAME, slack = 4u << 20;
    std::vector<char*> L(K);
    for (int k = 0; k < K; ++k) { CK(hipMalloc(&L[k], bytes + slack)); void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789)); }
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    int disp = 0;
    auto run = [&](int k, const char* mode, long val, char* base, int b0, int nb) {
        hipEventRecord(a); tensor_store<<<nb, 256>>>(base, b0); hipEventRecord(c); hipEventSynchronize(c);
        float ms; hipEventElapsedTime(&ms, a, c);
        printf("dispatch %3d  rep 0  set %d  which %ld  %7.3f ms  %5.2f TB/s  %s L@%p\n", disp++, k, val, ms, (double)nb * FRAME * 1e-9 / ms, mode, (void*)base);
    };
    const long shifts[] = {0, 0, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, 2 << 20};
    for (int k = 0; k < K; ++k) {
        for (long d : shifts) run(k, "shift", d, L[k] + d, 0, B);
        for (int i = 0; i < 8; ++i) run(k, "slice", i, L[k], i * 1250, 1250);
    }
    return 0;
}
