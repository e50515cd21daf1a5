// Which level of the two-level write rate an output set gets, with the per-channel L2 -> fabric write counters beside it.
// K sets of two equal tensors (the H_ls / H_mmse pair: 10 000 frames x 14 x 4 x 599 complex64 each) are allocated by separate
// hipMallocs and all kept; every set is written by the same kernel: workgroup b streams frame b of BOTH tensors as
// line-aligned 1 KiB wave stores, unit by unit (19 168 B of one tensor, then of the other), the order the frame kernel's
// store-out produces.  Run under  rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL ... --output-format json  and feed the
// result to tools/micro/store_channels_report.py; stdout lists the dispatch order (set, time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int B = 10000, S = 14, UNIT = 4 * 599 * 8;             // bytes per (symbol) unit of one tensor
constexpr size_t FRAME = (size_t)S * UNIT;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) pair_store(char* L, char* M, int which) {
    int b = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7; b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const v4f v = {(float)lane, 2.f, 3.f, (float)b};
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < 2; ++t) {
            if (!((which >> t) & 1)) continue;
            char* g = (t ? M : L) + (size_t)b * FRAME + (size_t)s * UNIT;
            const size_t lo = (size_t)g, hi = lo + UNIT, a0 = lo & ~(size_t)127;
            for (size_t w = a0 + (size_t)wave * 1024; w < hi; w += 4096) {
                const size_t p = w + (size_t)lane * 16;
                if (p >= lo && p + 16 <= hi) __builtin_nontemporal_store(v, (v4f*)p);
                else if (p + 8 == lo || p + 8 == hi - 8 + 8) { /* 8-B edge halves: skipped in the model (0.1 % of the bytes) */ }
            }
        }
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 6, reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t bytes = (size_t)B * FRAME;
    std::vector<char*> L(K), M(K);
    for (int k = 0; k < K; ++k) {
        CK(hipMalloc(&L[k], bytes + 4096)); CK(hipMalloc(&M[k], bytes + 4096));
        void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789));
    }
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    int disp = 0;
    for (int rep = 0; rep < reps; ++rep)
        for (int k = 0; k < K; ++k)
            for (int which : {3, 1, 2}) {
                hipEventRecord(a); pair_store<<<B, 256>>>(L[k], M[k], which); hipEventRecord(c); hipEventSynchronize(c);
                float ms; hipEventElapsedTime(&ms, a, c);
                const double gb = (double)bytes * (which == 3 ? 2 : 1) * 1e-9;
                printf("dispatch %3d  rep %d  set %d  which %d  %7.3f ms  %5.2f TB/s  L@%p M@%p\n", disp++, rep, k, which, ms, gb / ms, (void*)L[k], (void*)M[k]);
            }
    return 0;
}
