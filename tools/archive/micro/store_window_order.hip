// Does the window -> wave assignment of the store-out decide the write rate?  The frame kernel's store-out gives wave w the
// 1 KiB windows w, w+4, w+8, ... of a segment: successive stores of a wave are 4 KiB apart.  Modes (K sets of the H_ls / H_mmse
// pair, every mode on every set, no profiler):
//   0  stride 4 (the kernel's)           1  contiguous run per wave       2  skewed: window 4j + ((w + j) & 3)
//   3  5 waves, stride 5                 4  3 waves, stride 3             5  8 waves, stride 8
//   6  stride 4, units visited in a per-workgroup rotated symbol order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int B = 10000, S = 14, UNIT = 4 * 599 * 8;
constexpr size_t FRAME = (size_t)S * UNIT;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) pair_store(char* L, char* M, int mode) {
    int b = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7; b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const v4f v = {(float)lane, 2.f, 3.f, (float)b};
    for (int si = 0; si < S; ++si) {
        const int s = mode == 6 ? (si + b * 5) % S : si;
        for (int t = 0; t < 2; ++t) {
            char* g = (t ? M : L) + (size_t)b * FRAME + (size_t)s * UNIT;
            const size_t lo = (size_t)g, hi = lo + UNIT, a0 = lo & ~(size_t)127;
            const int nst = (int)((hi - a0 + 1023) >> 10);
            auto put = [&](int q) {
                const size_t p = a0 + (size_t)q * 1024 + (size_t)lane * 16;
                if (p >= lo && p + 16 <= hi) __builtin_nontemporal_store(v, (v4f*)p);
            };
            if (mode == 1) { const int m = (nst + nw - 1) / nw; for (int q = wave * m; q < min(nst, (wave + 1) * m); ++q) put(q); }
            else if (mode == 2) { for (int j = 0; 4 * j < nst; ++j) { const int q = 4 * j + ((wave + j) & 3); if (q < nst) put(q); } }
            else for (int q = wave; q < nst; q += nw) put(q);
        }
    }
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 6, reps = argc > 2 ? atoi(argv[2]) : 2;
    const size_t bytes = (size_t)B * FRAME;
    std::vector<char*> L(K), M(K);
    for (int k = 0; k < K; ++k) {
        CK(hipMalloc(&L[k], bytes + 4096)); CK(hipMalloc(&M[k], bytes + 4096));
        void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789));
    }
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    const int modes[] = {0, 1, 2, 3, 4, 5, 6}, waves[] = {4, 4, 4, 5, 3, 8, 4};
    for (int rep = 0; rep < reps; ++rep)
        for (int k = 0; k < K; ++k) {
            printf("rep %d set %d:", rep, k);
            for (int m = 0; m < 7; ++m) {
                pair_store<<<B, waves[m] * 64>>>(L[k], M[k], modes[m]);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int i = 0; i < 4; ++i) pair_store<<<B, waves[m] * 64>>>(L[k], M[k], modes[m]);
                hipEventRecord(c); hipEventSynchronize(c);
                float ms; hipEventElapsedTime(&ms, a, c); ms /= 4;
                printf("  m%d %6.3f ms %5.2f", modes[m], ms, 2.0 * bytes * 1e-9 / ms);
            }
            printf("   (TB/s)\n"); fflush(stdout);
        }
    return 0;
}
