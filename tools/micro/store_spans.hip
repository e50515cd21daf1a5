// What costs the interpolation kernel (config 5: 129 024 spans of 38 368 B, one workgroup each, 5.35 TB/s) against a plain fill of the
// same tensor (6.9 TB/s)?  Pure-store models of one workgroup per span, same process, same buffer:
//   span length 38 368 B (4 x 1 199 x 8: spans start at 96 mod 128) vs 38 400 B (line-aligned); nt vs plain stores; values from LDS vs registers;
//   the span written by 4 waves interleaving 1-KiB pieces (the product) vs each wave a contiguous quarter.
//   hipcc -O3 --offload-arch=gfx950 -o build/store_spans tools/micro/store_spans.hip && ./build/store_spans
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// n8 = span length in 8-B elements; the span starts at element span * n8 of p (p is 256-B aligned)
template <bool NT, bool LDS, int ORDER>
__global__ void __launch_bounds__(256) spans(float2* p, int n8, int nrow) {
    __shared__ float2 row[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (LDS) { for (int k = tid; k < nrow; k += 256) row[k] = make_float2((float)(k + blockIdx.x), 1.0f); __syncthreads(); }
    float2* g0 = p + (size_t)blockIdx.x * n8;
    const int head = (int)(((uintptr_t)g0 & 127u) >> 3);
    const int nst = (head + n8 + 127) >> 7;
    const int per = (nst + 3) >> 2;
    for (int j = 0; j < per; ++j) {
        const int q = ORDER == 0 ? wave + 4 * j : wave * per + j;
        if (q >= nst) break;
        const int e = (q << 7) - head + 2 * lane;
        if (e >= -1 && e < n8) {
            const int ec = e < 0 ? 0 : e;
            float2 a, c;
            if (LDS) { const int i0 = ec % nrow, i1 = i0 + 1 == nrow ? 0 : i0 + 1; a = row[i0]; c = row[i1]; }
            else { a = make_float2((float)ec, 2.0f); c = make_float2((float)(ec + 1), 3.0f); }
            if (e >= 0 && e + 1 < n8) {
                const v4f v = {a.x, a.y, c.x, c.y};
                if (NT) __builtin_nontemporal_store(v, (v4f*)(g0 + e)); else *(v4f*)(g0 + e) = v;
            } else {
                const v2f v = {a.x, a.y};
                if (NT) __builtin_nontemporal_store(v, (v2f*)(g0 + ec)); else *(v2f*)(g0 + ec) = v;
            }
        }
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; (void)hipEventCreate(&a); (void)hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    (void)hipDeviceSynchronize(); (void)hipEventRecord(a);
    for (int i = 0; i < 8; ++i) f();
    (void)hipEventRecord(c); (void)hipEventSynchronize(c);
    float ms; (void)hipEventElapsedTime(&ms, a, c); return ms / 8;
}
int main() {
    const int NS = 2304 * 14 * 4;
    float2* all; CK(hipMalloc(&all, (size_t)NS * 4800 * 8 + 4096));
    for (int round = 0; round < 2; ++round)
        for (int n8 : {4 * 1199, 4 * 1200}) {
            const double bytes = (double)NS * n8 * 8;
            auto rep = [&](const char* nm, float ms) { printf("span %5d B  %-58s %7.3f ms  %7.1f GB/s\n", n8 * 8, nm, ms, bytes / ms * 1e-6); fflush(stdout); };
            rep("nt, from LDS, waves interleave 1-KiB pieces (the product)", timeit([&] { spans<true, true, 0><<<NS, 256>>>(all, n8, n8 / 4); }));
            rep("nt, from registers, waves interleave", timeit([&] { spans<true, false, 0><<<NS, 256>>>(all, n8, n8 / 4); }));
            rep("plain stores, from LDS, waves interleave", timeit([&] { spans<false, true, 0><<<NS, 256>>>(all, n8, n8 / 4); }));
            rep("plain stores, from registers, waves interleave", timeit([&] { spans<false, false, 0><<<NS, 256>>>(all, n8, n8 / 4); }));
            rep("nt, from LDS, each wave a contiguous quarter", timeit([&] { spans<true, true, 1><<<NS, 256>>>(all, n8, n8 / 4); }));
            rep("plain, from registers, each wave a contiguous quarter", timeit([&] { spans<false, false, 1><<<NS, 256>>>(all, n8, n8 / 4); }));
        }
    return 0;
}
