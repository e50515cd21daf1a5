// Is the HBM write rate of a fill a matter of WHICH XCD writes WHICH 4-KiB page?  Workgroups are dealt round-robin to the 8 XCDs
// (workgroup i runs on XCD i % 8).  A fill in 4-KiB tiles lets workgroup i write page f(i):
//   shift k : f(i) = i + k             XCD x always writes the pages = x + k (mod 8)              (k = 0: the plain tile fill, 7.0 TB/s)
//   rotate  : f(i) = 8*(i/8) + (i%8 + i/8) % 8     same narrow window, but every XCD visits every page class
//   unit U  : a tile of U KiB per workgroup (f(i) = i in units of U KiB): which address bits pair an XCD with its pages
//   hipcc -O3 --offload-arch=gfx950 -o build/store_affinity tools/micro/store_affinity.hip && ./build/store_affinity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: shift by k tiles; mode 1: rotate; tile = 256 * PER float4 (PER = 1: 4 KiB)
template <int PER>
__global__ void __launch_bounds__(256) fill(v4f* p, size_t ntiles, int mode, int k) {
    size_t i = blockIdx.x;
    size_t t = mode == 0 ? (i + (size_t)k) % ntiles : ((i >> 3) << 3) + (((i & 7) + (i >> 3)) & 7);
    if (t >= ntiles) t = i;
    v4f* q = p + t * (256 * PER);
#pragma unroll
    for (int j = 0; j < PER; ++j) q[j * 256 + threadIdx.x] = v4f{(float)i, 2.f, 3.f, (float)j};
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
    const size_t bytes = (size_t)10 << 30;                       // 10 GiB
    v4f* all; CK(hipMalloc(&all, bytes));
    const size_t n4k = bytes / 4096;
    auto rep = [&](const char* nm, float ms) { printf("%-64s %7.3f ms  %7.1f GB/s\n", nm, ms, (double)bytes / ms * 1e-6); fflush(stdout); };
    char buf[128];
    for (int round = 0; round < 2; ++round) {
        for (int k = 0; k < 8; ++k) { snprintf(buf, sizeof buf, "4-KiB tiles, shift %d", k); rep(buf, timeit([&] { fill<1><<<(unsigned)n4k, 256>>>(all, n4k, 0, k); })); }
        rep("4-KiB tiles, rotating pairing", timeit([&] { fill<1><<<(unsigned)n4k, 256>>>(all, n4k, 1, 0); }));
        rep("8-KiB tiles", timeit([&] { fill<2><<<(unsigned)(n4k / 2), 256>>>(all, n4k / 2, 0, 0); }));
        rep("16-KiB tiles", timeit([&] { fill<4><<<(unsigned)(n4k / 4), 256>>>(all, n4k / 4, 0, 0); }));
        rep("32-KiB tiles", timeit([&] { fill<8><<<(unsigned)(n4k / 8), 256>>>(all, n4k / 8, 0, 0); }));
        rep("32-KiB tiles, shift 1", timeit([&] { fill<8><<<(unsigned)(n4k / 8), 256>>>(all, n4k / 8, 0, 1); }));
        
    }
    return 0;
}
