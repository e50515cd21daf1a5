// Does the HBM write rate depend on the DATA written (constant vs pseudo-random bits) and on nt vs plain stores, for a purely
// sequential fill of 10.9 GB -- the reference point the frame kernel's 5.2-5.9 TB/s is compared with (tools/fill_vs_kernel.py)?
//   hipcc -O3 --offload-arch=gfx950 -o build/store_data tools/micro/store_data.hip && ./build/store_data
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f value(size_t i, int mode, uint32_t salt) {
    if (mode == 0) return v4f{1.f, 2.f, 3.f, 4.f};
    uint32_t h = (uint32_t)i * 2654435761u + salt;                          // cheap hash: ~6 vector instructions per 16 B
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint32_t a = h, b = h * 3266489917u, c = h ^ 0x9E3779B9u, d = b ^ (h >> 7);
    // keep the bit patterns inside normal floats of magnitude ~1 (what the kernel writes): random mantissa + sign, fixed exponent
    auto f = [](uint32_t u) { return __uint_as_float((u & 0x807FFFFFu) | 0x3F000000u); };
    return v4f{f(a), f(b), f(c), f(d)};
}

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) flat(v4f* p, size_t n, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = value(i, MODE, salt);
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
// one workgroup per "frame": 5 arrays (tx, rx: 134 176 B; H, H_ls, H_mmse: 268 352 B per frame), each frame's span written whole
template <int MODE, bool NT, bool XCD = true, int THR = 320>
__global__ void __launch_bounds__(THR) per_frame(v4f* base, size_t fT, size_t fH, int B, uint32_t salt) {
    extern __shared__ float lds_dummy[];
    if (salt == 0xFFFFFFFFu) lds_dummy[threadIdx.x] = 0.f;             // (keeps the dynamic LDS allocation)
    int b = blockIdx.x;
    if (XCD) { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7; b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3); }
    const size_t nT = fT / B, nH = fH / B;
    v4f* arr[5] = {base, base + fT, base + 2 * fT, base + 2 * fT + fH, base + 2 * fT + 2 * fH};
    const size_t len[5] = {nT, nT, nH, nH, nH};
    for (int a = 0; a < 5; ++a) {
        v4f* p = arr[a] + (size_t)b * len[a];
        for (size_t i = threadIdx.x; i < len[a]; i += blockDim.x) {
            const v4f v = value((size_t)b * 131 + i + a, MODE, salt);
            if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
        }
    }
}
// a tile per workgroup, no loop over the grid (the shape of torch's vectorised fill): workgroup w writes float4 [w * TILE, (w + 1) * TILE)
template <int MODE, bool NT, int PER>
__global__ void __launch_bounds__(256) tile(v4f* p, size_t n, uint32_t salt) {
    const size_t base = (size_t)blockIdx.x * (256 * PER);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        if (i < n) { const v4f v = value(i, MODE, salt); if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
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
    const int B = 10000;
    const size_t fT = (size_t)B * 134176 / 16, fH = (size_t)B * 268352 / 16, n = 2 * fT + 3 * fH;
    v4f* all; CK(hipMalloc(&all, n * 16));
    const double bytes = (double)n * 16;
    auto rep = [&](const char* nm, float ms) { printf("%-72s %7.3f ms  %7.1f GB/s\n", nm, ms, bytes / ms * 1e-6); fflush(stdout); };
    for (int round = 0; round < 2; ++round) {
        rep("flat fill, constant data, plain stores", timeit([&] { flat<0, false><<<4096, 256>>>(all, n, 1); }));
        rep("flat fill, constant data, nt stores", timeit([&] { flat<0, true><<<4096, 256>>>(all, n, 1); }));
        rep("flat fill, pseudo-random data, plain stores", timeit([&] { flat<1, false><<<4096, 256>>>(all, n, 7); }));
        rep("flat fill, pseudo-random data, nt stores", timeit([&] { flat<1, true><<<4096, 256>>>(all, n, 7); }));
        rep("hipMemsetD32Async", timeit([&] { (void)hipMemsetD32Async((hipDeviceptr_t)all, 0x3F800000, n * 4, 0); }));
        rep("tile per workgroup (4 KiB), constant data, plain stores", timeit([&] { tile<0, false, 1><<<(unsigned)((n + 255) / 256), 256>>>(all, n, 1); }));
        rep("tile per workgroup (16 KiB), constant data, plain stores", timeit([&] { tile<0, false, 4><<<(unsigned)((n + 1023) / 1024), 256>>>(all, n, 1); }));
        rep("tile per workgroup (16 KiB), pseudo-random data, plain stores", timeit([&] { tile<1, false, 4><<<(unsigned)((n + 1023) / 1024), 256>>>(all, n, 1); }));
        rep("tile per workgroup (16 KiB), pseudo-random data, nt stores", timeit([&] { tile<1, true, 4><<<(unsigned)((n + 1023) / 1024), 256>>>(all, n, 1); }));
        rep("tile per workgroup (64 KiB), pseudo-random data, plain stores", timeit([&] { tile<1, false, 16><<<(unsigned)((n + 4095) / 4096), 256>>>(all, n, 1); }));
        rep("workgroup per frame (5 spans), constant data, nt stores", timeit([&] { per_frame<0, true><<<B, 320, 0>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame (5 spans), pseudo-random data, nt stores", timeit([&] { per_frame<1, true><<<B, 320>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame, nt, frames dealt round-robin to the XCDs (b = blockIdx)", timeit([&] { per_frame<1, true, false><<<B, 320>>>(all, fT, fH, B, 3); }));
        (void)hipFuncSetAttribute((const void*)per_frame<1, true, true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)per_frame<1, true, false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)per_frame<1, true, true, 320>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        rep("workgroup per frame, nt, 1024 threads, 2 per CU (70 KB LDS), XCD-contiguous", timeit([&] { per_frame<1, true, true, 1024><<<B, 1024, 70 * 1024>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame, nt, 1024 threads, 1 per CU (150 KB LDS), XCD-contiguous", timeit([&] { per_frame<1, true, true, 1024><<<B, 1024, 150 * 1024>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame, nt, 1024 threads, 1 per CU (150 KB LDS), b = blockIdx", timeit([&] { per_frame<1, true, false, 1024><<<B, 1024, 150 * 1024>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame, nt, 320 threads, 1 per CU (150 KB LDS), XCD-contiguous", timeit([&] { per_frame<1, true, true, 320><<<B, 320, 150 * 1024>>>(all, fT, fH, B, 3); }));
        rep("workgroup per frame (5 spans), pseudo-random data, plain stores", timeit([&] { per_frame<1, false><<<B, 320>>>(all, fT, fH, B, 3); }));
    }
    return 0;
}
