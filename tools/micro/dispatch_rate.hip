// How well does the chip keep three 5-wave workgroups resident per CU?  A grid of N workgroups with the footprint of the 2x2 frame
// kernel (320 threads, 52 KB of LDS -> three per CU by LDS) where every workgroup only WAITS (s_memrealtime, 100 MHz) for a duration
// that varies from workgroup to workgroup (100 ... 180 us, like the frame kernel's lifetimes) and exits.  Variants differ in the
// VECTOR REGISTERS the kernel claims (an asm that touches the highest register): 96 = five waves per SIMD (20 wave slots per CU,
// 15 used), 80 = six, 64 = eight, 24.  ideal = sum of the waits / (256 CUs x 3).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/dispatch_rate tools/micro/dispatch_rate.hip && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VG>
__global__ void __launch_bounds__(320) waiter(int base_ticks, int step_ticks, unsigned long long* out) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) lds[0] = 1.f;
    if constexpr (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (VG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if constexpr (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    const int ticks = base_ticks + step_ticks * (int)((blockIdx.x * 2654435761u >> 16) % 9);
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0 && out) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int VG>
static void run(int threads, int lds_kb, unsigned long long* d, int N) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)waiter<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int step : {0, 10}) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(waiter<VG>, dim3(N), dim3(threads), lds_kb * 1024, 0, 100 * 100, step * 100, d);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        std::vector<unsigned long long> h(2 * N);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0; double life = 0;
        for (int i = 0; i < N; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); life += (double)(h[2 * i + 1] - h[2 * i]); }
        const double span_us = (hi - lo) / 100.0, ideal_ms = life / 100.0 / (256.0 * 3) / 1000.0;
        printf("%3d VGPRs, %d threads, %d KB LDS, waits %s: kernel %.3f ms, ideal %.3f (x%.2f), mean resident workgroups per CU %.2f\n",
               VG, threads, lds_kb, step ? "100-180 us" : "100 us    ", best, ideal_ms, best / ideal_ms, life / 100.0 / span_us / 256.0);
    }
}

int main() {
    const int N = 10000;
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * N);
    run<24>(320, 52, d, N);
    run<64>(320, 52, d, N);
    run<80>(320, 52, d, N);
    run<96>(320, 52, d, N);
    run<96>(256, 52, d, N);        // four waves: one per SIMD
    run<96>(384, 52, d, N);        // six waves
    return 0;
}
