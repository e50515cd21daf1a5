// Is the write rate of the frame kernel's store schedule a property of WHERE the five tensors live?  K complete sets of
// (tx, rx, H, H_ls, H_mmse) are allocated (separate hipMallocs, all kept), each timed with the same kernel, round-robin.
// Also: the same five tensors written one after the other (no concurrency between tensors), and a flat fill per set.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, ROW4 = 300, B = 10000;
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float4* p, float4 v) { const v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4f*)p); }

__global__ void __launch_bounds__(448) frame(float4* tx, float4* rx, float4* H, float4* L, float4* M, int which) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 v = make_float4((float)lane, 2.f, 3.f, (float)b);
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4) {
                if (which & 1) for (int r = 0; r < 2; ++r) st(tx + ((size_t)(b * S + s) * 2 + r) * ROW4 + i, v);
                if (which & 2) for (int r = 0; r < 2; ++r) st(rx + ((size_t)(b * S + s) * 2 + r) * ROW4 + i, v);
                if (which & 4) for (int q = 0; q < 4; ++q) st(H + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v);
            }
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const int s = wave + 7 * pass;
        for (int j = 0; j < 5; ++j) {
            const int i = j * 64 + lane;
            if (i < ROW4)
                for (int q = 0; q < 4; ++q) {
                    if (which & 8) st(L + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v);
                    if (which & 16) st(M + ((size_t)(b * S + s) * 4 + q) * ROW4 + i, v);
                }
        }
    }
}
__global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st(p + i, make_float4(1.f, 2.f, 3.f, 4.f));
}
struct Set { float4 *tx, *rx, *H, *L, *M; };
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const size_t fT = (size_t)B * S * 2 * ROW4, fH = (size_t)B * S * 4 * ROW4;
    std::vector<Set> sets(K);
    for (int k = 0; k < K; ++k) {
        CK(hipMalloc(&sets[k].tx, fT * 16)); CK(hipMalloc(&sets[k].rx, fT * 16)); CK(hipMalloc(&sets[k].H, fH * 16));
        CK(hipMalloc(&sets[k].L, fH * 16)); CK(hipMalloc(&sets[k].M, fH * 16));
        void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789));
    }
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    auto timeit = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 6; ++i) f(); hipEventRecord(c); hipEventSynchronize(c); float ms; hipEventElapsedTime(&ms, a, c); return ms / 6; };
    const double bytes = (double)(2 * fT + 3 * fH) * 16;
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < K; ++k) {
            Set s = sets[k];
            const float t_all = timeit([&] { frame<<<B, 448>>>(s.tx, s.rx, s.H, s.L, s.M, 31); });
            const float t_seq = timeit([&] { for (int w : {1, 2, 4, 8, 16}) frame<<<B, 448>>>(s.tx, s.rx, s.H, s.L, s.M, w); });
            const float t_w = timeit([&] { frame<<<B, 448>>>(s.tx, s.rx, s.H, s.L, s.M, 7); });
            const float t_e = timeit([&] { frame<<<B, 448>>>(s.tx, s.rx, s.H, s.L, s.M, 24); });
            const float t_flat = timeit([&] { flat<<<8192, 256>>>(s.tx, fT); flat<<<8192, 256>>>(s.rx, fT); flat<<<8192, 256>>>(s.H, fH); flat<<<8192, 256>>>(s.L, fH); flat<<<8192, 256>>>(s.M, fH); });
            printf("set %d: all five %6.3f ms (%5.2f TB/s) | one tensor per launch %6.3f | tx+rx+H %6.3f | H_ls+H_mmse %6.3f | flat fills %6.3f ms   tx@%p\n", k, t_all,
                   bytes / t_all * 1e-9, t_seq, t_w, t_e, t_flat, (void*)s.tx);
            fflush(stdout);
        }
    return 0;
}
