// Partial-line hypothesis: the real frame kernel stores 63 lanes x 16 B = 1008 B per instruction into rows of 4792 B, so nearly
// every store instruction begins and ends inside a 128-B line that the neighbouring chunk completes microseconds later; the
// model in store_flavour.hip used 1024-B stores into 4800-B rows.  Same schedule, both geometries, same buffers, with and
// without work between a wave's chunks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, B = 10000;
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(char* p, float4 v) { const v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4f*)p); }

// rowB: bytes per row (4800 or 4792); lanes: storing lanes per chunk (64 or 63); rows that start at 8 mod 16 are shifted by 8 B
// so that every store stays 16-B aligned (as the real kernel does)
__global__ void __launch_bounds__(448) frame(char* tx, char* rx, char* H, char* L, char* M, int rowB, int lanes, int delay) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = (float)lane;
    const int chunkB = lanes * 16;
    for (int phase = 0; phase < 2; ++phase)
        for (int pass = 0; pass < 2; ++pass) {
            const int s = wave + 7 * pass;
            for (int j = 0; j < 5; ++j) {
                for (int d = 0; d < delay; ++d) acc = __builtin_amdgcn_sinf(acc) + 0.5f;
                const float4 v = make_float4(acc, 2.f, 3.f, (float)b);
                auto put = [&](char* base, size_t row) {
                    const size_t r0 = row * (size_t)rowB;
                    const size_t off = ((r0 + 15) & ~(size_t)15) + (size_t)j * chunkB + (size_t)lane * 16;    // 16-B aligned inside the row
                    if (lane < lanes && off + 16 <= r0 + rowB) st(base + off, v);
                };
                if (phase == 0) {
                    for (int r = 0; r < 2; ++r) put(tx, (size_t)(b * S + s) * 2 + r);
                    for (int r = 0; r < 2; ++r) put(rx, (size_t)(b * S + s) * 2 + r);
                    for (int q = 0; q < 4; ++q) put(H, (size_t)(b * S + s) * 4 + q);
                } else {
                    for (int q = 0; q < 4; ++q) { put(L, (size_t)(b * S + s) * 4 + q); put(M, (size_t)(b * S + s) * 4 + q); }
                }
            }
            if (pass == 1 && phase == 0) __syncthreads();
        }
}
int main() {
    const size_t fT = (size_t)B * S * 2 * 4864, fH = (size_t)B * S * 4 * 4864;
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    auto timeit = [&](auto f) { f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 6; ++i) f(); hipEventRecord(c); hipEventSynchronize(c); float ms; hipEventElapsedTime(&ms, a, c); return ms / 6; };
    for (int k = 0; k < 3; ++k) {
        char *tx, *rx, *H, *L, *M;
        CK(hipMalloc(&tx, fT)); CK(hipMalloc(&rx, fT)); CK(hipMalloc(&H, fH)); CK(hipMalloc(&L, fH)); CK(hipMalloc(&M, fH));
        void* pad; CK(hipMalloc(&pad, (size_t)(k + 1) * 123456789));
        for (int delay : {0, 120}) {
            const float t_full = timeit([&] { frame<<<B, 448>>>(tx, rx, H, L, M, 4800, 64, delay); });
            const float t_part = timeit([&] { frame<<<B, 448>>>(tx, rx, H, L, M, 4792, 63, delay); });
            const float t_mix = timeit([&] { frame<<<B, 448>>>(tx, rx, H, L, M, 4800, 63, delay); });
            const float t_pad = timeit([&] { frame<<<B, 448>>>(tx, rx, H, L, M, 4864, 64, delay); });
            printf("set %d delay %3d: 4800-B rows / 1024-B stores %6.3f ms | 4792-B rows / 1008-B stores %6.3f ms | 4800-B rows / 1008-B stores %6.3f ms | 4864-B rows (38 lines) / 1024-B stores %6.3f ms\n", k, delay, t_full, t_part, t_mix, t_pad);
            fflush(stdout);
        }
    }
    return 0;
}
