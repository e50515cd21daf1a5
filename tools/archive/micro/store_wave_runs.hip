// Does per-WAVE sequentiality explain the write bandwidth?  WG per frame; per symbol the 5 array slices are written
// (a) interleaved across the WG's waves at 1 KiB granularity, (b) each slice split into one contiguous run per wave,
// (c) whole slices handed to single waves (round-robin).  Flat aligned 16-B stores, no compute.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int S = 14, T = 2, R = 2, NSC = 599;
constexpr int nT = T * NSC / 2, nR = R * NSC / 2, nH = R * T * NSC / 2;
__device__ __forceinline__ void piece(float4* p, int n, float4 v, int mode, int& turn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (mode == 0) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v; }
    else if (mode == 1) { const int per = (n + nw - 1) / nw, a = wave * per, e = min(n, a + per); for (int i = a + lane; i < e; i += 64) p[i] = v; }
    else { if (wave == (turn % nw)) for (int i = lane; i < n; i += 64) p[i] = v; ++turn; }
}
__global__ void __launch_bounds__(1024) k(float4* tx, float4* rx, float4* H, float4* L, float4* M, int mode, int split_h) {
    const int b = blockIdx.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
    float4 *txb = tx + (size_t)b * S * nT, *rxb = rx + (size_t)b * S * nR, *Hb = H + (size_t)b * S * nH, *Lb = L + (size_t)b * S * nH, *Mb = M + (size_t)b * S * nH;
    int turn = 0;
    for (int s = 0; s < S; ++s) {
        piece(txb + s * nT, nT, v, mode, turn);
        if (split_h) { piece(Hb + s * nH, nH / 2, v, mode, turn); piece(Hb + s * nH + nH / 2, nH - nH / 2, v, mode, turn); }
        else piece(Hb + s * nH, nH, v, mode, turn);
        piece(rxb + s * nR, nR, v, mode, turn);
    }
    for (int s = 0; s < S; ++s) {
        if (split_h) { piece(Lb + s * nH, nH / 2, v, mode, turn); piece(Lb + s * nH + nH / 2, nH - nH / 2, v, mode, turn);
                       piece(Mb + s * nH, nH / 2, v, mode, turn); piece(Mb + s * nH + nH / 2, nH - nH / 2, v, mode, turn); }
        else { piece(Lb + s * nH, nH, v, mode, turn); piece(Mb + s * nH, nH, v, mode, turn); }
    }
}
// (d) wave per symbol, CHUNK-outer: for each 1-KiB column chunk the wave touches every row of its symbol
//     (rows are 4792 B apart), i.e. 8 short interleaved streams confined to one symbol slice.
// (e) frame_kernel v3: wave per column chunk, all rows of the whole frame (row-strided across 268 KB).
__global__ void __launch_bounds__(1024) kd(float2* tx, float2* rx, float2* H, float2* L, float2* M, int mode) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float2 *txb = tx + (size_t)b * S * T * NSC, *rxb = rx + (size_t)b * S * R * NSC, *Hb = H + (size_t)b * S * R * T * NSC,
           *Lb = L + (size_t)b * S * R * T * NSC, *Mb = M + (size_t)b * S * R * T * NSC;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)b);
    auto put = [&](float2* base, int row, int j) {             // aligned 16-B store of pair chunk j of a row (parity ignored)
        const int k0 = j * 128 + 2 * lane;
        if (k0 + 1 < NSC) *(float4*)(base + ((size_t)row * NSC + k0 & ~(size_t)1)) = v;
    };
    if (mode == 3) {
        for (int s = wave; s < S; s += nw) {
            for (int j = 0; j < 5; ++j) {
                for (int t = 0; t < T; ++t) put(txb, s * T + t, j);
                for (int r = 0; r < R; ++r) { for (int t = 0; t < T; ++t) put(Hb, (s * R + r) * T + t, j); put(rxb, s * R + r, j); }
            }
        }
        for (int s = wave; s < S; s += nw)
            for (int r = 0; r < R; ++r)
                for (int j = 0; j < 5; ++j)
                    for (int t = 0; t < T; ++t) { put(Lb, (s * R + r) * T + t, j); put(Mb, (s * R + r) * T + t, j); }
    } else {
        for (int j = wave; j < 5; j += nw) {
            for (int s = 0; s < S; ++s) {
                for (int t = 0; t < T; ++t) put(txb, s * T + t, j);
                for (int r = 0; r < R; ++r) { for (int t = 0; t < T; ++t) put(Hb, (s * R + r) * T + t, j); put(rxb, s * R + r, j); }
            }
            for (int s = 0; s < S; ++s)
                for (int r = 0; r < R; ++r)
                    for (int t = 0; t < T; ++t) { put(Lb, (s * R + r) * T + t, j); put(Mb, (s * R + r) * T + t, j); }
        }
    }
}
__global__ void __launch_bounds__(256) flat(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <typename F> float timeit(F f) {
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < 6; ++i) f();
    hipEventRecord(c); hipEventSynchronize(c);
    float ms; hipEventElapsedTime(&ms, a, c); return ms / 6;
}
int main() {
    const int B = 10000;
    const size_t fT = (size_t)B * S * nT, fR = (size_t)B * S * nR, fH = (size_t)B * S * nH;
    float4* all; CK(hipMalloc(&all, (fT + fR + 3 * fH) * 16));
    float4 *tx = all, *rx = tx + fT, *H = rx + fR, *L = H + fH, *M = L + fH;
    const double bytes = (double)(fT + fR + 3 * fH) * 16;
    printf("flat fill 2048x256: %.1f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    const char* mn[3] = {"(a) 1-KiB interleave across waves", "(b) one contiguous run per wave per slice", "(c) whole slice per wave, round-robin"};
    const int bts[] = {64, 128, 256, 320, 512, 640};
    printf("%-52s", "mode \\ block"); for (int bt : bts) printf("%7d", bt); printf("   (GB/s)\n");
    for (int sh = 0; sh < 2; ++sh)
        for (int mode = 0; mode < 3; ++mode) {
            char buf[96]; snprintf(buf, sizeof buf, "%s%s", mn[mode], sh ? ", H slices halved" : "");
            printf("%-52s", buf);
            for (int bt : bts) printf("%7.0f", bytes / timeit([&] { k<<<B, bt>>>(tx, rx, H, L, M, mode, sh); }) * 1e-6);
            printf("\n"); fflush(stdout);
        }
    for (int bt : {448, 896}) printf("(d) wave per symbol, chunk-outer, block %d: %.0f GB/s\n", bt, bytes / timeit([&] { kd<<<B, bt>>>((float2*)tx, (float2*)rx, (float2*)H, (float2*)L, (float2*)M, 3); }) * 1e-6);
    for (int bt : {320, 64}) printf("(e) wave per column chunk, all rows of the frame, block %d: %.0f GB/s\n", bt, bytes / timeit([&] { kd<<<B, bt>>>((float2*)tx, (float2*)rx, (float2*)H, (float2*)L, (float2*)M, 4); }) * 1e-6);
    for (int bt : {448, 896}) { char buf[64]; snprintf(buf, 64, "(c) again, block %d", bt); printf("%s: %.0f GB/s\n", buf, bytes / timeit([&] { k<<<B, bt>>>(tx, rx, H, L, M, 2, 1); }) * 1e-6); }
    printf("flat fill 2048x256: %.1f GB/s\n", bytes / timeit([&] { flat<<<2048, 256>>>(all, fT + fR + 3 * fH); }) * 1e-6);
    return 0;
}
