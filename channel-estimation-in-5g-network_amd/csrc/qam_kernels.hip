// Bit <-> symbol mapping and bit-error counting: the QPSK / 16-QAM helpers of src/utils.py:71-158.
//   qam_mod_kernel    bits (one byte per bit, MSB first) -> Gray-mapped constellation point   (utils.py:71-108)
//   qam_demod_kernel  minimum-distance decision, first minimum wins as np.argmin does,       (utils.py:111-153)
//                     inverse Gray map, bits written MSB first
//   bit_errors_kernel number of positions where two bit strings differ                        (utils.py:156-158)
// All three stream their operands once: HBM-bound byte work, decisions in float64 so a complex128 caller gets the
// reference's decision for every symbol (the tables are built exactly as NumPy builds them: level * (1/sqrt(E))).
#include <math.h>

#include "chest_device.h"
#include "chest_internal.h"

namespace chest {

struct QamTable {
    double re[16], im[16];      // constellation in the reference's listing order
    int32_t gray[16];           // decimal value -> constellation index   (gray_map)
    int32_t inv[16];            // constellation index -> decimal value   (np.argsort(gray_map))
    int32_t M, bps;
};

// Host-side tables.  NumPy evaluates `np.array([...]) / np.sqrt(E)` as complex / complex with a zero imaginary
// divisor, which its division loop reduces to a multiplication by 1/sqrt(E); doing the same keeps every bit.
int qam_table(int32_t M, QamTable* t) {
    if (M == 4) {
        const double s = 1.0 / sqrt(2.0);
        const double re[4] = {1, -1, 1, -1}, im[4] = {1, 1, -1, -1};
        const int g[4] = {0, 1, 3, 2};
        for (int i = 0; i < 4; ++i) { t->re[i] = re[i] * s; t->im[i] = im[i] * s; t->gray[i] = g[i]; }
        t->M = 4; t->bps = 2;
    } else if (M == 16) {
        const double s = 1.0 / sqrt(10.0);
        const double re[16] = {-3, -3, -3, -3, -1, -1, -1, -1, 3, 3, 3, 3, 1, 1, 1, 1};
        const double im[16] = {-3, -1, 3, 1, -3, -1, 3, 1, -3, -1, 3, 1, -3, -1, 3, 1};
        const int g[16] = {0, 1, 3, 2, 4, 5, 7, 6, 12, 13, 15, 14, 8, 9, 11, 10};
        for (int i = 0; i < 16; ++i) { t->re[i] = re[i] * s; t->im[i] = im[i] * s; t->gray[i] = g[i]; }
        t->M = 16; t->bps = 4;
    } else {
        return CHEST_EINVAL;
    }
    for (int d = 0; d < t->M; ++d) t->inv[t->gray[d]] = d;
    return CHEST_OK;
}

int qam_constellation_host(int32_t M, double* points, int32_t* gray_map) {
    QamTable t;
    if (qam_table(M, &t) != CHEST_OK) { set_error("modulation order %d not implemented (4 and 16 are)", M); return CHEST_EINVAL; }
    for (int i = 0; i < t.M; ++i) {
        if (points) { points[2 * i] = t.re[i]; points[2 * i + 1] = t.im[i]; }
        if (gray_map) gray_map[i] = t.gray[i];
    }
    return CHEST_OK;
}

template <typename CT, typename RT>
__global__ void __launch_bounds__(256)
qam_mod_kernel(const QamTable tab, int64_t n, const uint8_t* __restrict__ bits, CT* __restrict__ out) {
    __shared__ double sre[16], sim[16];
    if (threadIdx.x < tab.M) {      // entry d = constellation[gray_map[d]]
        sre[threadIdx.x] = tab.re[tab.gray[threadIdx.x]];
        sim[threadIdx.x] = tab.im[tab.gray[threadIdx.x]];
    }
    __syncthreads();
    const int bps = tab.bps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int d = 0;
        if (bps == 2) {
            const uint16_t w = *(const uint16_t*)(bits + i * 2);
            d = ((w & 0xFF) ? 2 : 0) | ((w >> 8) ? 1 : 0);
        } else {
            const uint32_t w = *(const uint32_t*)(bits + i * 4);
            d = ((w & 0xFF) ? 8 : 0) | (((w >> 8) & 0xFF) ? 4 : 0) | (((w >> 16) & 0xFF) ? 2 : 0) | ((w >> 24) ? 1 : 0);
        }
        CT v;
        v.x = (RT)sre[d]; v.y = (RT)sim[d];
        out[i] = v;
    }
}

template <typename CT>
__global__ void __launch_bounds__(256)
qam_demod_kernel(const QamTable tab, int64_t n, const CT* __restrict__ sym, uint8_t* __restrict__ bits) {
    __shared__ double sre[16], sim[16];
    __shared__ int sinv[16];
    if (threadIdx.x < tab.M) {
        sre[threadIdx.x] = tab.re[threadIdx.x];
        sim[threadIdx.x] = tab.im[threadIdx.x];
        sinv[threadIdx.x] = tab.inv[threadIdx.x];
    }
    __syncthreads();
    const int M = tab.M, bps = tab.bps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const CT v = sym[i];
        const double x = (double)v.x, y = (double)v.y;
        double best = INFINITY;
        int arg = 0;
        const bool bad = (x != x) || (y != y);              // np.argmin returns the first NaN: index 0
        for (int c = 0; c < M; ++c) {
            const double dx = x - sre[c], dy = y - sim[c];
            const double d2 = fma(dx, dx, dy * dy);
            if (d2 < best) { best = d2; arg = c; }            // strict: first minimum wins, as np.argmin
        }
        const int d = sinv[bad ? 0 : arg];
        if (bps == 2) {
            *(uint16_t*)(bits + i * 2) = (uint16_t)(((d >> 1) & 1) | ((d & 1) << 8));
        } else {
            *(uint32_t*)(bits + i * 4) = (uint32_t)(((d >> 3) & 1) | (((d >> 2) & 1) << 8) | (((d >> 1) & 1) << 16) | ((d & 1) << 24));
        }
    }
}

// 16 bytes per lane per trip when both strings are 16-B aligned (`vec`), bytewise otherwise; block 0 takes the tail
__global__ void __launch_bounds__(256)
bit_errors_kernel(int64_t n, int vec, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                  unsigned long long* __restrict__ count) {
    unsigned int local = 0;
    if (!vec) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            local += (a[i] != b[i]);
        n = 0;
    }
    const int64_t nvec = n >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 va = ((const uint4*)a)[i], vb = ((const uint4*)b)[i];
        const uint32_t x[4] = {va.x ^ vb.x, va.y ^ vb.y, va.z ^ vb.z, va.w ^ vb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // a byte differs iff any of its bits differs: fold each byte onto its low bit, then count
            uint32_t t = x[j];
            t |= t >> 4; t |= t >> 2; t |= t >> 1;
            local += __popc(t & 0x01010101u);
        }
    }
    const int64_t tail0 = nvec << 4;
    if (blockIdx.x == 0)
        for (int64_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) local += (a[i] != b[i]);
    unsigned int w = local;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(count, (unsigned long long)w);
}

static unsigned grid_for(int64_t items, int per_block) {
    int64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (unsigned)g;
}

int launch_qam_mod(chest_ctx*, hipStream_t st, int32_t M, int64_t n, const uint8_t* bits, int32_t wide, void* out) {
    QamTable t;
    if (qam_table(M, &t) != CHEST_OK) { set_error("modulation order %d not implemented (4 and 16 are)", M); return CHEST_EINVAL; }
    if (n <= 0) return CHEST_OK;
    if (wide) hipLaunchKernelGGL((qam_mod_kernel<double2, double>), dim3(grid_for(n, 1024)), dim3(256), 0, st, t, n, bits, (double2*)out);
    else hipLaunchKernelGGL((qam_mod_kernel<float2, float>), dim3(grid_for(n, 1024)), dim3(256), 0, st, t, n, bits, (float2*)out);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int launch_qam_demod(chest_ctx*, hipStream_t st, int32_t M, int64_t n, int32_t wide, const void* sym, uint8_t* bits) {
    QamTable t;
    if (qam_table(M, &t) != CHEST_OK) { set_error("demodulation order %d not implemented (4 and 16 are)", M); return CHEST_EINVAL; }
    if (n <= 0) return CHEST_OK;
    if (wide) hipLaunchKernelGGL((qam_demod_kernel<double2>), dim3(grid_for(n, 1024)), dim3(256), 0, st, t, n, (const double2*)sym, bits);
    else hipLaunchKernelGGL((qam_demod_kernel<float2>), dim3(grid_for(n, 1024)), dim3(256), 0, st, t, n, (const float2*)sym, bits);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

int launch_bit_errors(chest_ctx*, hipStream_t st, int64_t n, const uint8_t* a, const uint8_t* b, uint64_t* count) {
    CHEST_HIP(hipMemsetAsync(count, 0, sizeof(uint64_t), st));
    if (n <= 0) return CHEST_OK;
    const int vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    hipLaunchKernelGGL(bit_errors_kernel, dim3(grid_for(n, 256 * 16 * 4)), dim3(256), 0, st, n, vec, a, b,
                       (unsigned long long*)count);
    CHEST_HIP(hipGetLastError());
    return CHEST_OK;
}

}  // namespace chest
