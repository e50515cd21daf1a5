// Accuracy of cheap gfx950 transcendentals vs float64, for inputs the frame kernel uses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__device__ inline void sincos_turns_poly(float f, float* s, float* c) {
    // f in [0,1): exp(2 pi j f). quadrant reduction + Taylor on [-pi/4, pi/4]
    const float q = rintf(4.0f * f);
    const float r = fmaf(-0.25f, q, f);                   // exact: [-1/8, 1/8]
    const float a = 6.28318530717958647692f * r;
    const float a2 = a * a;
    float sp = fmaf(a2, -2.5052108e-08f, 2.7557319e-06f);
    sp = fmaf(a2, sp, -1.9841270e-04f);
    sp = fmaf(a2, sp, 8.3333333e-03f);
    sp = fmaf(a2, sp, -1.6666667e-01f);
    const float sn = fmaf(a * a2, sp, a);
    float cp = fmaf(a2, 2.4801587e-05f, -1.3888889e-03f);
    cp = fmaf(a2, cp, 4.1666667e-02f);
    cp = fmaf(a2, cp, -0.5f);
    const float cs = fmaf(a2, cp, 1.0f);
    const int qi = (int)q & 3;
    const float s0 = (qi & 1) ? cs : sn, c0 = (qi & 1) ? sn : cs;
    *s = (qi == 2 || qi == 3) ? -s0 : s0;       // q=1: sin=cos(a), cos=-sin(a); q=2: -sin,-cos; q=3: -cos, sin
    *c = (qi == 1 || qi == 2) ? -c0 : c0;
}

__global__ void k(int n, const float* x, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f = x[i];
    float s, c;
    sincospif(2.0f * f, &s, &c);
    out[0 * n + i] = s; out[1 * n + i] = c;
    out[2 * n + i] = __builtin_amdgcn_sinf(f); out[3 * n + i] = __builtin_amdgcn_cosf(f);
    sincos_turns_poly(f, &s, &c);
    out[4 * n + i] = s; out[5 * n + i] = c;
    float u = fmaxf(f, 5.9604644775390625e-8f);
    out[6 * n + i] = logf(u);
    out[7 * n + i] = __builtin_amdgcn_logf(u) * 0.693147180559945309f;
    out[8 * n + i] = __logf(u);
    float m = -2.0f * logf(u);
    out[9 * n + i] = sqrtf(m);
    out[10 * n + i] = __builtin_amdgcn_sqrtf(m);
    out[11 * n + i] = 1.0f / (1.0f + f);
    out[12 * n + i] = __builtin_amdgcn_rcpf(1.0f + f);
}

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = (float)i / (float)n;           // every 24-bit uniform
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 13ull * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(n, dx, dout);
    std::vector<float> o(13ull * n);
    hipMemcpy(o.data(), dout, 13ull * n * 4, hipMemcpyDeviceToHost);
    const char* names[13] = {"sincospif.sin", "sincospif.cos", "v_sin", "v_cos", "poly.sin", "poly.cos", "logf", "v_log*ln2",
                             "__logf", "sqrtf", "v_sqrt", "1/x", "v_rcp"};
    double maxabs[13] = {0}, maxrel[13] = {0};
    for (int i = 0; i < n; ++i) {
        double f = x[i], u = fmax(f, 5.9604644775390625e-8);
        double ref[13] = {sin(2 * M_PI * f), cos(2 * M_PI * f), sin(2 * M_PI * f), cos(2 * M_PI * f), sin(2 * M_PI * f),
                          cos(2 * M_PI * f), log(u), log(u), log(u), 0, 0, 1.0 / (1.0 + f), 1.0 / (1.0 + f)};
        double m = -2.0 * (double)o[6ull * n + i];
        ref[9] = sqrt(m); ref[10] = sqrt(m);
        for (int j = 0; j < 13; ++j) {
            double e = fabs((double)o[(size_t)j * n + i] - ref[j]);
            if (e > maxabs[j]) maxabs[j] = e;
            if (fabs(ref[j]) > 1e-30 && e / fabs(ref[j]) > maxrel[j]) maxrel[j] = e / fabs(ref[j]);
        }
    }
    for (int j = 0; j < 13; ++j) printf("%-16s max abs err %.3e   max rel err %.3e\n", names[j], maxabs[j], maxrel[j]);
    return 0;
}
