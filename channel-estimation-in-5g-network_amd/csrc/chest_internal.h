// Host <-> kernel shared parameter blocks and the context object behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/chest.h"

namespace chest {

constexpr int KT = 9;   // max EFFECTIVE taps (distinct delays) a kernel keeps in registers; EPA 5, EVA 8, ETU 9

// Tap profile after the delay-collision overwrite (quirk Q1, src/channel_simulator.py:125):
// one entry per distinct delay, carrying the LAST path that rounds to it.
struct TapProfile {
    int32_t ntaps;                 // distinct delays
    int32_t npaths;                // paths in the table (they all consume random draws)
    int32_t max_delay;
    int32_t delay[KT];
    int32_t path[KT];
    float amp[KT];                 // sqrt(p_l) / sqrt(2*20)
};

struct FrameParams {               // passed by value (kernarg segment -> scalar loads)
    int32_t N, cp, S, U, nsc, T, R, G, half;
    float ts[CHEST_MAX_SYMBOLS];   // symbol start time s*(N+cp)/fs, seconds
    TapProfile prof[CHEST_NUM_PROFILES];
};

struct PatternDev {
    const uint16_t* slot;          // [G] draw slot of each RE: < P pilot ordinal, else P + data ordinal
    const uint16_t* plan_idx;      // [3][G] or nullptr
    const float* plan_w;           // [3][G] or nullptr
    const int32_t* pilot_idx;      // [P]
    int32_t P;
    int32_t pad;
};

struct PatternHost {
    void* block = nullptr;         // one hipMalloc per pattern
    PatternDev dev{};
};

}  // namespace chest

struct chest_ctx {
    chest_config cfg;
    chest::FrameParams fp;
    double powers_linear[CHEST_NUM_PROFILES][CHEST_MAX_PATHS];
    int32_t delay_samples[CHEST_NUM_PROFILES][CHEST_MAX_PATHS];
    int32_t num_paths[CHEST_NUM_PROFILES];
    std::vector<chest::PatternHost> patterns;
    chest::PatternDev* d_patterns = nullptr;   // device mirror, capacity pattern_cap
    int32_t pattern_cap = 0;
    int32_t block_threads = 0;     // workgroup size of the per-frame kernels
    int32_t* d_status = nullptr;   // device error word (bad descriptor seen by a kernel)
    float2* d_twiddle = nullptr;   // FFT twiddles exp(-2 pi j n/N), n < N (built lazily)
};

namespace chest {
void set_error(const char* fmt, ...);
#define CHEST_HIP(call)                                                              \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            chest::set_error("%s failed: %s", #call, hipGetErrorString(e_));         \
            return CHEST_EHIP;                                                       \
        }                                                                            \
    } while (0)

// launchers implemented in the kernel translation units
int launch_frames(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, uint64_t seed,
                  uint64_t frame_offset, const chest_injected* inj, const chest_frame_outputs* out);
int launch_estimate(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx4d, const float* tx_pilots,
                    int32_t pilot_stride, int32_t slot, int32_t mmse, float snr_db, float* H_est);
int launch_at_pilots(chest_ctx* ctx, hipStream_t st, int32_t n, const float* rx, const float* tx, int32_t mmse,
                     float snr_db, float* out);
int launch_error_sums(chest_ctx* ctx, hipStream_t st, int64_t n, const float* a, const float* b, double* out);
int launch_apply_channel(chest_ctx* ctx, hipStream_t st, int32_t B, const float* tx, const float* H,
                         const float* snr_db, const float* n_re, const float* n_im, uint64_t seed,
                         uint64_t frame_offset, float* rx);
int launch_tv_channel(chest_ctx* ctx, hipStream_t st, int32_t type, float fd, int64_t ns, int32_t T, int32_t R,
                      const float* ua, const float* up, uint64_t seed, uint64_t frame, float* out);
int launch_ofdm(chest_ctx* ctx, hipStream_t st, int32_t n, const float* in, float* out, bool modulate);
int launch_equalize(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* H, int32_t method,
                    float* out);
}  // namespace chest
