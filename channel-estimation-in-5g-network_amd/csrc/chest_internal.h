// Host <-> kernel shared parameter blocks and the context object behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/chest.h"

namespace chest {

constexpr int KT = 9;   // max EFFECTIVE taps (distinct delays) a kernel keeps in registers; EPA 5, EVA 8, ETU 9
constexpr int KTS = 10;  // float2 stride of one tap vector in the frame-state record (80-B, 16-B aligned rows)

// Frame-state record written by taps_kernel and read (scalar loads) by the frame kernel:
//   float scal[ST_SCAL]; float2 c[S][KTS][R][T]; float2 csum[S][KTS][R]
// Tap-major inside a symbol: the gains of ONE tap for all antenna pairs are contiguous (one s_load per tap).
constexpr int ST_SCAL = 8;
enum { ST_NOISE_STD = 0, ST_SIGMA2 = 1, ST_SIGNAL_POWER = 2, ST_SUM_H2 = 3, ST_SUM_VAR = 4, ST_VALID = 5 };
__host__ __device__ inline size_t frame_state_floats(int S, int R, int T) {
    return (size_t)ST_SCAL + 2 * (size_t)KTS * ((size_t)S * R * T + (size_t)S * R);
}
__host__ __device__ inline int state_c_index(int s, int d, int r, int t, int R, int T) { return ((s * KTS + d) * R + r) * T + t; }
__host__ __device__ inline int state_csum_index(int s, int d, int r, int R) { return (s * KTS + d) * R + r; }

// Tap profile after the delay-collision overwrite (quirk Q1, src/channel_simulator.py:125):
// one entry per distinct delay, carrying the LAST path that rounds to it.
struct TapProfile {
    int32_t ntaps;                 // distinct delays
    int32_t npaths;                // paths in the table (they all consume random draws)
    int32_t max_delay;
    int32_t delay[KT];
    int32_t path[KT];
    float amp[KT];                 // sqrt(p_l) / sqrt(2*20)
    float step[KT][2];             // exp(-2 pi j delay/N): twiddle ratio of two adjacent used bins on the same side of DC
};

struct FrameParams {               // passed by value (kernarg segment -> scalar loads)
    int32_t N, cp, S, U, nsc, T, R, G, half;
    float ts[CHEST_MAX_SYMBOLS];   // symbol start time s*(N+cp)/fs, seconds
    TapProfile prof[CHEST_NUM_PROFILES];
};

// Which resource elements are pilots, 8 bytes per 32 elements: word w = {bit mask of elements 32 w .. 32 w + 31, number of pilots
// before element 32 w}.  The draw slot of element g (insert_pilots order, src/channel_simulator.py:249-252: the i-th pilot is the
// i-th True of the row-major mask, the data symbols follow in row-major order) is rank = .y + popcount(.x below bit g & 31) for a
// pilot and P + g - rank for a data element (chest_device.h: draw_slot).  [G / 32 rounded up, + 1 zero word]: 2.1 KB per 14 x 599
// pattern (round 5; a u16 slot per element was 16.8 KB of every per-frame pattern's 64 KB).
__host__ __device__ inline int pattern_bit_words(int G) { return (G + 31) / 32 + 1; }
struct PatternDev {
    const uint2* bits;             // [pattern_bit_words(G)]; nullptr: slot not set
    const void* plan;              // [G] 16-B records {w0, w1, i0|i1<<16, i2|valid<<16} or nullptr; compact plans: [2 Pcap] triangle records
    const int32_t* pilot_idx;      // [P]
    int32_t P;
    int32_t pad;                   // 0, or (compact plan) the byte offset from `plan` to the u16 triangle id of every element
};

struct PatternHost {
    void* block = nullptr;         // one hipMalloc per pattern (nullptr for patterns that live in a pool)
    PatternDev dev{};
};

// Storage of a batch of device-built patterns (chest_patterns_build): one allocation for `count` equally sized blocks
// slot u16[G] | pilot_idx i32[Pcap] | plan PlanRec[G], reused in place when the same slot range is rebuilt.
struct PatternPool {
    int32_t first_slot = 0, count = 0, Pcap = 0;
    size_t stride = 0, o_pil = 0, o_plan = 0, o_tri = 0;
    bool compact = false;          // plan blocks hold triangle records + per-element triangle ids (chest_patterns_build method 2)
    void* block = nullptr;
    void* d_meta = nullptr;        // per-pattern pilot counts and pattern ids on the device
};

}  // namespace chest

struct chest_ctx {
    chest_config cfg;
    chest::FrameParams fp;
    double powers_linear[CHEST_NUM_PROFILES][CHEST_MAX_PATHS];
    int32_t delay_samples[CHEST_NUM_PROFILES][CHEST_MAX_PATHS];
    int32_t num_paths[CHEST_NUM_PROFILES];
    std::vector<chest::PatternHost> patterns;
    std::vector<chest::PatternPool> pools;
    std::vector<float2> gram_host;
    chest::PatternDev* d_patterns = nullptr;   // device mirror, capacity pattern_cap
    int32_t pattern_cap = 0;
    int32_t* d_status = nullptr;   // device error word (bad descriptor seen by a kernel)
    float2* d_twiddle = nullptr;   // FFT twiddles exp(-2 pi j n/N), n < N (power-of-two N only)
    float2* d_gram = nullptr;      // [3][KT][KT] Gram matrices of the tap twiddles over the used bins
    chest::FrameParams* d_fp = nullptr;   // device copy of fp (kernels read it through uniform scalar loads)
    struct ProfSpan { int32_t id; void* a; void* b; };     // hipEvent_t pair around one kernel (group) of kind `id` (CHEST_K_*)
    std::vector<ProfSpan> prof_spans;     // pool created by chest_profile_begin; launches never allocate
    int32_t prof_used = 0;                // spans recorded since chest_profile_begin / the last chest_profile_read
    size_t lds_opt_in[40] = {};   // largest dynamic-LDS size already opted in per kernel (per context = per device)
    // chest_tune (tools / experiments only; nothing on the launch path reads the environment)
    int32_t tune_rx_group = 0;            // rx antennas per slab unit of the frame kernel (0 = chosen from the LDS budget)
    int32_t tune_taps_threads = 0;        // taps kernel workgroup size 64 / 128 / 256 (0 = by antenna count)
    int32_t tune_interp_mode = 0;         // interp_from_pilots_tfold_kernel<MODE>: 2 / 3 diagnostic (no plan reads / no stores); 0 = the product
    int32_t tune_interp_touch = 0;        // interp_from_pilots_tfold_kernel: rows ahead whose pilot values a workgroup touches (0 = one CU-slot generation, -1 = off)
    int32_t tune_interp_xcd = 0;          // -1: interp_from_pilots_tfold_kernel without the XCD-contiguous row ranges (A/B)
    int32_t tune_gemm_f32 = 0;            // 1: the full-covariance filter on the exact-f32 MFMA kernel (cgemm_w_kernel_3m) instead of the fp16-split one (A/B)
    int32_t tune_two_slabs = 0;           // -1: no second slab for the one-workgroup-per-CU frame kernel (A/B)
    int32_t tune_store_extra = 0;         // dedicated store waves per frame workgroup (0 = the default of the launcher, -1 = none); experiments
    int32_t tune_store_waves = 0;         // waves of a frame workgroup that stream the slab out (0 = the rule in slab_body); experiments
    int32_t tune_min_waves = 0;           // A/B of frame-kernel instantiations by register budget: 5 = 96 VGPRs (5-wave workgroups), 4 = 128 (2x2; also the vector-ALU form of the 2048-FFT 4x4 launch), 6 = 80 (SISO)
    int32_t tune_pattern_extra_lds = 0;   // bytes of unused LDS added to the pattern kernel's allocation (occupancy experiments)
    int32_t tune_pattern_solo_at = 0;     // active triangles at which the pattern kernel's flips go to one wave (0 = default 192)
    int32_t tune_pattern_stop_after = 0;  // diagnostic: the pattern kernel leaves after phase k (instruction counts per phase; tools/pmc_pattern_phases.sh)
    int32_t tune_pattern_flip_cap = 0;    // flip rounds per stage of the pattern kernel (0 = default); tests force the status-3 path with it
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

enum { LDS_TAPS = 0, LDS_FRAME = 1, LDS_ESTIMATE = 2, LDS_INTERP = 3, LDS_GEMM = 4, LDS_GEMM3M = 5, LDS_FRAME_640 = 6,
       LDS_FRAME_1024 = 7, LDS_FRAME_ROWS = 8, LDS_DELAUNAY = 9, LDS_FRAME_44 = 10, LDS_FRAME_11 = 11, LDS_FRAME_GEN = 12, LDS_FRAME_W5 = 13, LDS_GJ_PANEL = 14, LDS_GJ_UPDATE = 15, LDS_GJ_PANEL_R = 16, LDS_GJ_UPDATE_R = 17, LDS_FRAME_640_44 = 18, LDS_INTERP_TFOLD = 19,
       LDS_FRAME_CP = 20, LDS_FRAME_44_CP = 21, LDS_FRAME_11_CP = 22, LDS_FRAME_GEN_CP = 23, LDS_DELAUNAY_CP = 24, LDS_DELAUNAY_NEAREST = 25, LDS_FRAME_640_44_MF = 26, LDS_FRAME_W6 = 27, LDS_FRAME_CP_W5 = 28, LDS_FRAME_11_CP_W5 = 29, LDS_FRAME_11_W5 = 30, LDS_FRAME_11_W7 = 31, LDS_FRAME_44_W5 = 32,
       LDS_INTERP_TFOLD2 = 33, LDS_INTERP_TFOLD3 = 34 };

// RAII: make the context's device current for the duration of an entry point and restore the caller's afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == device) || hipSetDevice(device) == hipSuccess;
        if (prev == device) prev = -1;                            // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define CHEST_GUARD(ctx)                                                                     \
    chest::DeviceGuard guard_((ctx)->cfg.device);                                             \
    if (!guard_.ok) { chest::set_error("hipSetDevice(%d) failed", (ctx)->cfg.device); return CHEST_EHIP; }

// per-kernel timing spans (chest_profile_begin): record an event pair on the launch stream around a kernel group
inline int prof_open(chest_ctx* ctx, hipStream_t st, int id) {
    if ((size_t)ctx->prof_used >= ctx->prof_spans.size()) return -1;
    const int i = ctx->prof_used++;
    ctx->prof_spans[i].id = id;
    (void)hipEventRecord((hipEvent_t)ctx->prof_spans[i].a, st);
    return i;
}
inline void prof_close(chest_ctx* ctx, hipStream_t st, int i) {
    if (i >= 0) (void)hipEventRecord((hipEvent_t)ctx->prof_spans[i].b, st);
}
// kernels that use more than 64 KiB of dynamic LDS must opt in once per device; cached in the context
#define CHEST_LDS_OPT_IN(ctx, which, fn, bytes)                                                                    \
    do {                                                                                                       \
        if ((bytes) > 64 * 1024 && (bytes) > (ctx)->lds_opt_in[which]) {                                       \
            CHEST_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            (ctx)->lds_opt_in[which] = (bytes);                                                                \
        }                                                                                                      \
    } while (0)

// launchers implemented in the kernel translation units
int launch_frames(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, uint64_t seed,
                  uint64_t frame_offset, const chest_injected* inj, const chest_frame_outputs* out, void* workspace,
                  int64_t workspace_bytes);
size_t frames_workspace_bytes(const chest_ctx* ctx, int64_t B);
int launch_k2_slab(chest_ctx* ctx, hipStream_t st, int32_t B, const chest_frame_desc* descs, int Pmax, uint64_t seed,
                   uint64_t frame_offset, const chest_injected& inj, const chest_frame_outputs& out, const float* workspace, bool compact);
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
int launch_tdl(chest_ctx* ctx, hipStream_t st, int32_t type, int64_t ns, int32_t T, int32_t R, const float* x, const float* h,
               float* y);
int launch_pack_ml(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue,
                   const uint8_t* mask, int32_t normalize, float* inputs, float* targets);
int64_t ml_stats_workspace_bytes();
int launch_ml_stats(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue, double* stats,
                    void* workspace, int64_t workspace_bytes);
int launch_pack_train(chest_ctx* ctx, hipStream_t st, int32_t B, const float* rx, const float* Hls, const float* Htrue,
                      const void* mask, int32_t mask_is_f32, int32_t mode, const double* stats, float* inputs, float* targets,
                      float* mask_out);
int launch_plan_locate(chest_ctx* ctx, hipStream_t st, const int32_t* d_pilot_idx, const int32_t* d_simp, int32_t ns, void* d_packed,
                       void* d_plan);
struct QamTable;
int qam_table(int32_t M, QamTable* t);
int launch_qam_mod(chest_ctx* ctx, hipStream_t st, int32_t M, int64_t n, const uint8_t* bits, int32_t wide, void* out);
int launch_qam_demod(chest_ctx* ctx, hipStream_t st, int32_t M, int64_t n, int32_t wide, const void* sym, uint8_t* bits);
int launch_bit_errors(chest_ctx* ctx, hipStream_t st, int64_t n, const uint8_t* a, const uint8_t* b, uint64_t* count);
int qam_constellation_host(int32_t M, double* points, int32_t* gray_map);
}  // namespace chest
