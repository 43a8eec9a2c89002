// td_common.h — shared host-side context for the gfx950 assignment library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#include "taxidispatcher_amd.h"

namespace td {

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Ctx {
    bool inited = false;
    int device = -1;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // grow-only device workspace, reused across calls (no hipMalloc on the hot path)
    Buf stage_a, stage_b, stage_c, stage_d, stage_out;  // host<->device staging
    Buf cc;                                             // narrow code matrix of td_lcm / td_pool2
    Buf misc;                                           // small scratch (td_count_sum)
    Buf lcm_a, lcm_b, lcm_c, lcm_d;
    void *pinned = nullptr;  // small pinned host block for result read-back
    size_t pinned_cap = 0;
    // pinned bounce ring for SMALL host inputs (position arrays): a pageable hipMemcpyAsync blocks the host for
    // ~10 us per call, a host memcpy into this ring + an async copy from it does not
    void *pin_in = nullptr;
    size_t pin_in_cap = 0, pin_in_off = 0;
    // profiling
    bool prof = false;
    double prof_ms[TD_K_COUNT] = {0};
    int64_t prof_n[TD_K_COUNT] = {0};
    struct Pending {
        hipEvent_t a, b;
        int k;
    };
    Pending pend[4096];
    int n_pend = 0;
    hipEvent_t ev_pool[8192];
    int n_ev = 0, ev_next = 0;
    int64_t stats[16] = {0};
};

Ctx &ctx();
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
int ensure(Buf &b, size_t bytes);
bool is_device_ptr(const void *p);
// returns a device pointer holding `bytes` of src (src itself when already on the device)
int to_device(const void *src, size_t bytes, Buf &stage, const void **out);
// td_line.hip: sorted matching + certificate pass for line-metric matrices (tried first by td_assign)
//   line_probe_launch  queues the O(n) probe; *skip_dev = device word that is non-zero unless the probe refuses
//   line_probe_wait    waits for the probe alone (an event, not the stream) and returns its verdict:
//                      0 refused, 1 plausible, 2 constant trailing columns (retry on the transpose),
//                      3 plausible with *k constant rows (the unbalanced model)
//   line_finish        keys, sort, prices, certificate pass; *accepted = 1: *r2c_dev / *total are proven optimal
int line_probe_launch(int n, const int32_t *d_cost, const long long **skip_dev);
int line_probe_wait(int *mode, int *k, int *suspicious = nullptr /* the 1-byte attempt is void: the probe made the queued pass a no-op */,
                    int *shape3 = nullptr /* int[4]: estimated constant columns / rows, row 0 too wide for one byte, value of the last column */);
int line_finish(int n, int k, const int32_t *d_cost, const int32_t **r2c_dev, int64_t *total, int *accepted);
void line_release_workspace();
// td_assign.hip: a hint for the NEXT td_assign call of this process (consumed by it): the matrix is a model padded with
// `const_cols` dummy requests and `const_rows` dummy cabs of value `fill` (td_tick knows this from its position arrays)
void assign_hint_padded(int const_cols, int const_rows, int32_t fill);
// td_assign.hip: cost build + optimal assignment from DEVICE position arrays (td_build_assign after staging, td_tick's remainder);
// a model padded with dummy requests never exists as an int32 matrix (its cells are made inside the fused compress pass)
int build_assign_device(const int32_t *d_cab, int n_s, const int32_t *d_dem, int n_d, const int32_t *d_dist, int S, int32_t fill,
                        int32_t threshold, bool tick, int32_t *row_to_col, int64_t *total, int64_t *dual_bound);
// td_lcm.hip: td_lcm with the candidate cells' value range given by the caller (no min / max pass, no host round trip);
// a wrong hint is detected on the device and the call is redone with the measured range
int lcm_hinted(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on, int32_t stop_value, int stop_size,
               int64_t sum_below, int max_pairs, int32_t *rows, int32_t *cols, int32_t *n_pairs, int64_t *total, int32_t *last_min,
               int hint_vmin, int hint_vmax);
// td_core.hip: td_cost_build for device-resident library buffers, without the trailing stream synchronisation (td_tick)
int cost_build_async(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                     int32_t threshold, int32_t *cost);
// td_lcm.hip: td_tick's LCM of a thresholded |a - b| model on <= 64 stands straight from the position arrays (no matrix);
// *ok = 0: not such a model, nothing done
int lcm_stands(int n_s, int n_d, const int32_t *d_cab_to, const int32_t *d_dem_from, int32_t fill, int32_t threshold, int stop_size,
               int32_t *rows, int32_t *cols, int32_t *n_pairs, int32_t *last_min, int *ok);
void prof_begin(int k);
void prof_end(int k);
void prof_flush();

#define TD_HIP(call)                                       \
    do {                                                   \
        hipError_t _e = (call);                            \
        if (_e != hipSuccess) return td::hip_fail(_e, #call); \
    } while (0)

#define TD_REQUIRE_INIT()                                                   \
    do {                                                                    \
        if (!td::ctx().inited) return td::fail(TD_ENOINIT, "td_init() has not been called"); \
    } while (0)

struct ProfScope {
    int k;
    explicit ProfScope(int k_) : k(k_) { prof_begin(k); }
    ~ProfScope() { prof_end(k); }
};

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace td
