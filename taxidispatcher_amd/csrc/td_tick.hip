// td_tick.hip — one dispatcher tick behind ONE C-ABI call (BASELINE configs[4]).
//
// Replaces, for one time step, Simulator.java:163-208 (createTempDemand / createTempSupply have already produced
// the position arrays): calculate_cost (:493-520) -> LCM down to MAX_NON_LCM rows (:523-549) -> analyzePairs'
// removal of the matched cabs and requests (:613-674; filter_out of greedy_opt.py:32-37, simulate.py:64-69) ->
// calculate_cost of the remainder -> optimal assignment (the GLPK call of solver.py:26).
//
// Everything stays in HBM: the two cost matrices are library buffers, the shrink (a-6) is a compaction kernel
// over the LCM pair list that td_lcm leaves on the device, and what crosses PCIe per tick is the position arrays
// in, the pair list / kept indices / row_to_col out (a few KiB).  Host synchronisations per tick: the two inside
// td_lcm (level-list sizing + result), the one at the end of td_assign.
#include <limits.h>

#include "td_common.h"

using namespace td;

namespace {

struct TickBufs {
    Buf cost_a, cost_b, pos, keep;
    void *pin = nullptr;
    size_t pin_cap = 0;
};
TickBufs g_tick;

// kept[] = the cabs (requests) that are in no LCM pair, in their order; pos2[] = their positions.
// One workgroup: membership bitsets in LDS, then an ordered compaction (contiguous slice per thread + scan).
__global__ __launch_bounds__(1024) void k_tick_shrink(int n_s, int n_d, int k, const int32_t *__restrict__ rows,
                                                      const int32_t *__restrict__ cols, const int32_t *__restrict__ cab_to,
                                                      const int32_t *__restrict__ dem_from, int32_t *__restrict__ cab2,
                                                      int32_t *__restrict__ dem2, int32_t *__restrict__ keep_c,
                                                      int32_t *__restrict__ keep_d, int32_t *__restrict__ counts)
{
    extern __shared__ uint32_t s_bits[];   // [(n_s + 31) / 32] + [(n_d + 31) / 32]
    __shared__ int s_w[16];
    __shared__ int s_tot;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ws = (n_s + 31) / 32, wd = (n_d + 31) / 32;
    for (int i = tid; i < ws + wd; i += T) s_bits[i] = 0;
    __syncthreads();
    for (int i = tid; i < k; i += T) {
        const int r = rows[i], c = cols[i];
        if (r >= 0 && r < n_s) atomicOr(&s_bits[r >> 5], 1u << (r & 31));
        if (c >= 0 && c < n_d) atomicOr(&s_bits[ws + (c >> 5)], 1u << (c & 31));
    }
    __syncthreads();
    for (int side = 0; side < 2; side++) {
        const int n = side ? n_d : n_s;
        const uint32_t *bits = s_bits + (side ? ws : 0);
        const int32_t *pos = side ? dem_from : cab_to;
        int32_t *pos2 = side ? dem2 : cab2, *keep = side ? keep_d : keep_c;
        const int per = (n + T - 1) / T, lo = tid * per, hi = min(n, lo + per);
        int cnt = 0;
        for (int i = lo; i < hi; i++) cnt += ((bits[i >> 5] >> (i & 31)) & 1u) ? 0 : 1;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < w; q++) base += s_w[q];
        int at = base + incl - cnt;
        for (int i = lo; i < hi; i++)
            if (!((bits[i >> 5] >> (i & 31)) & 1u)) {
                keep[at] = i;
                pos2[at] = pos[i];
                at++;
            }
        if (tid == T - 1) s_tot = base + incl;
        __syncthreads();
        if (tid == 0) counts[side] = s_tot;
        __syncthreads();
    }
}

}  // namespace

extern "C" int td_tick(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                       int32_t threshold, int stop_size, int32_t *lcm_rows, int32_t *lcm_cols, int32_t *n_pairs,
                       int32_t *lcm_last_min, int32_t *kept_cabs, int32_t *kept_dems, int32_t *n_rest, int32_t *row_to_col,
                       int64_t *total)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n_s < 0 || n_d < 0) return fail(TD_EINVAL, "negative size");
    if (n_pairs) *n_pairs = 0;
    if (n_rest) *n_rest = 0;
    if (total) *total = 0;
    const int n = std::max(n_s, n_d);
    if (n == 0) return TD_OK;   // simulate.py:21 / Simulator.java:499: an empty model
    if ((n_s && !cab_to) || (n_d && !dem_from)) return fail(TD_EINVAL, "null position array");
    if (!lcm_rows || !lcm_cols || !n_pairs || !n_rest || !row_to_col || !total) return fail(TD_EINVAL, "null output");
    if (is_device_ptr(lcm_rows) || is_device_ptr(lcm_cols) || is_device_ptr(row_to_col) || (kept_cabs && is_device_ptr(kept_cabs)) ||
        (kept_dems && is_device_ptr(kept_dems)))
        return fail(TD_EINVAL, "td_tick hands its (small) results to HOST arrays");
    TickBufs &t = g_tick;
    int rc;
    if ((rc = ensure(t.pos, sizeof(int32_t) * 4 * (size_t)n + (dist && !is_device_ptr(dist) ? sizeof(int32_t) * (size_t)S * S : 0)))) return rc;
    if ((rc = ensure(t.keep, sizeof(int32_t) * (2 * (size_t)n + 4)))) return rc;
    const size_t pin_need = sizeof(int32_t) * (4 * (size_t)n + 8);   // kept cabs, kept requests, 2 counts (+2), then the pair list (rows, cols) of the stands LCM
    if (t.pin_cap < pin_need) {
        if (t.pin) (void)hipHostFree(t.pin);
        t.pin = nullptr;
        t.pin_cap = 0;
        TD_HIP(hipHostMalloc(&t.pin, pin_need, hipHostMallocDefault));
        t.pin_cap = pin_need;
    }
    // position arrays (and a host distance table) once on the device: both cost builds and the shrink read them there
    int32_t *d_pos = (int32_t *)t.pos.p;
    int32_t *d_cab = d_pos, *d_dem = d_pos + n, *d_cab2 = d_pos + 2 * (size_t)n, *d_dem2 = d_pos + 3 * (size_t)n;
    auto put = [&](const int32_t *src, int cnt, int32_t *dst) -> int {
        if (cnt == 0) return TD_OK;
        const void *d = nullptr;
        Buf tmp;   // to_device() stages small host arrays through the pinned ring into `tmp`... keep it simple: direct copies
        (void)tmp;
        (void)d;
        TD_HIP(hipMemcpyAsync(dst, src, sizeof(int32_t) * (size_t)cnt, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              c.stream));
        return TD_OK;
    };
    // small host arrays go through the library's pinned ring when they fit (a pageable async copy blocks the host)
    auto put_small = [&](const int32_t *src, int cnt, int32_t *dst) -> int {
        const size_t bytes = sizeof(int32_t) * (size_t)cnt;
        if (cnt > 0 && !is_device_ptr(src) && c.pin_in && bytes <= 32768) {
            const size_t need = (bytes + 255) & ~(size_t)255;
            if (c.pin_in_off + need > c.pin_in_cap) {
                TD_HIP(hipStreamSynchronize(c.stream));
                c.pin_in_off = 0;
            }
            void *slot = (char *)c.pin_in + c.pin_in_off;
            c.pin_in_off += need;
            memcpy(slot, src, bytes);
            TD_HIP(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c.stream));
            return TD_OK;
        }
        return put(src, cnt, dst);
    };
    if (n_s > 0 && n_d > 0 && !is_device_ptr(cab_to) && !is_device_ptr(dem_from) && c.pin_in && sizeof(int32_t) * 2 * (size_t)n <= 32768) {
        // both position arrays in ONE staged copy (d_cab and d_dem are n apart)
        const size_t bytes = sizeof(int32_t) * ((size_t)n + n_d), need = (bytes + 255) & ~(size_t)255;
        if (c.pin_in_off + need > c.pin_in_cap) {
            TD_HIP(hipStreamSynchronize(c.stream));
            c.pin_in_off = 0;
        }
        int32_t *slot = (int32_t *)((char *)c.pin_in + c.pin_in_off);
        c.pin_in_off += need;
        memcpy(slot, cab_to, sizeof(int32_t) * (size_t)n_s);
        memcpy(slot + n, dem_from, sizeof(int32_t) * (size_t)n_d);
        TD_HIP(hipMemcpyAsync(d_cab, slot, bytes, hipMemcpyHostToDevice, c.stream));
    } else {
        if ((rc = put_small(cab_to, n_s, d_cab))) return rc;
        if ((rc = put_small(dem_from, n_d, d_dem))) return rc;
    }
    const int32_t *d_dist = dist;
    if (dist && !is_device_ptr(dist)) {
        int32_t *dd = d_pos + 4 * (size_t)n;
        if ((rc = put_small(dist, S * S, dd))) return rc;
        d_dist = dd;
    }
    int k = 0;
    int64_t lcm_total = 0;
    int32_t last_min = fill;
    // the Simulator's own model — |a - b| below DROP_TIME on <= 64 stands — needs no matrix for its LCM (k_lcm_stands)
    int on_stands = 0;
    static const bool use_stands = !(getenv("TD_LCM_STANDS") && atoi(getenv("TD_LCM_STANDS")) == 0);
    if (use_stands && !dist && stop_size >= 0 && stop_size < n) {
        if ((rc = td::lcm_stands(n_s, n_d, d_cab, d_dem, fill, threshold, stop_size, lcm_rows, lcm_cols, &k, &last_min, &on_stands))) return rc;
    }
    int32_t *d_a = nullptr;
    if (!on_stands && stop_size >= 0 && stop_size < n) {
        if ((rc = ensure(t.cost_a, sizeof(int32_t) * (size_t)n * n))) return rc;
        d_a = (int32_t *)t.cost_a.p;
        if ((rc = td::cost_build_async(d_cab, n_s, d_dem, n_d, d_dist, S, fill, threshold, d_a))) return rc;
    }
    if (!on_stands && stop_size >= 0 && stop_size < n) {
        // Simulator.java:523-549: stop on big_cost or when MAX_NON_LCM rows are left; dummies are never summed
        // every real cell of a thresholded model lies in 0 .. threshold - 1 (distances are not negative): the level lists
        // are laid out without a min / max pass; a table with a negative distance is caught on the device and redone
        const bool hint = threshold >= 1 && threshold <= 256;
        if ((rc = td::lcm_hinted(n, d_a, fill, -1, 1, fill, stop_size, (int64_t)fill, n, lcm_rows, lcm_cols, &k, &lcm_total, &last_min,
                                 hint ? 0 : INT_MAX, hint ? threshold - 1 : INT_MIN)))
            return rc;
    }
    *n_pairs = k;
    if (lcm_last_min) *lcm_last_min = last_min;
    // ---- a-6 on the device: the pair list is still in td_lcm's device buffers (rows, then cols, n entries each)
    int32_t *d_keep_c = (int32_t *)t.keep.p, *d_keep_d = d_keep_c + n, *d_counts = d_keep_c + 2 * (size_t)n;
    const int32_t *d_rows = (const int32_t *)c.lcm_b.p, *d_cols = d_rows ? d_rows + n : nullptr;
    const size_t shm = sizeof(uint32_t) * (size_t)((n_s + 31) / 32 + (n_d + 31) / 32 + 2);
    if (shm > 60 * 1024) return fail(TD_ERANGE, "td_tick: n=%d is beyond the one-workgroup shrink", n);
    k_tick_shrink<<<1, 1024, shm, c.stream>>>(n_s, n_d, k, d_rows, d_cols, d_cab, d_dem, d_cab2, d_dem2, d_keep_c, d_keep_d, d_counts);
    TD_HIP(hipGetLastError());
    // every LCM pair is a real (cab, request) cell (< fill), so the remainder's size is known without a read-back;
    // the device counts come home with the kept lists and are checked below
    const int kc = n_s - k, kd = n_d - k, n2 = std::max(kc, kd);
    if (kc < 0 || kd < 0) return fail(TD_EINTERNAL, "td_tick: more LCM pairs than cabs or requests");
    int32_t *hp = (int32_t *)t.pin;
    TD_HIP(hipMemcpyAsync(hp, d_keep_c, sizeof(int32_t) * (2 * (size_t)n + 2), hipMemcpyDeviceToHost, c.stream));
    int32_t *hpairs = hp + 2 * (size_t)n + 4;
    if (on_stands && k > 0)   // the stands LCM left its pair list on the device: home with the same pinned read-back
        TD_HIP(hipMemcpyAsync(hpairs, d_rows, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyDeviceToHost, c.stream));
    *n_rest = n2;
    // Simulator.java:188-189: when the LCM ended on big_cost ("no input for the solver; continue") the reference does
    // not call the solver in this tick.  The pairs, LCM_min_val and the kept lists are reported, row_to_col and
    // *total (0) are not touched; the caller sees it as lcm_last_min == fill with the LCM having run.
    const bool lcm_ran = stop_size >= 0 && stop_size < n;
    if (n2 > 0 && !(lcm_ran && last_min == fill)) {
        // the remainder's cost build + optimal assignment: its cells are made from the kept position arrays inside the fused
        // compress pass when the model is padded with dummy requests (every reference tick), the matrix is never written
        if ((rc = td::build_assign_device(d_cab2, kc, d_dem2, kd, d_dist, S, fill, threshold, true, row_to_col, total, nullptr))) return rc;   // ends with a stream synchronisation
    } else {
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    if (hp[2 * (size_t)n] != kc || hp[2 * (size_t)n + 1] != kd) return fail(TD_EINTERNAL, "td_tick: shrink kept %d / %d, expected %d / %d", hp[2 * (size_t)n], hp[2 * (size_t)n + 1], kc, kd);
    if (on_stands && k > 0) {
        memcpy(lcm_rows, hpairs, sizeof(int32_t) * (size_t)k);
        memcpy(lcm_cols, hpairs + n, sizeof(int32_t) * (size_t)k);
    }
    if (kept_cabs) memcpy(kept_cabs, hp, sizeof(int32_t) * (size_t)kc);
    if (kept_dems) memcpy(kept_dems, hp + n, sizeof(int32_t) * (size_t)kd);
    return TD_OK;
}

extern "C" void td_tick_release_workspace(void)
{
    TickBufs &t = g_tick;
    Buf *bs[] = {&t.cost_a, &t.cost_b, &t.pos, &t.keep};
    for (Buf *b : bs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    if (t.pin) (void)hipHostFree(t.pin);
    t.pin = nullptr;
    t.pin_cap = 0;
}
