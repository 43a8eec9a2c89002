// td_core.hip — context, staging, profiling, cost-matrix build, synthetic
// generator (a-10), x expansion and objective evaluation (a-7).  gfx950 only.
#include <stdarg.h>

#include "td_common.h"

namespace td {

static Ctx g_ctx;
Ctx &ctx() { return g_ctx; }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ctx.err, sizeof(g_ctx.err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_ctx.err, sizeof(g_ctx.err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return TD_EHIP;
}

int ensure(Buf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return TD_OK;
    if (b.p) {
        hipError_t e0 = hipStreamSynchronize(g_ctx.stream);
        if (e0 != hipSuccess) return hip_fail(e0, "hipStreamSynchronize(before realloc)");
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return hip_fail(e, "hipMalloc(workspace)");
    }
    b.cap = want;
    return TD_OK;
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int to_device(const void *src, size_t bytes, Buf &stage, const void **out)
{
    if (is_device_ptr(src)) {
        *out = src;
        return TD_OK;
    }
    int rc = ensure(stage, bytes);
    if (rc) return rc;
    Ctx &c = g_ctx;
    if (c.pin_in && bytes > 0 && bytes <= 32768) {
        // small input: through the pinned ring (the slot is reused only after a stream synchronisation at wrap-around)
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (c.pin_in_off + need > c.pin_in_cap) {
            TD_HIP(hipStreamSynchronize(c.stream));
            c.pin_in_off = 0;
        }
        void *slot = (char *)c.pin_in + c.pin_in_off;
        c.pin_in_off += need;
        memcpy(slot, src, bytes);
        TD_HIP(hipMemcpyAsync(stage.p, slot, bytes, hipMemcpyHostToDevice, c.stream));
    } else {
        TD_HIP(hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, c.stream));
    }
    *out = stage.p;
    return TD_OK;
}

static hipEvent_t next_event()
{
    Ctx &c = g_ctx;
    if (c.ev_next < c.n_ev) return c.ev_pool[c.ev_next++];
    if (c.n_ev < 8192) {
        hipEvent_t e;
        if (hipEventCreate(&e) == hipSuccess) {
            c.ev_pool[c.n_ev++] = e;
            c.ev_next = c.n_ev;
            return e;
        }
    }
    return nullptr;
}

void prof_begin(int k)
{
    Ctx &c = g_ctx;
    if (!c.prof) return;
    if (c.n_pend >= 4096) prof_flush();
    hipEvent_t a = next_event(), b = next_event();
    if (!a || !b) {
        c.pend[c.n_pend].k = -1;   // no event left: prof_end must not match a stale entry
        return;
    }
    c.pend[c.n_pend] = {a, b, k};
    (void)hipEventRecord(a, c.stream);
}

void prof_end(int k)
{
    Ctx &c = g_ctx;
    if (!c.prof) return;
    if (c.n_pend >= 4096 || c.pend[c.n_pend].k != k) return;
    (void)hipEventRecord(c.pend[c.n_pend].b, c.stream);
    c.n_pend++;
}

void prof_flush()
{
    Ctx &c = g_ctx;
    if (c.n_pend == 0) return;
    (void)hipStreamSynchronize(c.stream);
    for (int i = 0; i < c.n_pend; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c.pend[i].a, c.pend[i].b) == hipSuccess) {
            c.prof_ms[c.pend[i].k] += ms;
            c.prof_n[c.pend[i].k] += 1;
        }
    }
    c.n_pend = 0;
    c.ev_next = 0;
}

}  // namespace td

using namespace td;

// =====================================================================================
// kernels
// =====================================================================================

// a-2 cost build, positional.  Thread = 4 consecutive requests (columns) of one cab (row);
// the thread keeps its 4 request positions in registers and walks down the rows
// (blockIdx.y-strided), so dem_from[] is read once per thread, coalesced, and every store is
// one 16-byte int4 (VEC4) per lane = 1 KiB per wave instruction.
template <bool VEC4, bool LDS_DIST>
__global__ __launch_bounds__(256) void k_cost_build(const int32_t *__restrict__ cab_to,
                                                    const int32_t *__restrict__ cab_id, int n_s,
                                                    const int32_t *__restrict__ dem_from,
                                                    const int32_t *__restrict__ dem_id, int n_d,
                                                    const int32_t *__restrict__ dist, int S, int32_t fill,
                                                    int32_t thr, int n, int row0, int nrows,
                                                    int32_t *__restrict__ cost /* rows [row0, row0 + nrows) */)
{
    extern __shared__ int32_t s_dist[];
    if (LDS_DIST) {
        for (int k = threadIdx.x; k < S * S; k += blockDim.x) s_dist[k] = dist[k];
        __syncthreads();
    }
    const int nq = (n + 3) >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int d0 = q * 4;
    int32_t b[4];
    bool bv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        int d = d0 + e;
        bv[e] = d < n_d && (!dem_id || dem_id[d] != -1);
        b[e] = bv[e] ? dem_from[d] : 0;
        if (dist && (uint32_t)b[e] >= (uint32_t)S) bv[e] = false;  // never index outside the table
    }
    for (int rr = blockIdx.y; rr < nrows; rr += gridDim.y) {
        const int r = row0 + rr;
        int32_t v[4] = {fill, fill, fill, fill};
        bool rv = r < n_s && (!cab_id || cab_id[r] != -1);
        const int32_t a = rv ? cab_to[r] : 0;
        if (dist && (uint32_t)a >= (uint32_t)S) rv = false;
        if (rv) {
            const int32_t *drow = LDS_DIST ? (s_dist + a * S) : (dist ? dist + (int64_t)a * S : nullptr);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (bv[e]) {
                    int32_t x = drow ? drow[b[e]] : (a > b[e] ? a - b[e] : b[e] - a);
                    if (thr < 0 || x < thr) v[e] = x;
                }
            }
        }
        int32_t *dst = cost + (int64_t)rr * n + d0;
        if (VEC4) {
            *reinterpret_cast<int4 *>(dst) = make_int4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (d0 + e < n) dst[e] = v[e];
        }
    }
}

__global__ void k_fill_i32(int32_t *p, int64_t count, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = v;
}

// procedure.py:9-12 — cells addressed by id
__global__ void k_cost_scatter_by_id(const int32_t *cab_to, const int32_t *cab_id, int n_s,
                                     const int32_t *dem_from, const int32_t *dem_id, int n_d,
                                     const int32_t *dist, int S, int32_t thr, int n, int row0, int nrows, int32_t *cost)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    int c = blockIdx.y;
    if (d >= n_d || c >= n_s) return;
    int ci = cab_id[c], di = dem_id[d];
    if (ci < row0 || ci >= row0 + nrows || ci >= n || di < 0 || di >= n) return;
    ci -= row0;
    int a = cab_to[c], b = dem_from[d];
    if (dist && ((uint32_t)a >= (uint32_t)S || (uint32_t)b >= (uint32_t)S)) return;
    int32_t x = dist ? dist[(int64_t)a * S + b] : (a > b ? a - b : b - a);
    if (thr < 0 || x < thr) cost[(int64_t)ci * n + di] = x;
}

#ifndef TD_NT
#define TD_NT 0
#endif
static constexpr bool NT_STORES = (TD_NT & 1) != 0;   // bit 0: nontemporal cost stores, bit 1: nontemporal compress loads

// a-10 perf.jl-style uniform instance; thread = 4 cells, one int4 store.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_gen_uniform(int n, uint64_t seed, int32_t lo, uint32_t span, int row0,
                                                     int nrows, int32_t *__restrict__ cost)
{
    const int nq = (n + 3) >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint64_t base = seed * 0x100000001B3ull;
    for (int r = blockIdx.y; r < nrows; r += gridDim.y) {
        int32_t v[4];
        const uint64_t cell0 = (uint64_t)(row0 + r) * (uint64_t)n + (uint64_t)q * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint64_t h = splitmix64(base + cell0 + e);
            v[e] = lo + (int32_t)(((h >> 32) * (uint64_t)span) >> 32);
        }
        int32_t *dst = cost + (int64_t)r * n + q * 4;
        if (VEC4) {
            if (NT_STORES) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i pk4 = {v[0], v[1], v[2], v[3]};
                __builtin_nontemporal_store(pk4, reinterpret_cast<v4i *>(dst));
            } else
                *reinterpret_cast<int4 *>(dst) = make_int4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (q * 4 + e < n) dst[e] = v[e];
        }
    }
}

__global__ void k_expand_x(int n, const int32_t *row_to_col, uint8_t *x)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j < n && i < n) x[(int64_t)i * n + j] = (row_to_col[i] == j) ? 1 : 0;
}

__global__ void k_count_sum(int n, const int32_t *cost, const int32_t *row_to_col, int64_t big,
                            unsigned long long *out /* [0]=sum (as int64), [1]=count */)
{
    int64_t s = 0;
    int k = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int j = row_to_col[i];
        if (j >= 0 && j < n) {
            int64_t c = cost[(int64_t)i * n + j];
            if (c < big) {
                s += c;
                k++;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        k += __shfl_down(k, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)s);
        atomicAdd(&out[1], (unsigned long long)k);
    }
}

// =====================================================================================
// API
// =====================================================================================
extern "C" void td_assign_release_workspace(void);
extern "C" void td_tick_release_workspace(void);

extern "C" {

int td_version(void) { return 100; }

const char *td_last_error(void) { return ctx().err; }

int td_init(int device)
{
    Ctx &c = ctx();
    if (c.inited) {
        if (c.device == device) return TD_OK;
        td_shutdown();
    }
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(TD_EHIP, "no HIP device visible (hipGetDeviceCount -> %d, count %d); this library has no CPU path", (int)e, cnt);
    if (device < 0 || device >= cnt) return fail(TD_EINVAL, "device %d out of range [0,%d)", device, cnt);
    TD_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TD_HIP(hipGetDeviceProperties(&prop, device));
    c.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    TD_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    c.device = device;
    c.pinned_cap = 1 << 16;
    TD_HIP(hipHostMalloc(&c.pinned, c.pinned_cap, hipHostMallocDefault));
    c.pin_in_cap = 1 << 20;
    c.pin_in_off = 0;
    TD_HIP(hipHostMalloc(&c.pin_in, c.pin_in_cap, hipHostMallocDefault));
    c.inited = true;
    c.err[0] = 0;
    return TD_OK;
}

void td_shutdown(void)
{
    Ctx &c = ctx();
    if (!c.inited) return;
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    td_assign_release_workspace();
    td_tick_release_workspace();
    Buf *bufs[] = {&c.stage_a, &c.stage_b, &c.stage_c, &c.stage_d, &c.stage_out, &c.cc,
                   &c.misc,    &c.lcm_a,   &c.lcm_b,   &c.lcm_c,   &c.lcm_d};
    for (Buf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    for (int i = 0; i < c.n_ev; i++) (void)hipEventDestroy(c.ev_pool[i]);
    c.n_ev = c.ev_next = c.n_pend = 0;
    if (c.pinned) (void)hipHostFree(c.pinned);
    c.pinned = nullptr;
    if (c.pin_in) (void)hipHostFree(c.pin_in);
    c.pin_in = nullptr;
    c.pin_in_cap = c.pin_in_off = 0;
    if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
    c.own_stream = c.stream = nullptr;
    c.inited = false;
}

int td_set_stream(void *s)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    prof_flush();
    TD_HIP(hipStreamSynchronize(c.stream));
    c.stream = s ? (hipStream_t)s : c.own_stream;
    return TD_OK;
}

int td_synchronize(void)
{
    TD_REQUIRE_INIT();
    TD_HIP(hipStreamSynchronize(ctx().stream));
    return TD_OK;
}

int td_profile_enable(int on)
{
    TD_REQUIRE_INIT();
    prof_flush();
    ctx().prof = on != 0;
    return TD_OK;
}

int td_profile_reset(void)
{
    TD_REQUIRE_INIT();
    prof_flush();
    for (int k = 0; k < TD_K_COUNT; k++) {
        ctx().prof_ms[k] = 0;
        ctx().prof_n[k] = 0;
    }
    return TD_OK;
}

int td_profile_get(int k, double *ms, int64_t *launches)
{
    TD_REQUIRE_INIT();
    if (k < 0 || k >= TD_K_COUNT) return fail(TD_EINVAL, "kernel class %d out of range", k);
    prof_flush();
    if (ms) *ms = ctx().prof_ms[k];
    if (launches) *launches = ctx().prof_n[k];
    return TD_OK;
}

int td_last_stats(int64_t *out, int n)
{
    TD_REQUIRE_INIT();
    for (int i = 0; i < n && i < 16; i++) out[i] = ctx().stats[i];
    return TD_OK;
}

static int cost_build_impl(const int32_t *cab_to, const int32_t *cab_id, int n_s, const int32_t *dem_from,
                           const int32_t *dem_id, int n_d, const int32_t *dist, int S, int32_t fill, int32_t threshold,
                           int by_id, int row0, int nrows, int32_t *cost, bool nosync = false)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n_s < 0 || n_d < 0) return fail(TD_EINVAL, "negative sizes n_s=%d n_d=%d", n_s, n_d);
    const int n = std::max(n_s, n_d);
    if (n == 0) return TD_OK;  // simulate.py:21 / Simulator.java:499: empty model
    if (nrows < 0) nrows = n;   // the whole matrix
    if (row0 < 0 || row0 + nrows > n) return fail(TD_EINVAL, "row window [%d, %d) outside the %d x %d model", row0, row0 + nrows, n, n);
    if (nrows == 0) return TD_OK;
    if (!cost || (n_s && !cab_to) || (n_d && !dem_from)) return fail(TD_EINVAL, "null array");
    if (dist && S <= 0) return fail(TD_EINVAL, "dist given but S=%d", S);
    if (by_id && (!cab_id || !dem_id)) return fail(TD_EINVAL, "by_id needs cab_id and dem_id");
    const void *d_cab = nullptr, *d_cid = nullptr, *d_dem = nullptr, *d_did = nullptr, *d_dist = nullptr;
    int rc;
    // position arrays are tiny (<= 256 KiB); stage them when they are host pointers
    if (n_s && (rc = ensure(c.stage_a, sizeof(int32_t) * 2 * (size_t)std::max(n_s, 1)))) return rc;
    if (n_d && (rc = ensure(c.stage_b, sizeof(int32_t) * 2 * (size_t)std::max(n_d, 1)))) return rc;
    auto stage = [&](const int32_t *src, int cnt, Buf &buf, size_t off, const void **out) -> int {
        if (!src || cnt == 0) {
            *out = nullptr;
            return TD_OK;
        }
        if (is_device_ptr(src)) {
            *out = src;
            return TD_OK;
        }
        char *dst = (char *)buf.p + off;
        TD_HIP(hipMemcpyAsync(dst, src, sizeof(int32_t) * (size_t)cnt, hipMemcpyHostToDevice, c.stream));
        *out = dst;
        return TD_OK;
    };
    if ((rc = stage(cab_to, n_s, c.stage_a, 0, &d_cab))) return rc;
    if ((rc = stage(cab_id, n_s, c.stage_a, sizeof(int32_t) * (size_t)n_s, &d_cid))) return rc;
    if ((rc = stage(dem_from, n_d, c.stage_b, 0, &d_dem))) return rc;
    if ((rc = stage(dem_id, n_d, c.stage_b, sizeof(int32_t) * (size_t)n_d, &d_did))) return rc;
    if (dist) {
        if ((rc = to_device(dist, sizeof(int32_t) * (size_t)S * S, c.stage_c, &d_dist))) return rc;
    }
    const bool out_dev = is_device_ptr(cost);
    int32_t *d_cost = cost;
    const size_t cbytes = sizeof(int32_t) * (size_t)nrows * n;
    if (!out_dev) {
        if ((rc = ensure(c.stage_out, cbytes))) return rc;
        d_cost = (int32_t *)c.stage_out.p;
    }
    {
        ProfScope ps(TD_K_COST_BUILD);
        if (by_id) {
            k_fill_i32<<<std::min<int64_t>(((int64_t)nrows * n + 255) / 256, 8192), 256, 0, c.stream>>>(d_cost, (int64_t)nrows * n, fill);
            if (n_s && n_d) {
                dim3 g((n_d + 255) / 256, n_s);
                k_cost_scatter_by_id<<<g, 256, 0, c.stream>>>((const int32_t *)d_cab, (const int32_t *)d_cid, n_s,
                                                              (const int32_t *)d_dem, (const int32_t *)d_did, n_d,
                                                              (const int32_t *)d_dist, S, threshold, n, row0, nrows, d_cost);
            }
        } else {
            const int nq = (n + 3) / 4;
            const bool vec = (n % 4 == 0) && (((uintptr_t)d_cost & 15) == 0);
            const bool lds = d_dist && (size_t)S * S * 4 <= 64 * 1024;
            const size_t shm = lds ? (size_t)S * S * 4 : 0;
            // rows per block column: enough workgroups to fill 256 CUs several times over
            int gx = (nq + 255) / 256;
            // (more, shorter workgroups as in k_gen_uniform were measured: 231 -> 223 us at best, and the table-staging
            // variant would stage its table more often)
            int gy = std::min(nrows, std::max(1, (c.n_cu * 16) / gx));
            dim3 g(gx, gy);
#define TD_LAUNCH_CB(V, L)                                                                                            \
    k_cost_build<V, L><<<g, 256, shm, c.stream>>>((const int32_t *)d_cab, (const int32_t *)d_cid, n_s,               \
                                                  (const int32_t *)d_dem, (const int32_t *)d_did, n_d,               \
                                                  (const int32_t *)d_dist, S, fill, threshold, n, row0, nrows, d_cost)
            if (vec && lds) TD_LAUNCH_CB(true, true);
            else if (vec) TD_LAUNCH_CB(true, false);
            else if (lds) TD_LAUNCH_CB(false, true);
            else TD_LAUNCH_CB(false, false);
#undef TD_LAUNCH_CB
        }
    }
    TD_HIP(hipGetLastError());
    if (!out_dev) {
        TD_HIP(hipMemcpyAsync(cost, d_cost, cbytes, hipMemcpyDeviceToHost, c.stream));
    }
    if (!(nosync && out_dev)) TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

}   // extern "C"

// td_tick's cost builds: every array is a library buffer on the device and the next kernel is queued on the same stream,
// so the call returns without waiting (the public entry points wait: their callers may release the arrays at once)
int td::cost_build_async(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                         int32_t threshold, int32_t *cost)
{
    return cost_build_impl(cab_to, nullptr, n_s, dem_from, nullptr, n_d, dist, S, fill, threshold, 0, 0, -1, cost, true);
}

extern "C" {

int td_cost_build(const int32_t *cab_to, const int32_t *cab_id, int n_s, const int32_t *dem_from,
                  const int32_t *dem_id, int n_d, const int32_t *dist, int S, int32_t fill, int32_t threshold,
                  int by_id, int32_t *cost)
{
    return cost_build_impl(cab_to, cab_id, n_s, dem_from, dem_id, n_d, dist, S, fill, threshold, by_id, 0, -1, cost);
}

int td_cost_build_rows(const int32_t *cab_to, const int32_t *cab_id, int n_s, const int32_t *dem_from,
                       const int32_t *dem_id, int n_d, const int32_t *dist, int S, int32_t fill, int32_t threshold,
                       int by_id, int row0, int nrows, int32_t *cost_rows)
{
    if (nrows < 0) return fail(TD_EINVAL, "nrows < 0");
    return cost_build_impl(cab_to, cab_id, n_s, dem_from, dem_id, n_d, dist, S, fill, threshold, by_id, row0, nrows, cost_rows);
}

int td_gen_uniform(int n, uint64_t seed, int32_t lo, int32_t hi, int row0, int nrows, int32_t *cost)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n <= 0 || nrows < 0 || hi < lo || !cost) return fail(TD_EINVAL, "bad arguments to td_gen_uniform");
    if (nrows == 0) return TD_OK;
    const bool out_dev = is_device_ptr(cost);
    int32_t *d_cost = cost;
    const size_t cbytes = sizeof(int32_t) * (size_t)n * nrows;
    int rc;
    if (!out_dev) {
        if ((rc = ensure(c.stage_out, cbytes))) return rc;
        d_cost = (int32_t *)c.stage_out.p;
    }
    const int nq = (n + 3) / 4;
    const bool vec = (n % 4 == 0) && (((uintptr_t)d_cost & 15) == 0);
    int gx = (nq + 255) / 256;
    // workgroups per CU: many short workgroups (a few rows each) write faster than few long ones (16 -> 128: 218 -> 171 us at 16384^2)
    static const int gen_wg_per_cu = getenv("TD_GEN_GRID") ? std::max(1, atoi(getenv("TD_GEN_GRID"))) : 128;
    int gy = std::min(nrows, std::max(1, (c.n_cu * gen_wg_per_cu) / gx));
    dim3 g(gx, gy);
    {
        ProfScope ps(TD_K_GEN);
        if (vec)
            k_gen_uniform<true><<<g, 256, 0, c.stream>>>(n, seed, lo, (uint32_t)(hi - lo + 1), row0, nrows, d_cost);
        else
            k_gen_uniform<false><<<g, 256, 0, c.stream>>>(n, seed, lo, (uint32_t)(hi - lo + 1), row0, nrows, d_cost);
    }
    TD_HIP(hipGetLastError());
    if (!out_dev) TD_HIP(hipMemcpyAsync(cost, d_cost, cbytes, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

int td_expand_x(int n, const int32_t *row_to_col, uint8_t *x)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n < 0) return fail(TD_EINVAL, "n < 0");
    if (n == 0) return TD_OK;
    if (!row_to_col || !x) return fail(TD_EINVAL, "null array");
    const void *d_r2c;
    int rc;
    if ((rc = to_device(row_to_col, sizeof(int32_t) * (size_t)n, c.stage_a, &d_r2c))) return rc;
    const bool out_dev = is_device_ptr(x);
    uint8_t *d_x = x;
    if (!out_dev) {
        if ((rc = ensure(c.stage_out, (size_t)n * n))) return rc;
        d_x = (uint8_t *)c.stage_out.p;
    }
    dim3 g((n + 255) / 256, n);
    k_expand_x<<<g, 256, 0, c.stream>>>(n, (const int32_t *)d_r2c, d_x);
    TD_HIP(hipGetLastError());
    if (!out_dev) TD_HIP(hipMemcpyAsync(x, d_x, (size_t)n * n, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

int td_count_sum(int n, const int32_t *cost, const int32_t *row_to_col, int64_t big_cost, int64_t *sum,
                 int32_t *n_real)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n < 0) return fail(TD_EINVAL, "n < 0");
    if (n == 0) {
        if (sum) *sum = 0;
        if (n_real) *n_real = 0;
        return TD_OK;
    }
    if (!cost || !row_to_col) return fail(TD_EINVAL, "null array");
    const void *d_cost, *d_r2c;
    int rc;
    if ((rc = to_device(cost, sizeof(int32_t) * (size_t)n * n, c.stage_d, &d_cost))) return rc;
    if ((rc = to_device(row_to_col, sizeof(int32_t) * (size_t)n, c.stage_a, &d_r2c))) return rc;
    if ((rc = ensure(c.misc, 4096))) return rc;
    TD_HIP(hipMemsetAsync(c.misc.p, 0, 16, c.stream));
    k_count_sum<<<std::min((n + 255) / 256, 1024), 256, 0, c.stream>>>(n, (const int32_t *)d_cost, (const int32_t *)d_r2c,
                                                                       big_cost, (unsigned long long *)c.misc.p);
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, c.misc.p, 16, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    if (sum) *sum = ((int64_t *)c.pinned)[0];
    if (n_real) *n_real = (int32_t)((int64_t *)c.pinned)[1];
    return TD_OK;
}

}  // extern "C"
