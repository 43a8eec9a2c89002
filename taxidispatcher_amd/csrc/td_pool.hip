// td_pool.hip — pools of up to 4 passengers on gfx950.
//
// Replaces the reference's only native code: pool_n.c:101-207 (findPool / drop_customers /
// removeDuplicates; the same enumeration as Pool.java:32-113) and the merge of findpool.c:73-98.
//
//   plan      ordered pick-up sequence p[0..k) of distinct requests + a drop-off order q (a
//             permutation of 0..k-1 into p)
//   wait      at level l >= 1 the pick-up path p[0] -> .. -> p[l] may not be longer than WAIT(p[l])
//             (pool_n.c:174-179; a failing candidate is skipped)
//   happy     every passenger's ride (rest of the pick-up path from its own pick-up, last pick-up ->
//             first drop-off, drop-off path up to its own drop-off) <= direct * (1 + LOSS / 100.0),
//             compared in double like pool_n.c:118-119
//   cost      whole pick-up path + last pick-up -> first drop-off + drop-off path (pool_n.c:129-137)
//   output    happy plans stably sorted by cost (enumeration order among equal costs: glibc's qsort
//             is a merge sort at these sizes), a plan kept iff it shares no request with an earlier
//             kept plan (pool_n.c:195-215)
//
// GPU mapping.  Enumeration: one thread per (p0, p1) pair of the first-pick-up slice — the slice is
// findpool.c's fan-out unit (child t gets first pick-ups [t*step, (t+1)*step), pool_n.c:243-246), so
// the 8 children map to 8 launches / 8 GPUs — with the request table staged in LDS; the thread
// walks p2 / p3 under the wait rule and tests the k! drop-off orders.  A happy plan becomes ONE
// 64-bit key  cost << 50 | sequence number in the reference's enumeration order : the key alone
// encodes the plan, and an ascending radix sort of the keys (hipCUB) IS the reference's stable sort
// by cost.  The greedy de-duplication runs in one workgroup over chunks of 1024 sorted plans: a
// live plan (all its requests unused) is taken iff no EARLIER live plan of the chunk shares a
// request with it (atomicMin of the plan index per request in LDS), repeated until the chunk has
// no live plan — exactly the plans the sequential scan keeps, in the same order.
#include <hipcub/hipcub.hpp>

#include "td_common.h"

using namespace td;

namespace {

constexpr int PN_MAXN = 2047;          // 4 * 11 bits + 5 bits of drop-off order < 2^50
constexpr int PN_SEQ_BITS = 50;
constexpr unsigned long long PN_SEQ_MASK = (1ull << PN_SEQ_BITS) - 1ull;
constexpr int PN_MAXCOST = (1 << 14) - 1;

// the k! drop-off orders in lexicographic order (the recursion order of pool_n.c:140-153), 2 bits per slot
__host__ __device__ inline int pn_perm(int k, int qi, int *q)
{
    // generate the qi-th permutation of 0..k-1 in lexicographic order
    int avail[4] = {0, 1, 2, 3};
    int f = 1;
    for (int i = 2; i < k; i++) f *= i;   // (k-1)!
    int left = k;
    for (int i = 0; i < k; i++) {
        const int idx = qi / f;
        qi -= idx * f;
        q[i] = avail[idx];
        for (int j = idx; j < left - 1; j++) avail[j] = avail[j + 1];
        left--;
        if (left > 1) f /= left;
    }
    return 0;
}

__device__ __forceinline__ int pn_d(const int32_t *dist, int S, int a, int b)
{
    return dist ? dist[(int64_t)a * S + b] : (a > b ? a - b : b - a);
}

struct PnCtl {
    unsigned long long happy;   // happy plans found (may exceed the capacity: then the call fails)
    int err;                    // 1: cost does not fit the key
    int kept;
};

// happiness of all passengers + plan cost for pick-ups p and drop-off order q
template <int K>
__device__ __forceinline__ bool pn_check(const int *p, const int *q, const int32_t *sf, const int32_t *st,
                                         const int32_t *sl, const int32_t *dist, int S, int *cost_out)
{
    int pick[K];   // pick[i] = d(from p[i], from p[i+1]); suffix sums give "rest of the pick-up path"
    int ptot = 0;
#pragma unroll
    for (int i = 0; i < K - 1; i++) {
        pick[i] = pn_d(dist, S, sf[p[i]], sf[p[i + 1]]);
        ptot += pick[i];
    }
    const int first = pn_d(dist, S, sf[p[K - 1]], st[p[q[0]]]);
    int drop = 0;
    bool happy = true;
#pragma unroll
    for (int d = 0; d < K; d++) {
        if (d > 0) drop += pn_d(dist, S, st[p[q[d - 1]]], st[p[q[d]]]);
        int c = first + drop;
#pragma unroll
        for (int ph = 0; ph < K - 1; ph++) c += (ph >= q[d]) ? pick[ph] : 0;
        const int x = p[q[d]];
        const double lim = (double)pn_d(dist, S, sf[x], st[x]) * (1 + sl[x] / 100.0);
        if ((double)c > lim) happy = false;
    }
    *cost_out = ptot + first + drop;
    return happy;
}

template <int K>
__global__ __launch_bounds__(256) void k_pooln_enum(int n, const int32_t *__restrict__ from, const int32_t *__restrict__ to,
                                                    const int32_t *__restrict__ wait, const int32_t *__restrict__ loss,
                                                    const int32_t *__restrict__ dist, int S, int first0, int first1,
                                                    unsigned long long cap, unsigned long long *__restrict__ keys,
                                                    PnCtl *__restrict__ ctl)
{
    extern __shared__ int32_t sm[];
    int32_t *sf = sm, *st = sm + n, *sw = sm + 2 * n, *sl = sm + 3 * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        sf[i] = from[i];
        st[i] = to[i];
        sw[i] = wait[i];
        sl[i] = loss[i];
    }
    __syncthreads();
    constexpr int NP = K == 1 ? 1 : (K == 2 ? 2 : (K == 3 ? 6 : 24));
    int p[4] = {0, 0, 0, 0};
    p[0] = first0 + blockIdx.y;
    if (p[0] >= first1 || 0 > sw[p[0]]) return;   // (pool_n.c:177 at level 0: an empty path against WAIT)
    const unsigned long long nn = (unsigned long long)n;
    auto emit = [&](int p2, int p3) {
        p[2] = p2;
        p[3] = p3;
        for (int qi = 0; qi < NP; qi++) {
            int q[4];
            pn_perm(K, qi, q);
            int cost;
            if (pn_check<K>(p, q, sf, st, sl, dist, S, &cost)) {
                const unsigned long long idx = atomicAdd(&ctl->happy, 1ull);
                if (cost > PN_MAXCOST || cost < 0) atomicOr(&ctl->err, 1);
                if (idx < cap) {
                    const unsigned long long seq = ((((unsigned long long)p[0] * nn + (K > 1 ? p[1] : 0)) * nn + (K > 2 ? p2 : 0)) * nn +
                                                    (K > 3 ? p3 : 0)) * 24ull + (unsigned long long)qi;
                    keys[idx] = ((unsigned long long)cost << PN_SEQ_BITS) | seq;
                }
            }
        }
    };
    if constexpr (K == 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) emit(0, 0);
        return;
    } else {
        const int p1 = blockIdx.x * blockDim.x + threadIdx.x;
        if (p1 >= n || p1 == p[0]) return;
        p[1] = p1;
        const int d01 = pn_d(dist, S, sf[p[0]], sf[p1]);
        if (d01 > sw[p1]) return;           // pool_n.c:177-178
        if constexpr (K == 2) {
            emit(0, 0);
        } else {
            for (int p2 = 0; p2 < n; p2++) {
                if (p2 == p[0] || p2 == p1) continue;
                const int d012 = d01 + pn_d(dist, S, sf[p1], sf[p2]);
                if (d012 > sw[p2]) continue;
                if constexpr (K == 3) {
                    emit(p2, 0);
                } else {
                    for (int p3 = 0; p3 < n; p3++) {
                        if (p3 == p[0] || p3 == p1 || p3 == p2) continue;
                        if (d012 + pn_d(dist, S, sf[p2], sf[p3]) > sw[p3]) continue;
                        emit(p2, p3);
                    }
                }
            }
        }
    }
}

// Pools of FOUR, balanced: one thread per (first, second, third pick-up) — the slice x n x n index space as the grid —
// walks the fourth pick-up.  With one thread per (first, second) pair the two inner loops are n^2 long for the few
// pairs that pass the wait rule and empty for the rest: at n = 600 one child took 56 ms with most lanes idle.  Here a
// whole workgroup leaves at once when (first, second) fails (it is uniform), the survivors' loops are n long, and the
// happy plans of a wave are appended with ONE atomic per wave and drop-off order instead of one per plan.
__global__ __launch_bounds__(256) void k_pooln_enum4(int n, const int32_t *__restrict__ from, const int32_t *__restrict__ to,
                                                     const int32_t *__restrict__ wait, const int32_t *__restrict__ loss,
                                                     const int32_t *__restrict__ dist, int S, int first0, int first1,
                                                     unsigned long long cap, unsigned long long *__restrict__ keys,
                                                     PnCtl *__restrict__ ctl)
{
    int p[4];
    p[0] = first0 + blockIdx.z;
    p[1] = blockIdx.y;
    if (p[0] >= first1 || p[1] == p[0] || 0 > wait[p[0]]) return;
    const int d01 = pn_d(dist, S, from[p[0]], from[p[1]]);
    if (d01 > wait[p[1]]) return;   // pool_n.c:177-178, uniform over the workgroup
    const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (p2 >= n || p2 == p[0] || p2 == p[1]) return;
    const int d012 = d01 + pn_d(dist, S, from[p[1]], from[p2]);
    if (d012 > wait[p2]) return;
    p[2] = p2;
    const unsigned long long nn = (unsigned long long)n;
    const int lane = threadIdx.x & 63;
    const int f2 = from[p2];
    for (int p3 = 0; p3 < n; p3++) {
        if (p3 == p[0] || p3 == p[1] || p3 == p2) continue;
        if (d012 + pn_d(dist, S, f2, from[p3]) > wait[p3]) continue;
        p[3] = p3;
        for (int qi = 0; qi < 24; qi++) {
            int q[4];
            pn_perm(4, qi, q);
            int cost = 0;
            const bool happy = pn_check<4>(p, q, from, to, loss, dist, S, &cost);
            const unsigned long long m = __ballot(happy);   // the lanes that are here AND happy
            if (!m) continue;
            unsigned long long base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&ctl->happy, (unsigned long long)__popcll(m));
            base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), leader) << 32) |
                   (unsigned long long)(uint32_t)__shfl((int)(uint32_t)base, leader);
            if (happy) {
                if (cost > PN_MAXCOST || cost < 0) atomicOr(&ctl->err, 1);
                const unsigned long long idx = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                if (idx < cap) {
                    const unsigned long long seq = ((((unsigned long long)p[0] * nn + p[1]) * nn + p2) * nn + p3) * 24ull + (unsigned long long)qi;
                    keys[idx] = ((unsigned long long)cost << PN_SEQ_BITS) | seq;
                }
            }
        }
    }
}

// decode the requests of the plan behind sorted key number i
template <bool FROM_KEY>
__device__ __forceinline__ void pn_requests(int k, int n, unsigned long long key, const int32_t *recs, int *c)
{
    if (FROM_KEY) {
        unsigned long long s = (key & PN_SEQ_MASK) / 24ull;
        for (int i = 3; i >= 0; i--) {
            c[i] = (int)(s % (unsigned long long)n);
            s /= (unsigned long long)n;
        }
    } else {
        const int32_t *r = recs + (size_t)(key & 0xFFFFFFFFull) * (size_t)(2 * k + 1);
        for (int i = 0; i < k; i++) c[i] = r[i];
    }
}

// Greedy de-duplication over the sorted plans, ONE workgroup.  kept[] receives the sorted positions
// of the kept plans in order.
template <bool FROM_KEY>
__global__ __launch_bounds__(1024) void k_pooln_greedy(int k, int n, unsigned long long count,
                                                       const unsigned long long *__restrict__ keys,
                                                       const int32_t *__restrict__ recs, int max_kept,
                                                       unsigned long long *__restrict__ kept, PnCtl *__restrict__ ctl)
{
    extern __shared__ int sm[];
    int *owner = sm;                                         // n: smallest live plan of the chunk claiming the request
    unsigned int *used = reinterpret_cast<unsigned int *>(sm + n);   // (n + 31) / 32
    __shared__ int s_live, s_nkept, s_nused, s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < n; i += 1024) owner[i] = INT_MAX;
    for (int i = tid; i < (n + 31) / 32; i += 1024) used[i] = 0u;
    if (tid == 0) {
        s_nkept = 0;
        s_nused = 0;
    }
    __syncthreads();
    for (unsigned long long base = 0; base < count; base += 1024ull) {
        if (s_nused + k > n || s_nkept >= max_kept) break;   // no k unused requests left / output full
        const unsigned long long i = base + (unsigned long long)tid;
        int c[4] = {0, 0, 0, 0};
        bool alive = false, taken = false;
        if (i < count) {
            pn_requests<FROM_KEY>(k, n, keys[i], recs, c);
            alive = true;
        }
        for (;;) {
            // a plan dies when one of its requests is used
            if (alive && !taken) {
                for (int a = 0; a < k; a++)
                    if ((used[c[a] >> 5] >> (c[a] & 31)) & 1u) alive = false;
            }
            const bool live = alive && !taken;
            if (tid == 0) s_live = 0;
            __syncthreads();
            if (live) {
                s_live = 1;
                for (int a = 0; a < k; a++) atomicMin(&owner[c[a]], tid);
            }
            __syncthreads();
            if (!s_live) break;
            bool win = live;
            if (live)
                for (int a = 0; a < k; a++) win = win && (owner[c[a]] == tid);
            __syncthreads();
            if (live)
                for (int a = 0; a < k; a++) owner[c[a]] = INT_MAX;
            if (win) {
                taken = true;
                for (int a = 0; a < k; a++) atomicOr(&used[c[a] >> 5], 1u << (c[a] & 31));
            }
            __syncthreads();
        }
        // ordered compaction of the chunk's taken plans
        const unsigned long long m = __ballot(taken);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[w] = __popcll(m);
        __syncthreads();
        int off = s_nkept, tot = 0;
        for (int q = 0; q < 16; q++) {
            if (q < w) off += s_w[q];
            tot += s_w[q];
        }
        if (taken && off + before < max_kept) kept[off + before] = i;
        __syncthreads();
        if (tid == 0) {
            s_nkept += tot;
            s_nused += tot * k;
        }
        __syncthreads();
    }
    if (tid == 0) ctl->kept = min(s_nkept, max_kept);
}

// kept sorted positions -> records (pick-ups, drop-offs, cost)
template <bool FROM_KEY>
__global__ void k_pooln_emit(int k, int n, int nkept, const unsigned long long *__restrict__ kept,
                             const unsigned long long *__restrict__ keys, const int32_t *__restrict__ recs,
                             int32_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nkept) return;
    const unsigned long long key = keys[kept[t]];
    int32_t *o = out + (size_t)t * (size_t)(2 * k + 1);
    if (FROM_KEY) {
        int c[4];
        pn_requests<true>(k, n, key, nullptr, c);
        const int qi = (int)((key & PN_SEQ_MASK) % 24ull);
        int q[4];
        pn_perm(k, qi, q);
        for (int i = 0; i < k; i++) {
            o[i] = c[i];
            o[k + i] = c[q[i]];
        }
        o[2 * k] = (int32_t)(key >> PN_SEQ_BITS);
    } else {
        const int32_t *r = recs + (size_t)(key & 0xFFFFFFFFull) * (size_t)(2 * k + 1);
        for (int i = 0; i < 2 * k + 1; i++) o[i] = r[i];
    }
}

// merge keys: (cost when sorting by cost) << 32 | input position
__global__ void k_pool_merge_keys(int k, int n_in, const int32_t *__restrict__ recs, int sort_by_cost,
                                  unsigned long long *__restrict__ keys)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_in) return;
    const unsigned long long c = sort_by_cost ? (unsigned long long)(uint32_t)recs[(size_t)t * (2 * k + 1) + 2 * k] : 0ull;
    keys[t] = (c << 32) | (unsigned long long)t;
}

struct PoolWs {
    Buf keys, keys2, kept, tmp, ctl, out, in;
};
PoolWs g_pw;

int sort_keys(unsigned long long *in, unsigned long long *out, size_t count)
{
    Ctx &c = ctx();
    size_t bytes = 0;
    TD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, in, out, (int)count, 0, 64, c.stream));
    int rc = ensure(g_pw.tmp, bytes + 256);
    if (rc) return rc;
    TD_HIP(hipcub::DeviceRadixSort::SortKeys(g_pw.tmp.p, bytes, in, out, (int)count, 0, 64, c.stream));
    return TD_OK;
}

template <bool FROM_KEY>
int greedy_and_emit(int k, int n, size_t count, const unsigned long long *sorted, const int32_t *recs, int max_pools,
                    int32_t *pools, int32_t *n_pools)
{
    Ctx &c = ctx();
    int rc;
    if ((rc = ensure(g_pw.kept, sizeof(unsigned long long) * (size_t)std::max(max_pools, 1)))) return rc;
    PnCtl *ctl = (PnCtl *)g_pw.ctl.p;
    const size_t shm = sizeof(int) * (size_t)n + sizeof(unsigned int) * (size_t)((n + 31) / 32);
    k_pooln_greedy<FROM_KEY><<<1, 1024, shm, c.stream>>>(k, n, (unsigned long long)count, sorted, recs, max_pools,
                                                        (unsigned long long *)g_pw.kept.p, ctl);
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, ctl, sizeof(PnCtl), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    const int nk = ((const PnCtl *)c.pinned)->kept;
    if (nk > 0) {
        const size_t ob = sizeof(int32_t) * (size_t)nk * (size_t)(2 * k + 1);
        int32_t *d_out = pools;
        if (!is_device_ptr(pools)) {
            if ((rc = ensure(g_pw.out, ob))) return rc;
            d_out = (int32_t *)g_pw.out.p;
        }
        k_pooln_emit<FROM_KEY><<<(nk + 255) / 256, 256, 0, c.stream>>>(k, n, nk, (const unsigned long long *)g_pw.kept.p, sorted, recs, d_out);
        TD_HIP(hipGetLastError());
        if (d_out != pools) TD_HIP(hipMemcpyAsync(pools, d_out, ob, hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    *n_pools = nk;
    return TD_OK;
}

}  // namespace

extern "C" int td_pool_n(int k, int n, const int32_t *from, const int32_t *to, const int32_t *max_wait,
                         const int32_t *max_loss, const int32_t *dist, int S, int first0, int first1,
                         int64_t max_happy, int max_pools, int32_t *pools, int32_t *n_pools, int64_t *n_happy)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    // k = 1 is refused on purpose: pool_n.c / findpool.c compare MAX_IN_POOL = 4 slots when they drop duplicates, so with
    // one passenger the padding makes every later plan a duplicate of the first — not a result worth reproducing
    if (k < 2 || k > 4) return fail(TD_EINVAL, "pool size %d (2..4)", k);
    if (n < 0 || n > PN_MAXN) return fail(TD_EINVAL, "n=%d (at most %d requests; the reference's MAX_DEMAND is 2000)", n, PN_MAXN);
    if (!n_pools || (max_pools > 0 && !pools)) return fail(TD_EINVAL, "null output");
    if (dist && S <= 0) return fail(TD_EINVAL, "dist given but S=%d", S);
    *n_pools = 0;
    if (n_happy) *n_happy = 0;
    first0 = std::max(first0, 0);
    first1 = std::min(first1, n);
    if (n < k || first1 <= first0 || max_pools <= 0) return TD_OK;
    if (!from || !to || !max_wait || !max_loss) return fail(TD_EINVAL, "null request array");
    if (max_happy <= 0) max_happy = 1 << 22;
    int rc;
    // stage the four request arrays (and the table)
    if ((rc = ensure(g_pw.in, sizeof(int32_t) * 4 * (size_t)n))) return rc;
    const int32_t *src[4] = {from, to, max_wait, max_loss};
    const int32_t *d_arr[4];
    for (int q = 0; q < 4; q++) {
        if (is_device_ptr(src[q])) d_arr[q] = src[q];
        else {
            int32_t *dst = (int32_t *)g_pw.in.p + (size_t)q * n;
            TD_HIP(hipMemcpyAsync(dst, src[q], sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c.stream));
            d_arr[q] = dst;
        }
    }
    const void *d_dist = nullptr;
    if (dist && (rc = to_device(dist, sizeof(int32_t) * (size_t)S * S, c.stage_c, &d_dist))) return rc;
    if ((rc = ensure(g_pw.keys, sizeof(unsigned long long) * (size_t)max_happy))) return rc;
    if ((rc = ensure(g_pw.keys2, sizeof(unsigned long long) * (size_t)max_happy))) return rc;
    if ((rc = ensure(g_pw.ctl, 256))) return rc;
    PnCtl *ctl = (PnCtl *)g_pw.ctl.p;
    TD_HIP(hipMemsetAsync(ctl, 0, sizeof(PnCtl), c.stream));
    {
        ProfScope ps(TD_K_LCM);
        const size_t shm = sizeof(int32_t) * 4 * (size_t)n;
        dim3 g((n + 255) / 256, first1 - first0);
        unsigned long long *keys = (unsigned long long *)g_pw.keys.p;
#define TD_PN(KV)                                                                                                        \
    k_pooln_enum<KV><<<g, 256, shm, c.stream>>>(n, d_arr[0], d_arr[1], d_arr[2], d_arr[3], (const int32_t *)d_dist, S, \
                                               first0, first1, (unsigned long long)max_happy, keys, ctl)
        switch (k) {
            case 1: TD_PN(1); break;
            case 2: TD_PN(2); break;
            case 3: TD_PN(3); break;
            default:
                if (n <= 65535 && first1 - first0 <= 65535)
                    k_pooln_enum4<<<dim3((n + 255) / 256, n, first1 - first0), 256, 0, c.stream>>>(
                        n, d_arr[0], d_arr[1], d_arr[2], d_arr[3], (const int32_t *)d_dist, S, first0, first1, (unsigned long long)max_happy, keys, ctl);
                else
                    TD_PN(4);
                break;
        }
#undef TD_PN
        TD_HIP(hipGetLastError());
    }
    TD_HIP(hipMemcpyAsync(c.pinned, ctl, sizeof(PnCtl), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    const PnCtl h = *(const PnCtl *)c.pinned;
    if (n_happy) *n_happy = (int64_t)h.happy;
    if (h.err) return fail(TD_ERANGE, "td_pool_n: a plan cost does not fit 14 bits");
    if (h.happy > (unsigned long long)max_happy)
        return fail(TD_ERANGE, "td_pool_n: %llu happy plans exceed max_happy=%lld", h.happy, (long long)max_happy);
    if (h.happy == 0) return TD_OK;
    if ((rc = sort_keys((unsigned long long *)g_pw.keys.p, (unsigned long long *)g_pw.keys2.p, (size_t)h.happy))) return rc;
    return greedy_and_emit<true>(k, n, (size_t)h.happy, (const unsigned long long *)g_pw.keys2.p, nullptr, max_pools, pools, n_pools);
}

// findpool.c:73-98,166-172: the children's lists concatenated in child order, sorted by cost when
// sort_by_cost (the reference sorts on the 9th field, which its reader only fills for 4-passenger
// pools: smaller pools keep the child order), then the same de-duplication.
extern "C" int td_pool_merge(int k, int n_requests, int n_in, const int32_t *pools_in, int sort_by_cost, int max_pools,
                             int32_t *pools_out, int32_t *n_out)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (k < 2 || k > 4 || n_requests < 0 || n_requests > PN_MAXN || n_in < 0) return fail(TD_EINVAL, "bad arguments to td_pool_merge (pool size 2..4)");
    if (!n_out || (n_in && !pools_in) || (max_pools > 0 && !pools_out)) return fail(TD_EINVAL, "null array");
    *n_out = 0;
    if (n_in == 0 || max_pools <= 0) return TD_OK;
    int rc;
    const void *d_in;
    if ((rc = to_device(pools_in, sizeof(int32_t) * (size_t)n_in * (size_t)(2 * k + 1), g_pw.in, &d_in))) return rc;
    if ((rc = ensure(g_pw.keys, sizeof(unsigned long long) * (size_t)n_in))) return rc;
    if ((rc = ensure(g_pw.keys2, sizeof(unsigned long long) * (size_t)n_in))) return rc;
    if ((rc = ensure(g_pw.ctl, 256))) return rc;
    TD_HIP(hipMemsetAsync(g_pw.ctl.p, 0, sizeof(PnCtl), c.stream));
    k_pool_merge_keys<<<(n_in + 255) / 256, 256, 0, c.stream>>>(k, n_in, (const int32_t *)d_in, sort_by_cost, (unsigned long long *)g_pw.keys.p);
    TD_HIP(hipGetLastError());
    if ((rc = sort_keys((unsigned long long *)g_pw.keys.p, (unsigned long long *)g_pw.keys2.p, (size_t)n_in))) return rc;
    return greedy_and_emit<false>(k, n_requests, (size_t)n_in, (const unsigned long long *)g_pw.keys2.p, (const int32_t *)d_in, max_pools,
                                  pools_out, n_out);
}
