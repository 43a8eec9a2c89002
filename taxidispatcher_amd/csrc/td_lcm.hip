// td_lcm.hip — LCM ("lowest cost method") greedy pre-reduce on gfx950.
//
// Reference: greedy_opt.py:61-82, simulate.py:76-98, heuristic.py:24-33, Simulator.java:523-549
// — k times { global argmin in row-major order over the whole n x n matrix; mask row+col }.
// The reference pays k full scans of n^2 cells.  Here the matrix is read ONCE:
//   k_lcm_rowscan  segmented argmin: one wavefront per row keeps the row's first minimum
//                  (value, col) as a packed 64-bit key (wave64 shuffle reduction);
//   k_lcm_loop     one persistent workgroup repeats { block-wide argmin over the n row keys
//                  (ties -> lowest row, then lowest col = the reference's row-major first
//                  minimum); record the pair; mask; re-scan ONLY the rows whose cached column
//                  was just taken } with the column mask as an LDS bitset.
// Pairs come out in exactly the order the reference takes them.
#include <limits.h>

#include "td_common.h"

using namespace td;

namespace {

constexpr unsigned long long LCM_INF = ~0ull;

__device__ __forceinline__ unsigned long long lcm_key(int32_t v, int col)
{
    return ((unsigned long long)((uint32_t)v ^ 0x80000000u) << 32) | (uint32_t)col;
}
__device__ __forceinline__ int32_t lcm_val(unsigned long long k) { return (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); }

// first minimum of one row among unmasked candidate columns, by one wavefront
__device__ __forceinline__ unsigned long long lcm_scan_row(const int32_t *__restrict__ rp, int n, int lane,
                                                           const uint32_t *colmask, int64_t cand_limit)
{
    unsigned long long best = LCM_INF;
    for (int j0 = lane; j0 < n; j0 += 256) {
        int32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (j0 + 64 * u < n) ? rp[j0 + 64 * u] : 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 64 * u;
            if (j < n) {
                const bool masked = colmask ? ((colmask[j >> 5] >> (j & 31)) & 1u) : false;
                if (!masked && (int64_t)v[u] < cand_limit) {
                    const unsigned long long k = lcm_key(v[u], j);
                    best = k < best ? k : best;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o);
        best = ob < best ? ob : best;
    }
    return best;
}

// Narrow copy for the re-scans: code = min(value - base, 255) as one byte per cell (255 also for
// non-candidates and padding), rows padded to 16 bytes.  A re-scan then streams 16 columns per
// lane per load instead of one; a row whose free columns all carry code 255 falls back to the
// exact int32 scan.  base = smallest row minimum (written by k_lcm_rowscan with atomicMin).
__global__ __launch_bounds__(256) void k_lcm_narrow(int n, int pitch, const int32_t *__restrict__ cost,
                                                    int64_t cand_limit, const int32_t *__restrict__ base_p,
                                                    uint8_t *__restrict__ codes)
{
    const int32_t base = *base_p;
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        const int32_t *rp = cost + (int64_t)row * n;
        uint8_t *dst = codes + (size_t)row * pitch;
        for (int j = threadIdx.x; j < pitch; j += blockDim.x) {
            uint32_t code = 255u;
            if (j < n) {
                const int32_t v = rp[j];
                const int64_t d = (int64_t)v - (int64_t)base;
                if ((int64_t)v < cand_limit && d >= 0 && d < 255) code = (uint32_t)d;
            }
            dst[j] = (uint8_t)code;
        }
    }
}

// first minimum code among unmasked columns of one narrow row, by one wavefront:
// returns code << 20 | col  (0xFFFFFFFF when nothing below 255 is left)
__device__ __forceinline__ uint32_t lcm_scan_narrow(const uint8_t *__restrict__ rp, int nchunks, int lane,
                                                    const uint32_t *colmask)
{
    uint32_t best = 0xFFFFFFFFu;
    for (int c0 = lane; c0 < nchunks; c0 += 128) {
        uint4 cv[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (c0 + 64 * u < nchunks) cv[u] = *reinterpret_cast<const uint4 *>(rp + (size_t)(c0 + 64 * u) * 16);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int ch = c0 + 64 * u;
            if (ch >= nchunks) continue;
            const uint32_t wmask = colmask[ch >> 1];
            const uint32_t m16 = (ch & 1) ? (wmask >> 16) : (wmask & 0xFFFFu);
            const uint32_t wv[4] = {cv[u].x, cv[u].y, cv[u].z, cv[u].w};
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t code = (wv[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                const uint32_t k = (code << 20) | (uint32_t)(ch * 16 + e);
                const bool ok = !((m16 >> e) & 1u) && code < 255u;
                best = (ok && k < best) ? k : best;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ob = __shfl_xor(best, o);
        best = ob < best ? ob : best;
    }
    return best;
}

__global__ __launch_bounds__(256) void k_lcm_rowscan(int n, const int32_t *__restrict__ cost, int64_t cand_limit,
                                                     unsigned long long *__restrict__ rowbest, int32_t *__restrict__ base)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int32_t mn = INT_MAX;
    for (int row = blockIdx.x * nw + w; row < n; row += gridDim.x * nw) {
        const unsigned long long b = lcm_scan_row(cost + (int64_t)row * n, n, lane, nullptr, cand_limit);
        if (lane == 0) {
            rowbest[row] = b;
            if (b != LCM_INF) mn = min(mn, lcm_val(b));
        }
    }
    if (lane == 0 && mn != INT_MAX) atomicMin(base, mn);
}

struct LcmOut {
    int32_t n_pairs;
    int32_t last_min;
    int64_t total;
};

__global__ __launch_bounds__(1024) void k_lcm_loop(int n, const int32_t *__restrict__ cost, int64_t cand_limit,
                                                   int32_t mask, int32_t threshold, int stop_value_on,
                                                   int32_t stop_value, int stop_size, int64_t sum_below, int max_pairs,
                                                   unsigned long long *__restrict__ rowbest, int rb_in_lds,
                                                   int32_t *__restrict__ rows,
                                                   int32_t *__restrict__ cols, int *__restrict__ rescan,
                                                   LcmOut *__restrict__ out, const uint8_t *__restrict__ codes,
                                                   int pitch, const int32_t *__restrict__ base_p, int symmetric)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    // dynamic LDS: [rowbest copy: n x 8 B when it fits] [column mask: (n+31)/32 words]
    unsigned long long *rb = rb_in_lds ? reinterpret_cast<unsigned long long *>(s_dyn) : rowbest;
    uint32_t *s_colmask = reinterpret_cast<uint32_t *>(s_dyn + (rb_in_lds ? (size_t)n * 8 : 0));
    __shared__ unsigned long long s_red[16];
    __shared__ int s_nres;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    for (int k = tid; k < (n + 31) / 32; k += T) s_colmask[k] = 0u;
    if (rb_in_lds)
        for (int i = tid; i < n; i += T) rb[i] = rowbest[i];
    if (tid == 0) s_nres = 0;
    __syncthreads();
    int npairs = 0, size = n;
    int32_t last_min = stop_value;
    int64_t total = 0;
    const int iters = n < max_pairs ? n : max_pairs;
    for (int it = 0; it < iters; it++) {
        // block argmin over (value, row); the column rides along in the row's cached key
        unsigned long long best = LCM_INF;
        for (int i = tid; i < n; i += T) {
            const unsigned long long k = rb[i];
            if (k != LCM_INF) {
                const unsigned long long kk = (k & 0xFFFFFFFF00000000ull) | (uint32_t)i;
                best = kk < best ? kk : best;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o);
            best = ob < best ? ob : best;
        }
        if (lane == 0) s_red[w] = best;
        __syncthreads();
        best = s_red[0];
        for (int k = 1; k < nw; k++) best = s_red[k] < best ? s_red[k] : best;
        if (best == LCM_INF) {  // nothing left to look at
            last_min = stop_value_on ? stop_value : mask;
            break;
        }
        const int r = (int)(uint32_t)best;
        const int32_t v = lcm_val(best);
        const int c = (int)(uint32_t)rb[r];
        last_min = v;
        if (threshold >= 0 && v > threshold) break;       // greedy_opt.py:68-69
        if (stop_value_on && v >= stop_value) break;       // Simulator.java:538
        if (v >= mask) break;                              // only masked-valued cells remain
        if (tid == 0) {
            rows[npairs] = r;
            cols[npairs] = c;
        }
        npairs++;
        if ((int64_t)v < sum_below) total += v;
        size--;
        __syncthreads();  // everyone has read rb[r] / s_red
        if (tid == 0) {
            rb[r] = LCM_INF;
            s_colmask[c >> 5] |= 1u << (c & 31);
            if (symmetric) {  // pool of two: both customers leave the game in both roles
                rb[c] = LCM_INF;
                s_colmask[r >> 5] |= 1u << (r & 31);
            }
            s_nres = 0;
        }
        __syncthreads();
        if (stop_size >= 0 && size == stop_size) break;    // Simulator.java:544-545
        // rows whose cached first minimum sat in a column that was just taken must be re-scanned
        for (int i = tid; i < n; i += T) {
            const unsigned long long k = rb[i];
            if (k != LCM_INF && ((int)(uint32_t)k == c || (symmetric && (int)(uint32_t)k == r)))
                rescan[atomicAdd(&s_nres, 1)] = i;
        }
        __syncthreads();
        const int nres = s_nres;
        for (int q = w; q < nres; q += nw) {
            const int i = rescan[q];
            unsigned long long b;
            const uint32_t nk = codes ? lcm_scan_narrow(codes + (size_t)i * pitch, pitch >> 4, lane, s_colmask) : 0xFFFFFFFFu;
            if (nk != 0xFFFFFFFFu)
                b = lcm_key(*base_p + (int32_t)(nk >> 20), (int)(nk & 0xFFFFFu));   // exact: code < 255
            else
                b = lcm_scan_row(cost + (int64_t)i * n, n, lane, s_colmask, cand_limit);
            if (lane == 0) rb[i] = b;
        }
        __syncthreads();
    }
    if (tid == 0) {
        out->n_pairs = npairs;
        out->last_min = last_min;
        out->total = total;
    }
}

// =====================================================================================
// Level-list LCM (the fast path of td_lcm).
//
// When the candidate values span < LV_MAX levels — true for every reference variant: distances
// up to the threshold (greedy_opt.py: 0..10, simulate.py: 0..20), stand distances below
// DROP_TIME (Simulator.java: 0..9), heuristic.py costs 1..39 — the greedy is done on explicit
// cell lists instead of row re-scans:
//   k_lcms_minmax   one streaming read: min / max / count of the candidate cells
//   k_lcms_hist     one read: per row and level the number of candidates
//   k_lcms_scan     exclusive scan (level-major, then row) -> list offsets
//   k_lcms_scatter  one read: cells written as (row<<16 | col) into their level list, row-major
//                   inside a level  ==> the lists, concatenated, are the matrix sorted by the
//                   reference's key (value, row, col)
//   k_lcms_greedy   one workgroup walks the lists 1024 cells at a time: a live cell (row and
//                   column still free) is taken iff no EARLIER live cell of the chunk shares its
//                   row or column (LDS atomicMin per row / column), repeated until the chunk has
//                   no live cell; survivors are exactly the cells the sequential greedy takes,
//                   and they are emitted in list order.
// =====================================================================================
constexpr int LV_MAX = 256;   // (pool-of-two costs are sums of three distances: up to ~150 levels on 50 stands)
static int g_lcm_lists = getenv("TD_LCM_LISTS") ? atoi(getenv("TD_LCM_LISTS")) : 1;   // 0: always the row-scan loop

struct LcmsInfo {
    long long count;   // candidate cells
    int vmin, vmax;
    int pad[2];
};

__global__ __launch_bounds__(256) void k_lcms_minmax(int n, const int32_t *__restrict__ cost, int64_t hi,
                                                     LcmsInfo *__restrict__ info)
{
    int mn = INT_MAX, mx = INT_MIN;
    long long cnt = 0;
    const int64_t total = (int64_t)n * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = cost[i];
        if ((int64_t)v <= hi) {
            mn = min(mn, v);
            mx = max(mx, v);
            cnt++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o));
        mx = max(mx, __shfl_xor(mx, o));
        cnt += __shfl_xor(cnt, o);
    }
    // one partial per workgroup, reduced by the host with the read-back it does anyway (hundreds of same-address
    // atomics cost ~30 ns each: more than the scan of a simulator-sized matrix)
    __shared__ int s_mn[4], s_mx[4];
    __shared__ long long s_cnt[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mn[w] = mn;
        s_mx[w] = mx;
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            mn = min(mn, s_mn[k]);
            mx = max(mx, s_mx[k]);
            cnt += s_cnt[k];
        }
        LcmsInfo part;
        part.count = cnt;
        part.vmin = mn;
        part.vmax = mx;
        part.pad[0] = part.pad[1] = 0;
        info[blockIdx.x] = part;
    }
}

// one wavefront per row; SCATTER = false: count per level, true: write the cells
constexpr int LCH = 16;
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_lcms_rows(int n, const int32_t *__restrict__ cost, int64_t hi, int vmin, int nlev,
                                                   int *__restrict__ rowcnt /* [nlev][n] counts, then offsets */,
                                                   uint32_t *__restrict__ cells)
{
    __shared__ int s_cnt[4][LV_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = blockIdx.x * nw + w; row < n; row += gridDim.x * nw) {
        for (int l = lane; l < nlev; l += 64) s_cnt[w][l] = SCATTER ? rowcnt[(size_t)l * n + row] : 0;
        const int32_t *rp = cost + (int64_t)row * n;
        // 16 chunks of 64 cells are loaded together (one memory round trip per 1024 cells instead of
        // one per 64: the scan of a row is a dependent chain, a 1300-cell row took 21 round trips)
        for (int jb = 0; jb < n; jb += 64 * LCH) {
          int vv[LCH];
#pragma unroll
          for (int u = 0; u < LCH; u++) {
              const int j = jb + 64 * u + lane;
              vv[u] = (j < n) ? rp[j] : 0;
          }
#pragma unroll
          for (int u = 0; u < LCH; u++) {
            const int j0 = jb + 64 * u;
            if (j0 >= n) break;
            const int j = j0 + lane;
            const int v = vv[u];
            int lv = (j < n && (int64_t)v <= hi) ? v - vmin : -1;
            unsigned long long act = __ballot(lv >= 0);
            if (nlev <= 16) {
                // few levels (a simulator model has 10): one ballot per level instead of a wave-wide minimum per level present
                for (int l = 0; l < nlev && act; l++) {
                    const unsigned long long m = __ballot(lv == l);
                    if (!m) continue;
                    if (SCATTER) {
                        const int base = s_cnt[w][l];
                        if (lv == l) cells[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)row << 16) | (uint32_t)j;
                    }
                    if (lane == 0) s_cnt[w][l] += __popcll(m);
                    act &= ~m;
                }
                act = 0;
            }
            while (act) {
                // lowest level present among the still-active lanes (wave-uniform)
                int cur = lv >= 0 ? lv : INT_MAX;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cur = min(cur, __shfl_xor(cur, o));
                const unsigned long long m = __ballot(lv == cur);
                if (SCATTER) {
                    const int base = s_cnt[w][cur];
                    if (lv == cur) cells[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)row << 16) | (uint32_t)j;
                }
                if (lane == 0) s_cnt[w][cur] += __popcll(m);
                if (lv == cur) lv = -1;
                act &= ~m;
            }
          }
        }
        if (!SCATTER)
            for (int l = lane; l < nlev; l += 64) rowcnt[(size_t)l * n + row] = s_cnt[w][l];
    }
}

// The same for n <= 4096 with ONE WORKGROUP per row: wave w takes the w-th quarter of the columns (all of it loaded at
// once, <= 16 chunks per lane), so the dependent ballot chain of a 1300-cell row is 6 chunks instead of 21 and a tick's
// two passes take a third of the time.  Counts per wave and level go through LDS; a row's cells still land in column
// order (wave-major, then chunk, then lane).  `bad` is raised when a candidate cell lies outside [vmin, vmin + nlev):
// possible only when the caller HINTED the value range (td_tick knows it from the threshold) — the host then re-runs
// with the measured range.
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_lcms_rows4(int n, const int32_t *__restrict__ cost, int64_t hi, int vmin, int nlev,
                                                    int *__restrict__ rowcnt, uint32_t *__restrict__ cells, int *__restrict__ bad, int bad_tag)
{
    __shared__ int s_cnt[4][LV_MAX];
    __shared__ int s_base[4][LV_MAX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x;
    const int q = (((n + 3) / 4) + 63) / 64 * 64;   // columns per wave (a multiple of 64, <= 1024)
    const int jlo = w * q;
    for (int l = lane; l < nlev; l += 64) s_cnt[w][l] = 0;
    const int32_t *rp = cost + (int64_t)row * n;
    int lv[LCH];
    bool oob = false;
#pragma unroll
    for (int u = 0; u < LCH; u++) {
        const int j = jlo + 64 * u + lane;
        const bool in = 64 * u < q && j < n;
        const int v = in ? rp[j] : 0;
        int l = (in && (int64_t)v <= hi) ? v - vmin : -1;
        if (in && (int64_t)v <= hi && (l < 0 || l >= nlev)) {
            oob = true;
            l = -1;
        }
        lv[u] = l;
    }
    if (__any(oob) && lane == 0) atomicMax(bad, bad_tag);   // tagged with the call's number: the flag never needs clearing
    // pass 1: counts of this wave per level
    if (nlev <= 16) {
#pragma unroll
        for (int u = 0; u < LCH; u++) {
            if (64 * u >= q) break;
            for (int l = 0; l < nlev; l++) {
                const unsigned long long m = __ballot(lv[u] == l);
                if (m && lane == 0) s_cnt[w][l] += __popcll(m);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < LCH; u++) {
            if (64 * u >= q) break;
            int cur_lv = lv[u];
            unsigned long long act = __ballot(cur_lv >= 0);
            while (act) {
                int cur = cur_lv >= 0 ? cur_lv : INT_MAX;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cur = min(cur, __shfl_xor(cur, o));
                const unsigned long long m = __ballot(cur_lv == cur);
                if (lane == 0) s_cnt[w][cur] += __popcll(m);
                if (cur_lv == cur) cur_lv = -1;
                act &= ~m;
            }
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (int l = tid; l < nlev; l += 256) rowcnt[(size_t)l * n + row] = s_cnt[0][l] + s_cnt[1][l] + s_cnt[2][l] + s_cnt[3][l];
        return;
    }
    for (int l = tid; l < nlev; l += 256) {
        int b = rowcnt[(size_t)l * n + row];   // offset of (level, row) after the scan
        for (int ww = 0; ww < 4; ww++) {
            s_base[ww][l] = b;
            b += s_cnt[ww][l];
        }
    }
    __syncthreads();
    // pass 2: scatter in column order
    if (nlev <= 16) {
#pragma unroll
        for (int u = 0; u < LCH; u++) {
            if (64 * u >= q) break;
            const int j = jlo + 64 * u + lane;
            for (int l = 0; l < nlev; l++) {
                const unsigned long long m = __ballot(lv[u] == l);
                if (!m) continue;
                const int base = s_base[w][l];
                if (lv[u] == l) cells[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)row << 16) | (uint32_t)j;
                if (lane == 0) s_base[w][l] = base + __popcll(m);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < LCH; u++) {
            if (64 * u >= q) break;
            const int j = jlo + 64 * u + lane;
            int cur_lv = lv[u];
            unsigned long long act = __ballot(cur_lv >= 0);
            while (act) {
                int cur = cur_lv >= 0 ? cur_lv : INT_MAX;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cur = min(cur, __shfl_xor(cur, o));
                const unsigned long long m = __ballot(cur_lv == cur);
                const int base = s_base[w][cur];
                if (cur_lv == cur) cells[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)row << 16) | (uint32_t)j;
                if (lane == 0) s_base[w][cur] = base + __popcll(m);
                if (cur_lv == cur) cur_lv = -1;
                act &= ~m;
            }
        }
    }
}

// exclusive scan of rowcnt[nlev*n] in place (one workgroup), level starts to lvstart[nlev+1]; CH tiles of 1024 are
// loaded before the serial chain of tile scans starts, so the chain does not wait on memory
__global__ __launch_bounds__(1024) void k_lcms_scan(int total, int n, int nlev, int *__restrict__ a, int *__restrict__ lvstart)
{
    constexpr int CH = 8;
    __shared__ int s_w[2][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int carry = 0, par = 0;
    for (int base = 0; base < total; base += 1024 * CH) {
        int v[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int i = base + u * 1024 + tid;
            v[u] = i < total ? a[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            if (base + u * 1024 >= total) break;
            int incl = v[u];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_w[par][w] = incl;
            __syncthreads();
            int wb = 0, all = 0;
            for (int k = 0; k < 16; k++) {
                const int x = s_w[par][k];
                wb += (k < w) ? x : 0;
                all += x;
            }
            par ^= 1;
            const int i = base + u * 1024 + tid;
            const int excl = carry + wb + incl - v[u];
            if (i < total) {
                a[i] = excl;
                if (i % n == 0) lvstart[i / n] = excl;
            }
            carry += all;
        }
    }
    if (tid == 0) lvstart[nlev] = carry;
}

// SYM (pool of two, Simulator.java:729-739): a kept plan (A, B) takes BOTH customers out of both roles, i.e. one
// "taken" set and one conflict table for rows and columns together.
template <bool SYM>
__global__ __launch_bounds__(1024) void k_lcms_greedy(int n, int hmask, int nlev, int vmin,
                                                      const uint32_t *__restrict__ cells,
                                                      const int *__restrict__ lvstart, int limit, int64_t sum_below,
                                                      int32_t stop_value, int32_t *__restrict__ rows,
                                                      int32_t *__restrict__ cols, LcmOut *__restrict__ out,
                                                      int *__restrict__ exhausted, uint32_t *__restrict__ g_taken)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    // [row table: hmask+1 ints][column table: hmask+1 ints][rows taken: bits][columns taken: bits]
    // The tables are indexed by (row & hmask) / (col & hmask): when n exceeds the table two rows can
    // share a slot, which only delays the later cell to the next pass (the test stays sufficient).
    int *rmin = reinterpret_cast<int *>(s_dyn);
    int *cmin = SYM ? rmin : rmin + (hmask + 1);
    const int nw32 = (n + 31) / 32;
    uint32_t *rtk = reinterpret_cast<uint32_t *>(rmin + 2 * (hmask + 1));
    uint32_t *ctk = SYM ? rtk : rtk + nw32;
    __shared__ int s_w[16];
    __shared__ int s_any;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i <= hmask; i += 1024) {
        rmin[i] = INT_MAX;
        cmin[i] = INT_MAX;
    }
    for (int i = tid; i < nw32; i += 1024) {
        rtk[i] = 0u;
        ctk[i] = 0u;
    }
    __syncthreads();
    int npairs = 0;
    int64_t total = 0;
    int32_t last_min = stop_value;
    bool done = (limit <= 0);
    for (int lv = 0; lv < nlev && !done; lv++) {
        const int32_t val = vmin + lv;
        const int beg = lvstart[lv], end = lvstart[lv + 1];
        for (int sbase = beg; sbase < end && !done; sbase += 4096) {
          // most of a long list is dead once the first picks are made: look at 4096 cells at a time first
          // (one barrier), and only walk a 1024-cell chunk when something in the 4096 is still live
          uint32_t rc4[4];
          bool any4 = false;
#pragma unroll
          for (int q = 0; q < 4; q++) {
              const int i = sbase + q * 1024 + tid;
              rc4[q] = i < end ? cells[i] : 0xFFFFFFFFu;
              if (i < end) {
                  const int r = (int)(rc4[q] >> 16), c = (int)(rc4[q] & 0xFFFFu);
                  any4 |= !((rtk[r >> 5] >> (r & 31)) & 1u) && !((ctk[c >> 5] >> (c & 31)) & 1u);
              }
          }
          if (!__syncthreads_or(any4)) continue;
#pragma unroll 1
          for (int q = 0; q < 4 && !done; q++) {
            const int base = sbase + q * 1024;
            if (base >= end) break;
            const int i = base + tid;
            int r = 0, c = 0;
            bool live = false;
            if (i < end) {
                const uint32_t rc = rc4[q];
                r = (int)(rc >> 16);
                c = (int)(rc & 0xFFFFu);
                live = !((rtk[r >> 5] >> (r & 31)) & 1u) && !((ctk[c >> 5] >> (c & 31)) & 1u);
            }
            const int hr = r & hmask, hc = c & hmask;
            bool taken = false;
            for (int pass = 0; pass < 1024; pass++) {
                if (tid == 0) s_any = 0;
                __syncthreads();
                if (live) {
                    atomicMin(&rmin[hr], tid);
                    atomicMin(&cmin[hc], tid);
                    s_any = 1;
                }
                __syncthreads();
                if (!s_any) break;
                // (SYM: a cell whose two customers share a table slot would see its own second atomicMin: same tid, fine)
                const bool win = live && rmin[hr] == tid && cmin[hc] == tid;
                __syncthreads();
                if (live) {   // reset only what was touched
                    rmin[hr] = INT_MAX;
                    cmin[hc] = INT_MAX;
                }
                if (win) {
                    atomicOr(&rtk[r >> 5], 1u << (r & 31));
                    atomicOr(&ctk[c >> 5], 1u << (c & 31));
                    taken = true;
                    live = false;
                }
                __syncthreads();
                if (live) live = !((rtk[r >> 5] >> (r & 31)) & 1u) && !((ctk[c >> 5] >> (c & 31)) & 1u);
            }
            // emit the taken cells of this chunk in list order
            const unsigned long long m = __ballot(taken);
            if (lane == 0) s_w[w] = __popcll(m);
            __syncthreads();
            int wb = 0, tot = 0;
            for (int k = 0; k < 16; k++) {
                wb += (k < w) ? s_w[k] : 0;
                tot += s_w[k];
            }
            const int pos = npairs + wb + __popcll(m & ((1ull << lane) - 1ull));
            if (taken && pos < limit) {
                rows[pos] = r;
                cols[pos] = c;
            }
            const int acc = min(tot, limit - npairs);
            if (acc > 0) {
                last_min = val;
                if ((int64_t)val < sum_below) total += (int64_t)val * acc;
            }
            npairs += acc;
            if (npairs >= limit) done = true;
            __syncthreads();
          }
        }
    }
    // Lists exhausted before the size limit: the reference's next look at the matrix decides
    // last_min (k_lcms_lastmin), it needs to know which rows / columns are gone.
    if (!done) {
        for (int i = tid; i < nw32; i += 1024) {
            g_taken[i] = rtk[i];
            g_taken[nw32 + i] = ctk[i];
        }
    }
    if (tid == 0) {
        *exhausted = done ? 0 : 1;
        out->n_pairs = npairs;
        out->last_min = done ? last_min : INT_MAX;
        out->total = total;
    }
}

// after exhausted lists: smallest cell (below cand_limit) among the rows / columns still free
__global__ __launch_bounds__(256) void k_lcms_lastmin(int n, const int32_t *__restrict__ cost, int64_t cand_limit,
                                                      const int *__restrict__ exhausted,
                                                      const uint32_t *__restrict__ g_taken, LcmOut *__restrict__ out)
{
    if (!*exhausted) return;
    const int nw32 = (n + 31) / 32;
    const uint32_t *rtk = g_taken, *ctk = g_taken + nw32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int mn = INT_MAX;
    for (int row = blockIdx.x * nw + w; row < n; row += gridDim.x * nw) {
        if ((rtk[row >> 5] >> (row & 31)) & 1u) continue;
        const int32_t *rp = cost + (int64_t)row * n;
        for (int j = lane; j < n; j += 64) {
            const int v = rp[j];
            if (!((ctk[j >> 5] >> (j & 31)) & 1u) && (int64_t)v < cand_limit) mn = min(mn, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
    if (lane == 0 && mn != INT_MAX) atomicMin(&out->last_min, mn);
}

}  // namespace

static int lcm_impl(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on, int32_t stop_value, int stop_size,
                    int64_t sum_below, int max_pairs, int32_t *rows, int32_t *cols, int32_t *n_pairs, int64_t *total,
                    int32_t *last_min, int hint_vmin, int hint_vmax);

extern "C" int td_lcm(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on,
                      int32_t stop_value, int stop_size, int64_t sum_below, int max_pairs, int32_t *rows,
                      int32_t *cols, int32_t *n_pairs, int64_t *total, int32_t *last_min)
{
    return lcm_impl(n, cost, mask, threshold, stop_value_on, stop_value, stop_size, sum_below, max_pairs, rows, cols, n_pairs, total,
                    last_min, INT_MAX, INT_MIN);
}

// td_tick knows the candidate cells' value range from the model (0 .. DROP_TIME - 1): no min / max pass, no host round
// trip for it.  A cell outside the hinted range is detected on the device and the call is redone with the measured range.
int td::lcm_hinted(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on, int32_t stop_value, int stop_size,
                   int64_t sum_below, int max_pairs, int32_t *rows, int32_t *cols, int32_t *n_pairs, int64_t *total,
                   int32_t *last_min, int hint_vmin, int hint_vmax)
{
    return lcm_impl(n, cost, mask, threshold, stop_value_on, stop_value, stop_size, sum_below, max_pairs, rows, cols, n_pairs, total,
                    last_min, hint_vmin, hint_vmax);
}

static int lcm_impl(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on, int32_t stop_value, int stop_size,
                    int64_t sum_below, int max_pairs, int32_t *rows, int32_t *cols, int32_t *n_pairs, int64_t *total,
                    int32_t *last_min, int hint_vmin, int hint_vmax)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n < 0 || max_pairs < 0) return fail(TD_EINVAL, "negative size");
    if (n_pairs) *n_pairs = 0;
    if (total) *total = 0;
    if (last_min) *last_min = stop_value;
    if (n == 0) return TD_OK;
    if (!cost || !rows || !cols) return fail(TD_EINVAL, "null array");
    int rc;
    const void *d_cost_v;
    if ((rc = to_device(cost, sizeof(int32_t) * (size_t)n * n, c.stage_d, &d_cost_v))) return rc;
    const int32_t *d_cost = (const int32_t *)d_cost_v;
    if ((rc = ensure(c.lcm_a, sizeof(unsigned long long) * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_b, sizeof(int32_t) * 2 * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_c, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_d, 512 + sizeof(LcmsInfo) * (size_t)c.n_cu * 4))) return rc;
    const int pitch = ((n + 15) / 16) * 16;
    const bool narrow = n >= 128;   // below that a row is a single load per lane anyway
    if (narrow && (rc = ensure(c.cc, (size_t)n * pitch))) return rc;
    int32_t *d_base = (int32_t *)((char *)c.lcm_d.p + 128);
    const int cap = std::min(n, max_pairs);
    int32_t *d_rows = (int32_t *)c.lcm_b.p, *d_cols = d_rows + n;
    // Java's scan only ever sees cells strictly below big_cost (Simulator.java:529-537)
    const int64_t cand_limit = stop_value_on ? (int64_t)stop_value : (int64_t)INT64_MAX;
    const size_t shm_mask = sizeof(uint32_t) * (size_t)((n + 31) / 32);
    bool fast = false, hinted = false;
    int bad_tag = -1;
    int *d_bad = (int *)((char *)c.lcm_d.p + 104);   // a candidate cell outside the level range (hinted ranges only)
    if (g_lcm_lists && n >= 64 && n <= 65536) {
        // level lists: candidates are the cells the loop could ever take
        int64_t hi = std::min<int64_t>(cand_limit - 1, (int64_t)mask - 1);
        if (threshold >= 0) hi = std::min<int64_t>(hi, threshold);
        const int64_t cellsN = (int64_t)n * n;
        const int grid = (int)std::min<int64_t>((cellsN + 4095) / 4096, (int64_t)c.n_cu * 4);
        LcmsInfo *d_info = (LcmsInfo *)((char *)c.lcm_d.p + 512);   // one partial per workgroup
        LcmsInfo info;
        hinted = hint_vmin <= hint_vmax && n <= 4096 && (int64_t)hint_vmax - hint_vmin < LV_MAX;
        if (hinted) {
            info.count = cellsN;   // upper bound: sizes the list buffer
            info.vmin = hint_vmin;
            info.vmax = (int)std::min<int64_t>(hint_vmax, hi);
            if (info.vmax < info.vmin) info.vmax = info.vmin;
        } else {
        {
            ProfScope ps(TD_K_LCM);
            k_lcms_minmax<<<grid, 256, 0, c.stream>>>(n, d_cost, hi, d_info);
        }
        TD_HIP(hipMemcpyAsync(c.pinned, d_info, sizeof(LcmsInfo) * (size_t)grid, hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        info.count = 0;
        info.vmin = INT_MAX;
        info.vmax = INT_MIN;
        for (int g = 0; g < grid; g++) {
            const LcmsInfo &pt = ((const LcmsInfo *)c.pinned)[g];
            if (pt.count) {
                info.count += pt.count;
                info.vmin = std::min(info.vmin, pt.vmin);
                info.vmax = std::max(info.vmax, pt.vmax);
            }
        }
        }
        if (info.count > 0 && info.count <= (1ll << 28) && (int64_t)info.vmax - info.vmin < LV_MAX) {
            fast = true;
            const int nlev = info.vmax - info.vmin + 1;
            const int nw32 = (n + 31) / 32;
            if ((rc = ensure(c.lcm_a, sizeof(int) * (size_t)nlev * n))) return rc;
            if ((rc = ensure(c.cc, sizeof(uint32_t) * (size_t)info.count))) return rc;
            if ((rc = ensure(c.lcm_c, sizeof(int) * (LV_MAX + 2) + sizeof(uint32_t) * 2 * (size_t)nw32 + 64))) return rc;
            int *d_cnt = (int *)c.lcm_a.p;
            int *d_lvstart = (int *)c.lcm_c.p;
            uint32_t *d_taken = (uint32_t *)(d_lvstart + LV_MAX + 2);
            int *d_exh = (int *)((char *)c.lcm_d.p + 96);
            int hsz = 64;
            while (hsz < n && hsz < 16384) hsz <<= 1;
            const int limit = (stop_size >= 0 && stop_size < n) ? std::min(cap, n - stop_size) : cap;
            const size_t shm = sizeof(int) * 2 * (size_t)hsz + sizeof(uint32_t) * 2 * (size_t)nw32;
            ProfScope ps(TD_K_LCM);
            const int rgrid = std::min((n + 3) / 4, c.n_cu * 8);
            if (n <= 4096) {   // one workgroup per row: a third of the latency on tick-sized models
                static int s_tag = 0;
                static const void *s_buf = nullptr;
                if (s_buf != c.lcm_d.p || s_tag == INT_MAX) {   // a fresh buffer (first use, re-init) or a wrap: start the tags over
                    TD_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), c.stream));
                    s_tag = 0;
                    s_buf = c.lcm_d.p;
                }
                bad_tag = ++s_tag;
                k_lcms_rows4<false><<<n, 256, 0, c.stream>>>(n, d_cost, hi, info.vmin, nlev, d_cnt, nullptr, d_bad, bad_tag);
                k_lcms_scan<<<1, 1024, 0, c.stream>>>(nlev * n, n, nlev, d_cnt, d_lvstart);
                k_lcms_rows4<true><<<n, 256, 0, c.stream>>>(n, d_cost, hi, info.vmin, nlev, d_cnt, (uint32_t *)c.cc.p, d_bad, bad_tag);
            } else {
                k_lcms_rows<false><<<rgrid, 256, 0, c.stream>>>(n, d_cost, hi, info.vmin, nlev, d_cnt, nullptr);
                k_lcms_scan<<<1, 1024, 0, c.stream>>>(nlev * n, n, nlev, d_cnt, d_lvstart);
                k_lcms_rows<true><<<rgrid, 256, 0, c.stream>>>(n, d_cost, hi, info.vmin, nlev, d_cnt, (uint32_t *)c.cc.p);
            }
            if (shm > 48 * 1024)
                (void)hipFuncSetAttribute((const void *)k_lcms_greedy<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            k_lcms_greedy<false><<<1, 1024, shm, c.stream>>>(n, hsz - 1, nlev, info.vmin, (const uint32_t *)c.cc.p, d_lvstart, limit,
                                                      sum_below, stop_value, d_rows, d_cols, (LcmOut *)c.lcm_d.p, d_exh, d_taken);
            k_lcms_lastmin<<<rgrid, 256, 0, c.stream>>>(n, d_cost, cand_limit, d_exh, d_taken, (LcmOut *)c.lcm_d.p);
        }
    }
    if (!fast) {
        ProfScope ps(TD_K_LCM);
        TD_HIP(hipMemsetD32Async((hipDeviceptr_t)d_base, INT_MAX, 1, c.stream));
        k_lcm_rowscan<<<std::min((n + 3) / 4, c.n_cu * 8), 256, 0, c.stream>>>(n, d_cost, cand_limit,
                                                                               (unsigned long long *)c.lcm_a.p, d_base);
        if (narrow)
            k_lcm_narrow<<<std::min(n, c.n_cu * 8), 256, 0, c.stream>>>(n, pitch, d_cost, cand_limit, d_base, (uint8_t *)c.cc.p);
        int T = std::min(1024, std::max(64, ((n + 63) / 64) * 64));
        const int rb_in_lds = ((size_t)n * 8 + shm_mask) <= 96 * 1024;
        const size_t shm = shm_mask + (rb_in_lds ? (size_t)n * 8 : 0);
        if (shm > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_lcm_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        k_lcm_loop<<<1, T, shm, c.stream>>>(n, d_cost, cand_limit, mask, threshold, stop_value_on, stop_value,
                                            stop_size, sum_below, cap, (unsigned long long *)c.lcm_a.p, rb_in_lds, d_rows, d_cols,
                                            (int *)c.lcm_c.p, (LcmOut *)c.lcm_d.p, narrow ? (const uint8_t *)c.cc.p : nullptr,
                                            pitch, d_base, 0);
    }
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, c.lcm_d.p, 128, hipMemcpyDeviceToHost, c.stream));   // LcmOut + the flags behind it
    // host outputs of a tick-sized model: the pair lists ride along into the pinned block (one round trip, no
    // blocking pageable copies); d_rows and d_cols are contiguous (n entries each)
    const bool via_pinned = !is_device_ptr(rows) && !is_device_ptr(cols) && (size_t)8192 + sizeof(int32_t) * 2 * (size_t)n <= c.pinned_cap;
    if (via_pinned)
        TD_HIP(hipMemcpyAsync((char *)c.pinned + 8192, d_rows, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    if (hinted && fast && ((const int *)c.pinned)[104 / 4] == bad_tag) {   // the hint was wrong: the lists are void
        return lcm_impl(n, cost, mask, threshold, stop_value_on, stop_value, stop_size, sum_below, max_pairs, rows, cols, n_pairs, total,
                        last_min, INT_MAX, INT_MIN);
    }
    LcmOut o = *(const LcmOut *)c.pinned;
    if (fast && o.last_min == INT_MAX) o.last_min = stop_value_on ? stop_value : mask;   // nothing left to look at
    if (o.n_pairs > 0 && via_pinned) {
        const int32_t *hp = (const int32_t *)((const char *)c.pinned + 8192);
        memcpy(rows, hp, sizeof(int32_t) * (size_t)o.n_pairs);
        memcpy(cols, hp + n, sizeof(int32_t) * (size_t)o.n_pairs);
    } else if (o.n_pairs > 0) {
        const hipMemcpyKind kr = is_device_ptr(rows) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        const hipMemcpyKind kc = is_device_ptr(cols) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        TD_HIP(hipMemcpyAsync(rows, d_rows, sizeof(int32_t) * (size_t)o.n_pairs, kr, c.stream));
        TD_HIP(hipMemcpyAsync(cols, d_cols, sizeof(int32_t) * (size_t)o.n_pairs, kc, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    if (n_pairs) *n_pairs = o.n_pairs;
    if (total) *total = o.total;
    if (last_min) *last_min = o.last_min;
    return TD_OK;
}


// =====================================================================================
// f-3 pool of two (Simulator.java:681-758, pool.c:64-131): every ordered pair (A, B), A != B,
// is a candidate; cost = min(plan1, plan2); plans are taken in stable order of cost (insertion
// order A-major, then B) and a plan is kept iff neither customer is in an earlier kept plan.
// That is the lowest-cost method with SYMMETRIC masking on the n x n pair-cost matrix, so the
// same two kernels do it: k_lcm_rowscan + k_lcm_loop(symmetric).
// =====================================================================================
namespace {

__device__ __forceinline__ int pool_d(const int32_t *dist, int S, int a, int b)
{
    if (!dist) return a > b ? a - b : b - a;
    // never index outside the table: stands outside [0, S) are reported by k_pool2_check (TD_EINVAL)
    a = min(max(a, 0), S - 1);
    b = min(max(b, 0), S - 1);
    return dist[(int64_t)a * S + b];
}

// with a distance table every from / to must be a stand of the table
__global__ void k_pool2_check(int n, const int32_t *__restrict__ from, const int32_t *__restrict__ to, int S, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ((uint32_t)from[i] >= (uint32_t)S || (uint32_t)to[i] >= (uint32_t)S)) atomicOr(flag, 1);
}

// pair cost matrix: pc[A][B] = min(cost1, cost2), diagonal = INT_MAX (never a candidate)
__global__ __launch_bounds__(256) void k_pool2_cost(int n, const int32_t *__restrict__ from,
                                                    const int32_t *__restrict__ to, const int32_t *__restrict__ dist,
                                                    int S, int32_t *__restrict__ pc)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int bf = from[b], bt = to[b];
    for (int a = blockIdx.y; a < n; a += gridDim.y) {
        const int af = from[a], at = to[a];
        const int head = pool_d(dist, S, af, bf);
        const int cost1 = head + pool_d(dist, S, bf, at) + pool_d(dist, S, at, bt);   // Simulator.java:693-695
        const int cost2 = head + pool_d(dist, S, bf, bt) + pool_d(dist, S, bt, at);   // :697-699
        pc[(int64_t)a * n + b] = (a == b) ? INT_MAX : (cost1 < cost2 ? cost1 : cost2);
    }
}

// plan / cost of the kept pairs (Simulator.java:710-717)
__global__ void k_pool2_plans(int k, int n, const int32_t *__restrict__ from, const int32_t *__restrict__ to,
                              const int32_t *__restrict__ dist, int S, const int32_t *__restrict__ ca,
                              const int32_t *__restrict__ cb, int32_t *__restrict__ plan, int32_t *__restrict__ cost)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int a = ca[i], b = cb[i];
    const int af = from[a], at = to[a], bf = from[b], bt = to[b];
    const int head = pool_d(dist, S, af, bf);
    const int cost1 = head + pool_d(dist, S, bf, at) + pool_d(dist, S, at, bt);
    const int cost2 = head + pool_d(dist, S, bf, bt) + pool_d(dist, S, bt, at);
    plan[i] = cost1 < cost2 ? 1 : 0;   // CLNT_B_ENDS : CLNT_A_ENDS
    cost[i] = cost1 < cost2 ? cost1 : cost2;
}

}  // namespace


// =====================================================================================
// Row-sharded LCM: rank r owns cost rows [row0, row0 + nrows) x all n columns.  Per
// pick: every shard reports its smallest live cell in the reference's order (value, row, column),
// the caller takes the lexicographic minimum over all shards (ONE small all-gather), every shard
// masks the row / column and re-scans only its rows whose cached first minimum sat in the taken
// column — the same bookkeeping as k_lcm_loop, so the pairs come out bit-identical to td_lcm.
// The stop rules live in the host driver (taxidispatcher_amd/sharded.py lcm_sharded).
// =====================================================================================
struct td_lcm_shard {
    int n = 0, row0 = 0, nrows = 0;
    int64_t cand_limit = INT64_MAX;
    const int32_t *d_cost = nullptr;
    Buf stage, rowbest, colmask, out, vec, vec2;
};

namespace {

__global__ __launch_bounds__(256) void k_lcmsh_init(int n, int nrows, const int32_t *__restrict__ cost, int64_t cand_limit,
                                                    unsigned long long *__restrict__ rowbest)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = blockIdx.x * nw + w; row < nrows; row += gridDim.x * nw) {
        const unsigned long long b = lcm_scan_row(cost + (int64_t)row * n, n, lane, nullptr, cand_limit);
        if (lane == 0) rowbest[row] = b;
    }
}

// smallest (value, row) among the local rows; out3 = {value, global row, column}, value = INT64_MAX when none
__global__ __launch_bounds__(1024) void k_lcmsh_min(int nrows, int row0, const unsigned long long *__restrict__ rowbest,
                                                    long long *__restrict__ out3)
{
    __shared__ unsigned long long s_red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned long long best = LCM_INF;
    for (int i = tid; i < nrows; i += 1024) {
        const unsigned long long k = rowbest[i];
        if (k != LCM_INF) {
            const unsigned long long kk = (k & 0xFFFFFFFF00000000ull) | (uint32_t)i;
            best = kk < best ? kk : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o);
        best = ob < best ? ob : best;
    }
    if (lane == 0) s_red[w] = best;
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 16; k++) best = s_red[k] < best ? s_red[k] : best;
        best = s_red[0] < best ? s_red[0] : best;
        if (best == LCM_INF) {
            out3[0] = LLONG_MAX;
            out3[1] = -1;
            out3[2] = -1;
        } else {
            const int r = (int)(uint32_t)best;
            out3[0] = (long long)lcm_val(best);
            out3[1] = (long long)(row0 + r);
            out3[2] = (long long)(uint32_t)rowbest[r];
        }
    }
}

__global__ void k_lcmsh_mark(int row_local, int col, unsigned long long *rowbest, uint32_t *colmask)
{
    if (row_local >= 0) rowbest[row_local] = LCM_INF;
    colmask[col >> 5] |= 1u << (col & 31);
}

__global__ __launch_bounds__(256) void k_lcmsh_rescan(int n, int nrows, int col, const int32_t *__restrict__ cost,
                                                      int64_t cand_limit, unsigned long long *__restrict__ rowbest,
                                                      const uint32_t *__restrict__ colmask)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = blockIdx.x * nw + w; row < nrows; row += gridDim.x * nw) {
        const unsigned long long k = rowbest[row];
        if (k == LCM_INF || (int)(uint32_t)k != col) continue;
        const unsigned long long b = lcm_scan_row(cost + (int64_t)row * n, n, lane, colmask, cand_limit);
        if (lane == 0) rowbest[row] = b;
    }
}

// ---- rounds of locally dominant cells --------------------------------------------
// A live cell that is the first minimum of its row AND of its column under the reference's order (value, row,
// column) is taken by the sequential greedy before anything else of its row or column, so ALL such cells can
// be taken in one round; the rounds end when no candidate is left and the picks, sorted by that order, are the
// sequential greedy's picks in its order.  Transport keys are SIGNED: (value << 32) | row, so a MIN / MAX
// all-reduce of int64 vectors orders them like (value, row).
constexpr long long LCMR_NONE_MIN = LLONG_MAX;   // column without a candidate (MIN all-reduce)
constexpr long long LCMR_NONE_MAX = LLONG_MIN;   // column not taken in this round (MAX all-reduce)

// colmin[c] = smallest (value, global row) over this shard's live rows, cells below `limit`, live columns
__global__ __launch_bounds__(256) void k_lcmsh_colmin(int n, int nrows, int row0, const int32_t *__restrict__ cost, long long limit,
                                                      const unsigned long long *__restrict__ rowbest,
                                                      const uint32_t *__restrict__ colmask, long long *__restrict__ colmin)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    long long best = LCMR_NONE_MIN;
    if (!((colmask[c >> 5] >> (c & 31)) & 1u)) {
        for (int r = 0; r < nrows; r++) {
            if (rowbest[r] == LCM_INF) continue;   // taken, or no candidate left in the row (uniform over the workgroup)
            const int32_t v = cost[(int64_t)r * n + c];
            if ((long long)v < limit) {
                const long long k = ((long long)v << 32) | (long long)(uint32_t)(row0 + r);
                best = k < best ? k : best;
            }
        }
    }
    colmin[c] = best;
}

// a row whose first minimum is also its column's minimum takes the column
__global__ __launch_bounds__(256) void k_lcmsh_apply(int n, int nrows, int row0, long long limit,
                                                     const unsigned long long *__restrict__ rowbest,
                                                     const long long *__restrict__ colmin, long long *__restrict__ taken)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const unsigned long long k = rowbest[r];
    if (k == LCM_INF) return;
    const int32_t v = lcm_val(k);
    if ((long long)v >= limit) return;
    const int c = (int)(uint32_t)k;
    const long long mine = ((long long)v << 32) | (long long)(uint32_t)(row0 + r);
    if (colmin[c] == mine) taken[c] = mine;   // one writer per column: the column's minimum is unique
}

__global__ void k_lcmsh_fill64(int n, long long *p, long long v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the round's picks (all shards): their columns are masked everywhere, their rows die on the owning shard
__global__ __launch_bounds__(256) void k_lcmsh_commit(int n, int nrows, int row0, const long long *__restrict__ taken,
                                                      unsigned long long *__restrict__ rowbest, uint32_t *__restrict__ colmask)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const long long t = taken[c];
    if (t == LCMR_NONE_MAX) return;
    atomicOr(&colmask[c >> 5], 1u << (c & 31));
    const int r = (int)(uint32_t)t - row0;
    if (r >= 0 && r < nrows) rowbest[r] = LCM_INF;
}

// rows whose cached first-minimum column has just been masked look for their next one
__global__ __launch_bounds__(256) void k_lcmsh_rescan_masked(int n, int nrows, const int32_t *__restrict__ cost, int64_t cand_limit,
                                                             unsigned long long *__restrict__ rowbest,
                                                             const uint32_t *__restrict__ colmask)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int row = blockIdx.x * nw + w; row < nrows; row += gridDim.x * nw) {
        const unsigned long long k = rowbest[row];
        if (k == LCM_INF) continue;
        const int cc = (int)(uint32_t)k;
        if (!((colmask[cc >> 5] >> (cc & 31)) & 1u)) continue;
        const unsigned long long b = lcm_scan_row(cost + (int64_t)row * n, n, lane, colmask, cand_limit);
        if (lane == 0) rowbest[row] = b;
    }
}

}  // namespace

extern "C" int td_lcm_shard_create(int n, int row0, int nrows, const int32_t *cost_rows, int stop_value_on,
                                   int32_t stop_value, td_lcm_shard **out)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!out) return fail(TD_EINVAL, "null out");
    if (n < 1 || row0 < 0 || nrows < 0 || row0 + nrows > n) return fail(TD_EINVAL, "bad shard geometry n=%d row0=%d nrows=%d", n, row0, nrows);
    if (nrows > 0 && !cost_rows) return fail(TD_EINVAL, "null cost rows");
    td_lcm_shard *s = new td_lcm_shard();
    s->n = n;
    s->row0 = row0;
    s->nrows = nrows;
    s->cand_limit = stop_value_on ? (int64_t)stop_value : (int64_t)INT64_MAX;   // Simulator.java:529-537
    int rc = TD_OK;
    const void *d = nullptr;
    if (nrows) rc = to_device(cost_rows, sizeof(int32_t) * (size_t)nrows * n, s->stage, &d);
    if (!rc) rc = ensure(s->rowbest, sizeof(unsigned long long) * (size_t)std::max(nrows, 1));
    if (!rc) rc = ensure(s->colmask, sizeof(uint32_t) * (size_t)((n + 31) / 32 + 1));
    if (!rc) rc = ensure(s->out, 64);
    if (rc) {
        delete s;
        return rc;
    }
    s->d_cost = (const int32_t *)d;
    TD_HIP(hipMemsetAsync(s->colmask.p, 0, sizeof(uint32_t) * (size_t)((n + 31) / 32 + 1), c.stream));
    if (nrows) {
        ProfScope ps(TD_K_LCM);
        k_lcmsh_init<<<std::max(1, std::min((nrows + 3) / 4, c.n_cu * 8)), 256, 0, c.stream>>>(n, nrows, s->d_cost, s->cand_limit,
                                                                                             (unsigned long long *)s->rowbest.p);
    }
    TD_HIP(hipGetLastError());
    *out = s;
    return TD_OK;
}

extern "C" int td_lcm_shard_destroy(td_lcm_shard *s)
{
    TD_REQUIRE_INIT();
    if (!s) return TD_OK;
    (void)hipStreamSynchronize(ctx().stream);
    Buf *bs[] = {&s->stage, &s->rowbest, &s->colmask, &s->out, &s->vec, &s->vec2};
    for (Buf *b : bs)
        if (b->p) (void)hipFree(b->p);
    delete s;
    return TD_OK;
}

extern "C" int td_lcm_shard_local_min(td_lcm_shard *s, int64_t *out3)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !out3) return fail(TD_EINVAL, "null argument");
    k_lcmsh_min<<<1, 1024, 0, c.stream>>>(s->nrows, s->row0, (const unsigned long long *)s->rowbest.p, (long long *)s->out.p);
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, s->out.p, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    memcpy(out3, c.pinned, 3 * sizeof(int64_t));
    return TD_OK;
}

extern "C" int td_lcm_shard_take(td_lcm_shard *s, int row, int col)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || col < 0 || col >= s->n || row < 0 || row >= s->n) return fail(TD_EINVAL, "bad pick (%d, %d)", row, col);
    const int rl = (row >= s->row0 && row < s->row0 + s->nrows) ? row - s->row0 : -1;
    k_lcmsh_mark<<<1, 1, 0, c.stream>>>(rl, col, (unsigned long long *)s->rowbest.p, (uint32_t *)s->colmask.p);
    if (s->nrows)
        k_lcmsh_rescan<<<std::max(1, std::min((s->nrows + 3) / 4, c.n_cu * 8)), 256, 0, c.stream>>>(
            s->n, s->nrows, col, s->d_cost, s->cand_limit, (unsigned long long *)s->rowbest.p, (const uint32_t *)s->colmask.p);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

// ---- rounds of locally dominant cells: three calls per round, two exchanges of n int64 between them ----
extern "C" int td_lcm_shard_round_colmin(td_lcm_shard *s, int64_t limit, int64_t *colmin)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !colmin) return fail(TD_EINVAL, "null argument");
    const bool dev = is_device_ptr(colmin);
    long long *d = (long long *)colmin;
    if (!dev) {
        int rc = ensure(s->vec, sizeof(long long) * (size_t)s->n);
        if (rc) return rc;
        d = (long long *)s->vec.p;
    }
    ProfScope ps(TD_K_LCM);
    k_lcmsh_colmin<<<(s->n + 255) / 256, 256, 0, c.stream>>>(s->n, s->nrows, s->row0, s->d_cost, (long long)limit,
                                                             (const unsigned long long *)s->rowbest.p, (const uint32_t *)s->colmask.p, d);
    TD_HIP(hipGetLastError());
    if (!dev) {
        TD_HIP(hipMemcpyAsync(colmin, d, sizeof(long long) * (size_t)s->n, hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    return TD_OK;
}

extern "C" int td_lcm_shard_round_apply(td_lcm_shard *s, int64_t limit, const int64_t *colmin, int64_t *taken)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !colmin || !taken) return fail(TD_EINVAL, "null argument");
    const void *dcm = nullptr;
    int rc = to_device(colmin, sizeof(long long) * (size_t)s->n, s->vec2, &dcm);
    if (rc) return rc;
    const bool dev = is_device_ptr(taken);
    long long *d = (long long *)taken;
    if (!dev) {
        rc = ensure(s->vec, sizeof(long long) * (size_t)s->n);
        if (rc) return rc;
        d = (long long *)s->vec.p;
    }
    ProfScope ps(TD_K_LCM);
    k_lcmsh_fill64<<<(s->n + 255) / 256, 256, 0, c.stream>>>(s->n, d, LCMR_NONE_MAX);
    if (s->nrows)
        k_lcmsh_apply<<<(s->nrows + 255) / 256, 256, 0, c.stream>>>(s->n, s->nrows, s->row0, (long long)limit,
                                                                    (const unsigned long long *)s->rowbest.p, (const long long *)dcm, d);
    TD_HIP(hipGetLastError());
    if (!dev) {
        TD_HIP(hipMemcpyAsync(taken, d, sizeof(long long) * (size_t)s->n, hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    return TD_OK;
}

extern "C" int td_lcm_shard_round_commit(td_lcm_shard *s, const int64_t *taken)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !taken) return fail(TD_EINVAL, "null argument");
    const void *dt = nullptr;
    int rc = to_device(taken, sizeof(long long) * (size_t)s->n, s->vec2, &dt);
    if (rc) return rc;
    ProfScope ps(TD_K_LCM);
    k_lcmsh_commit<<<(s->n + 255) / 256, 256, 0, c.stream>>>(s->n, s->nrows, s->row0, (const long long *)dt,
                                                             (unsigned long long *)s->rowbest.p, (uint32_t *)s->colmask.p);
    if (s->nrows)
        k_lcmsh_rescan_masked<<<std::max(1, std::min((s->nrows + 3) / 4, c.n_cu * 8)), 256, 0, c.stream>>>(
            s->n, s->nrows, s->d_cost, s->cand_limit, (unsigned long long *)s->rowbest.p, (const uint32_t *)s->colmask.p);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

extern "C" int td_pool2(int n, const int32_t *from, const int32_t *to, const int32_t *dist, int S, int32_t *cust_a,
                        int32_t *cust_b, int32_t *plan, int32_t *cost, int32_t *n_pairs)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n_pairs) *n_pairs = 0;
    if (n < 0) return fail(TD_EINVAL, "n < 0");
    if (n < 2) return TD_OK;
    if (!from || !to || !cust_a || !cust_b || !plan || !cost) return fail(TD_EINVAL, "null array");
    if (dist && S <= 0) return fail(TD_EINVAL, "dist given but S=%d", S);
    int rc;
    const void *d_from, *d_to, *d_dist = nullptr;
    if ((rc = ensure(c.stage_a, sizeof(int32_t) * 2 * (size_t)n))) return rc;
    if (is_device_ptr(from)) d_from = from;
    else {
        TD_HIP(hipMemcpyAsync(c.stage_a.p, from, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c.stream));
        d_from = c.stage_a.p;
    }
    if (is_device_ptr(to)) d_to = to;
    else {
        TD_HIP(hipMemcpyAsync((int32_t *)c.stage_a.p + n, to, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c.stream));
        d_to = (int32_t *)c.stage_a.p + n;
    }
    if (dist && (rc = to_device(dist, sizeof(int32_t) * (size_t)S * S, c.stage_c, &d_dist))) return rc;
    if ((rc = ensure(c.stage_d, sizeof(int32_t) * (size_t)n * n))) return rc;   // pair-cost matrix
    if ((rc = ensure(c.lcm_a, sizeof(unsigned long long) * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_b, sizeof(int32_t) * 4 * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_c, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_d, 512 + sizeof(LcmsInfo) * (size_t)c.n_cu * 4))) return rc;   // out, flags, min/max partials
    int32_t *pc = (int32_t *)c.stage_d.p;
    int32_t *d_a = (int32_t *)c.lcm_b.p, *d_b = d_a + n, *d_plan = d_a + 2 * n, *d_cost = d_a + 3 * n;
    const int pitch = ((n + 15) / 16) * 16;
    const bool narrow = n >= 128;
    if (narrow && (rc = ensure(c.cc, (size_t)n * pitch))) return rc;
    int32_t *d_base = (int32_t *)((char *)c.lcm_d.p + 128);
    const int64_t cand_limit = (int64_t)INT_MAX;   // the diagonal (INT_MAX) is not a candidate
    const size_t shm_mask = sizeof(uint32_t) * (size_t)((n + 31) / 32);
    {
        ProfScope ps(TD_K_LCM);
        TD_HIP(hipMemsetAsync((char *)c.lcm_d.p + 192, 0, sizeof(int), c.stream));
        if (d_dist)
            k_pool2_check<<<(n + 255) / 256, 256, 0, c.stream>>>(n, (const int32_t *)d_from, (const int32_t *)d_to, S, (int *)((char *)c.lcm_d.p + 192));
        dim3 g((n + 255) / 256, std::min(n, 1024));
        k_pool2_cost<<<g, 256, 0, c.stream>>>(n, (const int32_t *)d_from, (const int32_t *)d_to, (const int32_t *)d_dist, S, pc);
    }
    // The greedy pass over the n (n - 1) ordered pairs: level lists as in td_lcm (pair costs are sums of three
    // distances: a few dozen to ~150 distinct values), walked by k_lcms_greedy<SYM>; the row-scan loop otherwise.
    bool fast = false, hinted = false;
    int bad_tag = -1;
    int *d_bad = (int *)((char *)c.lcm_d.p + 104);   // a candidate cell outside the level range (hinted ranges only)
    if (g_lcm_lists && n >= 64 && n <= 65536) {
        const int64_t hi = (int64_t)INT_MAX - 1;
        const int64_t cellsN = (int64_t)n * n;
        const int grid = (int)std::min<int64_t>((cellsN + 4095) / 4096, (int64_t)c.n_cu * 4);
        LcmsInfo *d_info = (LcmsInfo *)((char *)c.lcm_d.p + 512);
        {
            ProfScope ps(TD_K_LCM);
            k_lcms_minmax<<<grid, 256, 0, c.stream>>>(n, pc, hi, d_info);
        }
        TD_HIP(hipMemcpyAsync(c.pinned, d_info, sizeof(LcmsInfo) * (size_t)grid, hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        LcmsInfo info;
        info.count = 0;
        info.vmin = INT_MAX;
        info.vmax = INT_MIN;
        for (int g = 0; g < grid; g++) {
            const LcmsInfo &pt = ((const LcmsInfo *)c.pinned)[g];
            if (pt.count) {
                info.count += pt.count;
                info.vmin = std::min(info.vmin, pt.vmin);
                info.vmax = std::max(info.vmax, pt.vmax);
            }
        }
        if (info.count > 0 && info.count <= (1ll << 28) && (int64_t)info.vmax - info.vmin < LV_MAX) {
            fast = true;
            const int nlev = info.vmax - info.vmin + 1;
            const int nw32 = (n + 31) / 32;
            if ((rc = ensure(c.lcm_a, sizeof(int) * (size_t)nlev * n))) return rc;
            if ((rc = ensure(c.cc, sizeof(uint32_t) * (size_t)info.count))) return rc;
            if ((rc = ensure(c.lcm_c, sizeof(int) * (LV_MAX + 2) + sizeof(uint32_t) * 2 * (size_t)nw32 + 64))) return rc;
            int *d_cnt = (int *)c.lcm_a.p;
            int *d_lvstart = (int *)c.lcm_c.p;
            uint32_t *d_taken = (uint32_t *)(d_lvstart + LV_MAX + 2);
            int *d_exh = (int *)((char *)c.lcm_d.p + 96);
            int hsz = 64;
            while (hsz < n && hsz < 16384) hsz <<= 1;
            const size_t shm = sizeof(int) * 2 * (size_t)hsz + sizeof(uint32_t) * 2 * (size_t)nw32;
            ProfScope ps(TD_K_LCM);
            const int rgrid = std::min((n + 3) / 4, c.n_cu * 8);
            k_lcms_rows<false><<<rgrid, 256, 0, c.stream>>>(n, pc, hi, info.vmin, nlev, d_cnt, nullptr);
            k_lcms_scan<<<1, 1024, 0, c.stream>>>(nlev * n, n, nlev, d_cnt, d_lvstart);
            k_lcms_rows<true><<<rgrid, 256, 0, c.stream>>>(n, pc, hi, info.vmin, nlev, d_cnt, (uint32_t *)c.cc.p);
            if (shm > 48 * 1024)
                (void)hipFuncSetAttribute((const void *)k_lcms_greedy<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            k_lcms_greedy<true><<<1, 1024, shm, c.stream>>>(n, hsz - 1, nlev, info.vmin, (const uint32_t *)c.cc.p, d_lvstart, n / 2,
                                                            (int64_t)INT64_MAX, 0, d_a, d_b, (LcmOut *)c.lcm_d.p, d_exh, d_taken);
        }
    }
    if (!fast) {
        ProfScope ps(TD_K_LCM);
        TD_HIP(hipMemsetD32Async((hipDeviceptr_t)d_base, INT_MAX, 1, c.stream));
        k_lcm_rowscan<<<std::min((n + 3) / 4, c.n_cu * 8), 256, 0, c.stream>>>(n, pc, cand_limit, (unsigned long long *)c.lcm_a.p, d_base);
        if (narrow) k_lcm_narrow<<<std::min(n, c.n_cu * 8), 256, 0, c.stream>>>(n, pitch, pc, cand_limit, d_base, (uint8_t *)c.cc.p);
        int T = std::min(1024, std::max(64, ((n + 63) / 64) * 64));
        const int rb_in_lds = ((size_t)n * 8 + shm_mask) <= 96 * 1024;
        const size_t shm = shm_mask + (rb_in_lds ? (size_t)n * 8 : 0);
        if (shm > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_lcm_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        k_lcm_loop<<<1, T, shm, c.stream>>>(n, pc, cand_limit, INT_MAX, -1, 0, 0, -1, (int64_t)INT_MAX, n / 2, (unsigned long long *)c.lcm_a.p,
                                            rb_in_lds, d_a, d_b, (int *)c.lcm_c.p, (LcmOut *)c.lcm_d.p,
                                            narrow ? (const uint8_t *)c.cc.p : nullptr, pitch, d_base, 1);
    }
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, c.lcm_d.p, 256, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    if (*(const int *)((const char *)c.pinned + 192)) return fail(TD_EINVAL, "td_pool2: a from / to stand lies outside the %d x %d distance table", S, S);
    const int k = ((const LcmOut *)c.pinned)->n_pairs;
    if (k > 0) {
        k_pool2_plans<<<(k + 255) / 256, 256, 0, c.stream>>>(k, n, (const int32_t *)d_from, (const int32_t *)d_to,
                                                             (const int32_t *)d_dist, S, d_a, d_b, d_plan, d_cost);
        TD_HIP(hipGetLastError());
        int32_t *outs[4] = {cust_a, cust_b, plan, cost};
        int32_t *srcs[4] = {d_a, d_b, d_plan, d_cost};
        for (int q = 0; q < 4; q++)
            TD_HIP(hipMemcpyAsync(outs[q], srcs[q], sizeof(int32_t) * (size_t)k,
                                  is_device_ptr(outs[q]) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    if (n_pairs) *n_pairs = k;
    return TD_OK;
}

// =====================================================================================
// td_tick's LCM on STANDS (round 4).  The Simulator's model is cost = |cab stand - request stand| below DROP_TIME
// on N_STANDS = 50 stands (Simulator.java:110,493-520,553-560): every cab of a stand has the SAME row, every request
// of a stand the same column.  The lowest-cost method (Simulator.java:523-549: repeat { first minimum in row-major order;
// mask its row and column }) then never needs the matrix:
//   * at value L a cab at stand a can only take a request at stand a - L or a + L, and among those the one with the
//     smallest index — the HEAD of that stand's queue of untaken requests (requests of a stand are taken in index order);
//   * among the cabs of a stand the one with the smallest index goes first (same reason): per stand two counters
//     describe everything that has happened;
//   * value 0: the stands are independent — stand a pairs its t-th cab with its t-th request, t < min(cabs, requests);
//     the picks of a level come out in increasing row order, so a pick's place in the list is the rank of its row among
//     the picked rows (a bitmap + prefix popcounts), and the `limit` smallest rows are the ones the loop gets to;
//   * value L >= 1: one pick at a time — every stand (a lane) offers its smallest untaken cab if one of its two queues
//     has a head, the wave minimum is the next row, its column the smaller of the two heads.
// One workgroup; the typical tick (supply above demand at most stands) is done at value 0: ~10 us instead of the ~110 us
// of the level-list build + greedy walk over the 1300 x 900 matrix, whose 6.8 MB are then never written either.
// Positions outside 0..63 (or a threshold above 64) raise a flag: the caller takes the general path.
// =====================================================================================
namespace {

constexpr int LST_MAXN = 2048;   // cabs / requests the one-workgroup kernel holds in LDS (4 arrays of them: 32 KB)

__device__ __forceinline__ int lst_wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(256) void k_lcm_stands(int n_s, int n_d, int thr, int limit, int32_t fill, const int32_t *__restrict__ cab_to,
                                                    const int32_t *__restrict__ dem_from, int32_t *__restrict__ rows,
                                                    int32_t *__restrict__ cols, LcmOut *__restrict__ out, int *__restrict__ bad)
{
    __shared__ int s_pos_c[LST_MAXN], s_pos_r[LST_MAXN];     // stand of every cab / request
    __shared__ int s_clist[LST_MAXN], s_rlist[LST_MAXN];     // cabs / requests grouped by stand, index order inside a stand
    __shared__ int s_cnt_c[64], s_cnt_r[64], s_off_c[64], s_off_r[64];
    __shared__ unsigned short s_sl[2][64][64];               // per side: elements of stand a in slice t (then their offset)
    __shared__ uint32_t s_bits[LST_MAXN / 32];
    __shared__ int s_wpre[LST_MAXN / 32 + 1];
    __shared__ int s_bad, s_k, s_done;
    __shared__ long long s_total;
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 64) s_cnt_c[tid] = 0, s_cnt_r[tid] = 0;
    if (tid == 0) s_bad = 0, s_k = 0, s_done = 0, s_total = 0, s_last = fill;
    for (int i = tid; i < LST_MAXN / 32; i += 256) s_bits[i] = 0u;
    __syncthreads();
    for (int i = tid; i < n_s; i += 256) {
        const int a = cab_to[i];
        s_pos_c[i] = a;
        if ((unsigned)a >= 64u) s_bad = 1;
        else atomicAdd(&s_cnt_c[a], 1);
    }
    for (int i = tid; i < n_d; i += 256) {
        const int b = dem_from[i];
        s_pos_r[i] = b;
        if ((unsigned)b >= 64u) s_bad = 1;
        else atomicAdd(&s_cnt_r[b], 1);
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) *bad = 1;
        return;
    }
    if (tid == 0) {
        int oc = 0, orr = 0;
        for (int a = 0; a < 64; a++) {
            s_off_c[a] = oc;
            s_off_r[a] = orr;
            oc += s_cnt_c[a];
            orr += s_cnt_r[a];
        }
    }
    // stable grouping by stand (index order inside a stand) without a 1300-step walk per stand: thread t of wave 0 (cabs) /
    // wave 1 (requests) owns a contiguous slice of the elements, counts its slice per stand into s_sl[side][t][stand],
    // lane `stand` turns the 64 slice counts of its stand into offsets, then every thread places its slice
    if (w < 2) {
        const int cnt = w == 0 ? n_s : n_d;
        const int *pos = w == 0 ? s_pos_c : s_pos_r;
        const int per = (cnt + 63) / 64, lo = lane * per, hi = min(cnt, lo + per);
        unsigned short *mine = &s_sl[w][lane][0];
        for (int a = 0; a < 64; a++) mine[a] = 0;
        for (int i = lo; i < hi; i++) mine[pos[i]]++;
    }
    __syncthreads();
    if (w < 2) {   // lane = stand: exclusive prefix over the 64 slices
        int acc = 0;
        for (int t = 0; t < 64; t++) {
            const int v = s_sl[w][t][lane];
            s_sl[w][t][lane] = (unsigned short)acc;
            acc += v;
        }
    }
    __syncthreads();
    if (w < 2) {
        const int cnt = w == 0 ? n_s : n_d;
        const int *pos = w == 0 ? s_pos_c : s_pos_r;
        const int *off = w == 0 ? s_off_c : s_off_r;
        int *list = w == 0 ? s_clist : s_rlist;
        const int per = (cnt + 63) / 64, lo = lane * per, hi = min(cnt, lo + per);
        unsigned short *mine = &s_sl[w][lane][0];
        for (int i = lo; i < hi; i++) {
            const int a = pos[i];
            list[off[a] + mine[a]] = i;
            mine[a]++;
        }
    }
    __syncthreads();
    // ---- value 0: every stand pairs its t-th cab with its t-th request; the picked rows are marked in a bitmap
    const int m0 = (tid < 64) ? min(s_cnt_c[tid], s_cnt_r[tid]) : 0;
    if (tid < 64 && thr > 0)
        for (int t = 0; t < m0; t++) {
            const int r = s_clist[s_off_c[tid] + t];
            atomicOr(&s_bits[r >> 5], 1u << (r & 31));
        }
    __syncthreads();
    const int nw32 = (n_s + 31) / 32;
    if (tid == 0) {   // prefix of the word popcounts (<= 128 words)
        int acc = 0;
        for (int i = 0; i < nw32; i++) {
            s_wpre[i] = acc;
            acc += __popc(s_bits[i]);
        }
        s_wpre[nw32] = acc;
    }
    __syncthreads();
    const int p0 = thr > 0 ? s_wpre[nw32] : 0;   // picks available at value 0
    const int take0 = min(p0, limit);
    if (tid < 64 && thr > 0)
        for (int t = 0; t < m0; t++) {
            const int r = s_clist[s_off_c[tid] + t];
            const int rank = s_wpre[r >> 5] + __popc(s_bits[r >> 5] & ((1u << (r & 31)) - 1u));
            if (rank < take0) {
                rows[rank] = r;
                cols[rank] = s_rlist[s_off_r[tid] + t];
            }
        }
    // how far every stand got at value 0 (only picks of rank < take0 happened)
    __shared__ int s_ca[64], s_rb[64];
    if (tid < 64) {
        int used = 0;
        if (thr > 0) {
            if (take0 == p0)
                used = m0;
            else   // the loop stopped inside value 0: the stand's picks with rank < take0 (its cabs are in increasing order)
                for (int t = 0; t < m0; t++) {
                    const int r = s_clist[s_off_c[tid] + t];
                    const int rank = s_wpre[r >> 5] + __popc(s_bits[r >> 5] & ((1u << (r & 31)) - 1u));
                    used += rank < take0 ? 1 : 0;
                }
        }
        s_ca[tid] = used;
        s_rb[tid] = used;
    }
    __syncthreads();
    int k = take0;
    if (w == 0) {
        long long total = 0;
        int last = take0 > 0 ? 0 : fill;
        bool done = k >= limit;
        int ca = s_ca[lane], rb = s_rb[lane];
        const int ncab = s_cnt_c[lane], nreq = s_cnt_r[lane], offc = s_off_c[lane], offr = s_off_r[lane];
        for (int L = 1; L < thr && !done; L++) {
            for (;;) {
                const int h = rb < nreq ? s_rlist[offr + rb] : INT_MAX;     // head of this stand's request queue
                const int hl = __shfl(h, (lane - L) & 63), hr = __shfl(h, (lane + L) & 63);
                const int req = min(lane - L >= 0 ? hl : INT_MAX, lane + L < 64 ? hr : INT_MAX);
                const int cand = (ca < ncab && req != INT_MAX) ? s_clist[offc + ca] : INT_MAX;
                const int best = lst_wave_min(cand);
                if (best == INT_MAX) break;   // nothing left at this value
                const unsigned long long wm = __ballot(cand == best);
                const int wl = __builtin_ctzll(wm);
                const int c = __shfl(req, wl);
                const int bst = s_pos_r[c];   // the stand whose queue loses its head
                if (lane == 0) {
                    rows[k] = best;
                    cols[k] = c;
                }
                if (lane == wl) ca++;
                if (lane == bst) rb++;
                k++;
                total += L;
                last = L;
                if (k >= limit) {
                    done = true;
                    break;
                }
            }
        }
        if (lane == 0) {
            out->n_pairs = k;
            out->last_min = done ? last : fill;   // lists exhausted before the size limit: the next look at the matrix sees big_cost (Simulator.java:538)
            out->total = total;
        }
    }
}

}  // namespace

// td_tick: the LCM of a thresholded |a - b| model straight from the position arrays (k_lcm_stands).  *ok = 0: the model is
// not of that kind (a position outside 0..63) — nothing was done, the caller builds the matrix and calls lcm_hinted.
int td::lcm_stands(int n_s, int n_d, const int32_t *d_cab_to, const int32_t *d_dem_from, int32_t fill, int32_t threshold, int stop_size,
                   int32_t *rows, int32_t *cols, int32_t *n_pairs, int32_t *last_min, int *ok)
{
    Ctx &c = ctx();
    const int n = std::max(n_s, n_d);
    *ok = 0;
    if (threshold < 1 || threshold > 64 || n_s > LST_MAXN || n_d > LST_MAXN || stop_size < 0 || stop_size >= n) return TD_OK;
    int rc;
    if ((rc = ensure(c.lcm_b, sizeof(int32_t) * 2 * (size_t)n))) return rc;
    if ((rc = ensure(c.lcm_d, 512 + sizeof(LcmsInfo) * (size_t)c.n_cu * 4))) return rc;
    int32_t *d_rows = (int32_t *)c.lcm_b.p, *d_cols = d_rows + n;
    LcmOut *d_out = (LcmOut *)c.lcm_d.p;
    int *d_bad = (int *)((char *)c.lcm_d.p + 104);
    ProfScope ps(TD_K_LCM);
    TD_HIP(hipMemsetAsync(c.lcm_d.p, 0, 128, c.stream));
    k_lcm_stands<<<1, 256, 0, c.stream>>>(n_s, n_d, threshold, n - stop_size, fill, d_cab_to, d_dem_from, d_rows, d_cols, d_out, d_bad);
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, c.lcm_d.p, 128, hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    if (*(const int *)((const char *)c.pinned + 104)) return TD_OK;   // not a model on <= 64 stands
    const LcmOut *ho = (const LcmOut *)c.pinned;
    const int k = ho->n_pairs;
    const int32_t lm = ho->last_min;
    // (the pair list stays on the device — c.lcm_b: rows, then cols, n entries each — for the caller's shrink kernel; td_tick
    // brings it home in its one pinned read-back at the end: two copies into pageable host arrays here cost ~20 us each)
    (void)rows;
    (void)cols;
    *n_pairs = k;
    *last_min = lm;
    *ok = 1;
    return TD_OK;
}
