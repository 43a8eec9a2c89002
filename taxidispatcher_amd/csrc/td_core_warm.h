// td_core_warm.h — included by td_assign.hip (inside its anonymous namespace, after k_bid / k_assign).
//
// SPARSE CORE for the price warm start of wide, tie-free rows (uniform 0..10^6, 2-D grids, general S x S tables
// without ties).  The warm start (sv_warm_t: Bertsekas auction phases with eps > 0, of which ONLY
// THE PRICES are kept — any non-negative price vector is dual feasible, exactness rests on the eps = 0 rounds, the
// shortest augmenting paths and the LP certificate that follow) was the largest piece of those solves: hundreds to
// thousands of rounds that each STREAM the 4-byte rows of every bidder (43 us per round at n = 16 384).  But a bid only
// needs a row's two best values of cost + price, and on such rows those sit among the row's few dozen cheapest cells.
//
//   k_core_build   one workgroup per row: every thread takes the minimum of its own cells (256 group minima), the
//                  k-th smallest of them is the row's threshold T (at least k cells are <= T; about 256 * -ln(1 - k/256)
//                  on random rows: 74 for k = 64), a second pass over the row (L2) lists the cells <= T as
//                  (value, column) pairs — at most CORE_CAP per row, the cheapest-first order does not matter.
//   k_bid_core     the eps-phase bidding round on those lists: one wave per free row, 2 entries per lane, prices
//                  gathered from L2 — 8 MB instead of 1 GiB per full round at n = 16 384.
// The list is only trusted where it provably holds the row's best column: every cell outside it is >= t (the smallest
// value above T) and every price is >= pmin (the smallest price of any column, refreshed every 8 rounds by k_price_min;
// prices only rise, so an older value is still a bound), so a best list value <= t + pmin IS the row's best value, and
// min(second list value, t + pmin) is a lower bound of its second best (a smaller raise: still a valid bid).  A row whose best list value has been priced
// above t is flagged and bids on its DENSE row in the same round (k_bid with a row mask): on random rows that is a
// handful per round, on geometric rows (2-D grids, |a - b| with the recogniser off: whole crowded regions outgrow their
// lists) it is most of them — the first version, which let every row bid on its list alone, left 8 090 instead of 1 369
// rows free on the 2-D grid and took 672 instead of 223 ms.  When most bidders are flagged the lists are dropped.
// None of this touches exactness: only the prices of the warm start are kept.

constexpr int CORE_CAP = 128;   // entries kept per row

template <typename CT>
__global__ __launch_bounds__(256) void k_core_build(int n, int nrows, int nchunks, int kth, const CT *__restrict__ cc,
                                                    uint2 *__restrict__ core /* [nrows][CORE_CAP] (value, column) */,
                                                    int *__restrict__ core_n, uint32_t *__restrict__ core_t /* smallest value outside the list */,
                                                    unsigned long long *__restrict__ t_sum = nullptr)
{
    constexpr int E = Tr<CT>::E;
    __shared__ uint32_t s_min[256];
    __shared__ uint32_t s_T;
    __shared__ int s_cnt;
    __shared__ uint32_t s_out[4];
    const int tid = threadIdx.x;
    const size_t pitch = (size_t)nchunks * E;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const CT *rp = cc + (size_t)row * pitch;
        uint32_t mn = 0xFFFFFFFFu;
        for (int t = tid; t < nchunks; t += 256) {
            uint32_t c[E];
            unpack<CT>(*reinterpret_cast<const uint4 *>(rp + (size_t)t * E), c);
#pragma unroll
            for (int e = 0; e < E; e++) mn = min(mn, c[e]);   // (pad cells hold the sentinel: never the minimum of a group that has a real cell)
        }
        s_min[tid] = mn;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        // rank of this thread's minimum among the 256 (ties by thread id): the thread of rank kth - 1 publishes T
        int rank = 0;
        for (int k = 0; k < 256; k++) {
            const uint32_t o = s_min[k];
            rank += (o < mn || (o == mn && k < tid)) ? 1 : 0;
        }
        if (rank == kth - 1) s_T = mn;
        __syncthreads();
        const uint32_t T = s_T;
        uint2 *dst = core + (size_t)row * CORE_CAP;
        uint32_t mo = 0xFFFFFFFFu;   // smallest cell above T
        for (int t = tid; t < nchunks; t += 256) {
            uint32_t c[E];
            unpack<CT>(*reinterpret_cast<const uint4 *>(rp + (size_t)t * E), c);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int j = t * E + e;
                if (j >= n) continue;
                if (c[e] <= T) {
                    const int k = atomicAdd(&s_cnt, 1);
                    if (k < CORE_CAP) dst[k] = make_uint2(c[e], (uint32_t)j);
                } else
                    mo = min(mo, c[e]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mo = min(mo, (uint32_t)__shfl_xor((int)mo, o));
        if ((tid & 63) == 0) s_out[tid >> 6] = mo;
        __syncthreads();
        if (tid == 0) {
            core_n[row] = min(s_cnt, CORE_CAP);
            // more cells <= T than the list holds: some of them are outside, the list proves nothing (t = 0: always dense)
            const uint32_t tv = s_cnt > CORE_CAP ? 0u : min(min(s_out[0], s_out[1]), min(s_out[2], s_out[3]));
            core_t[row] = tv;
            if (t_sum && tv != 0xFFFFFFFFu) atomicAdd(t_sum, (unsigned long long)tv);   // the lists' average reach (sv_warm_t: which phases may use them)
        }
        __syncthreads();
    }
}

// eps-phase bidding round on the core lists (the EPSM branch of k_bid with kscale = 1): key = 2 * cost + packed price
template <typename PT>
__global__ __launch_bounds__(256) void k_bid_core(int n, int nrows, int row0, const uint2 *__restrict__ core,
                                                  const int *__restrict__ core_n, const uint32_t *__restrict__ core_t,
                                                  const PT *__restrict__ pk, const int *__restrict__ r2c,
                                                  unsigned long long *__restrict__ bid, int *__restrict__ ctl, long long eps,
                                                  uint8_t *__restrict__ need_dense /* 1: the row's list does not prove its best column, k_bid takes the dense row */)
{
    constexpr PT KMAXV = sizeof(PT) == 4 ? (PT)INT32_MAX : (PT)INT64_MAX;
    if (ctl[CTL_FLAG]) return;
    const int lane = threadIdx.x & 63;
    const int lrow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lrow >= nrows) return;
    if (r2c[lrow] != -1) return;
    const int m = core_n[lrow];
    const uint2 *src = core + (size_t)lrow * CORE_CAP;
    PT k1 = KMAXV, k2 = KMAXV;
    int j1 = -1;
#pragma unroll
    for (int u = 0; u < CORE_CAP / 64; u++) {
        const int k = lane + 64 * u;
        if (k < m) {
            const uint2 e = src[k];
            const PT key = (PT)(2 * (PT)e.x) + pk[e.y];
            const bool lt = key < k1;
            const PT mx = key > k1 ? key : k1;
            k2 = k2 < mx ? k2 : mx;
            j1 = lt ? (int)e.y : j1;
            k1 = lt ? key : k1;
        }
    }
    // wave butterfly: lexicographic min of (key, column)
    PT bk = k1;
    int bj = (k1 == KMAXV) ? INT_MAX : j1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const PT ok = shfl_xor_t(bk, o);
        const int oj = __shfl_xor(bj, o);
        if (ok < bk || (ok == bk && oj < bj)) {
            bk = ok;
            bj = oj;
        }
    }
    const bool winner = (k1 == bk) && (bj == j1) && (k1 != KMAXV);
    PT x = winner ? k2 : k1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const PT ox = shfl_xor_t(x, o);
        x = ox < x ? ox : x;
    }
    if (lane == 0) {
        atomicAdd(&ctl[CTL_COREMISS + 1], 1);   // bidders seen
        const PT tv = (PT)core_t[lrow] + (PT)(*reinterpret_cast<const long long *>(&ctl[CTL_COREMISS + 2]));   // every cell outside the list is >= t, at a price >= pmin
        const bool have = bj != INT_MAX && bj < n;
        const bool proven = need_dense ? (have && (bk >> 1) <= tv) : have;   // need_dense == null: the list is taken on trust (late phases, see sv_warm_t)
        if (need_dense) need_dense[lrow] = proven ? 0 : 1;
        if (proven) {
            const PT xe = need_dense ? ((x == KMAXV || (x >> 1) > tv) ? (PT)(2 * tv) : x)   // second best: the list's, or the bound of everything outside
                                     : (x == KMAXV ? bk : x);
            const PT inc = (PT)((xe >> 1) - (bk >> 1));
            const PT newp = (pk[bj] >> 1) + inc + (PT)eps;
            atomicMax(&bid[bj], ((unsigned long long)newp << ROW_BITS) | (unsigned long long)(row0 + lrow + 1));
        } else
            atomicAdd(&ctl[CTL_COREMISS], 1);
    }
}

// smallest packed price over all columns -> the 64-bit word at ctl[CTL_COREMISS + 2] (as a plain price)
template <typename PT>
__global__ __launch_bounds__(1024) void k_price_min(int n, const PT *__restrict__ pk, int *__restrict__ ctl)
{
    __shared__ long long s_m[16];
    long long m = LLONG_MAX;
    for (int j = threadIdx.x; j < n; j += 1024) m = min(m, (long long)(pk[j] >> 1));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long om = __shfl_xor(m, o);
        m = om < m ? om : m;
    }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; k++) m = min(m, s_m[k]);
        *reinterpret_cast<long long *>(&ctl[CTL_COREMISS + 2]) = m;
    }
}

// Is the matrix "random-like" (independent cells: uniform 0..10^6) or does it have metric structure (2-D grids, |a - b|)?  The
// warm start's best schedule differs (profiles/r4/general_solver_r4_schedule.txt: random-like rows are fastest when the eps
// ladder starts at range / 8192 and 48 eps = 0 rounds follow — 32 -> 23 ms at n = 16 384 —, geometric ones need the whole
// ladder from range / 4: 212 -> 490 ms otherwise) and nothing cheap in the sizes or the value distribution tells them apart
// (|a - b| on a line has the same linear density of small cells as a random row).  What does: the cost vectors of two ROWS
// are uncorrelated in a random matrix (|r| ~ 1 / sqrt(samples)) and strongly correlated under any metric.  One workgroup per
// sampled row pair, 256 sampled columns, Pearson's r in double; the host averages |r| over the pairs.
template <typename CT>
__global__ __launch_bounds__(256) void k_row_corr(int n, int nchunks, const CT *__restrict__ cc, double *__restrict__ out)
{
    constexpr int E = Tr<CT>::E;
    __shared__ double s_acc[4][5];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t pitch = (size_t)nchunks * E;
    const uint32_t h1 = (blockIdx.x + 1u) * 0x9E3779B1u, h2 = (blockIdx.x + 1u) * 0x85EBCA6Bu + 0x27D4EB2Fu;
    const int i1 = (int)(((uint64_t)(h1 ^ (h1 >> 15)) * (uint64_t)n) >> 32);
    int i2 = (int)(((uint64_t)(h2 ^ (h2 >> 13)) * (uint64_t)n) >> 32);
    if (i2 == i1) i2 = (i1 + n / 2) % n;
    const int j = (int)(((long long)tid * n) >> 8);   // 256 evenly spread columns
    const double x = (double)(uint32_t)cc[(size_t)i1 * pitch + j], y = (double)(uint32_t)cc[(size_t)i2 * pitch + j];
    double a[5] = {x, y, x * y, x * x, y * y};
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o);
        if (lane == 0) s_acc[w][k] = a[k];
    }
    __syncthreads();
    if (tid == 0) {
        double t[5];
        for (int k = 0; k < 5; k++) t[k] = s_acc[0][k] + s_acc[1][k] + s_acc[2][k] + s_acc[3][k];
        const double m = 256.0, cov = t[2] - t[0] * t[1] / m, vx = t[3] - t[0] * t[0] / m, vy = t[4] - t[1] * t[1] / m;
        out[blockIdx.x] = (vx > 0 && vy > 0) ? fabs(cov) / sqrt(vx * vy) : 1.0;   // a constant row is not random
    }
}
