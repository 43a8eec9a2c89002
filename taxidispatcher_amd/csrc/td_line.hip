// Line-metric recogniser for td_assign.
//
// The reference's distance table is a line: greedy_opt.py:122-127 (and simulate.py / procedure.py the
// same way) fills dist[i][j] = |i - j|, and calculate_cost (greedy_opt.py:86-99) copies
// dist[cab.to][request.from] into the cost matrix.  A square matrix |a_i - b_j| sorted by a and by b is a
// Monge matrix: its optimal assignment is the sorted matching, and prices that prove it are a prefix sum
// of n adjacent differences.  The general solver (td_assign.hip) finds those instances hard when the
// stands are tie-free, so td_assign tries this first:
//
//   1. k_line_probe   one workgroup, O(n) reads: two columns p, q and two rows i1, i2 that lie at different
//                     positions, and a plausibility test (sampled rows against (p, q), all columns against
//                     (i1, i2): inside the anchors' span the two distances add up to D, outside they differ by D).
//                     The host reads the verdict: a matrix that is no line metric costs this kernel and one
//                     round trip, then the general solver runs.
//   2. k_line_keys    row key  c[i][p]^2 - c[i][q]^2 = (b_q - b_p)(2 a_i - b_p - b_q): strictly monotone
//                     in a_i for ANY two distinct columns; columns likewise from rows i1, i2, negated when
//                     K(i2) < K(i1) says the two orders run in opposite directions;
//   3. one radix sort of the 2n keys (hipCUB);
//   4. k_line_gather  row_to_col = the sorted matching, its total, the adjacent differences f[k], and the
//                     Monge test on the diagonal; k_line_scan: prices v[tau(k+1)] = v[tau(k)] + f[k];
//   5. k_line_cert    ONE streaming pass over the int32 matrix: for every row, min_j (c[i][j] - v[j]) must be
//                     attained on the row's matched cell.  Then sum_i c[i][m(i)] = sum_i min_j(..) + sum_j v[j]
//                     is a lower bound of every assignment: the matching is optimal, whatever the matrix was.
//
// Nothing here ASSUMES the structure: the answer is used only when step 5 accepts it on the actual matrix
// (exact 64-bit arithmetic), otherwise td_assign runs the general solver as before.  Step 5 is the only
// O(n^2) step and it is HBM-bound: 4 n^2 bytes read once.
#include <limits.h>

#include <hipcub/hipcub.hpp>

#include "td_common.h"

namespace td {
namespace {

// LC_PLAUS: 0 refused; 1 balanced line metric plausible; 2 constant trailing COLUMNS (retry on the transpose);
//           3 plausible with LC_K constant rows (the unbalanced model: fewer cabs than requests)
enum { LC_P = 0, LC_Q, LC_I1, LC_I2, LC_PLAUS, LC_REV, LC_FAIL, LC_FITS32, LC_TOTAL, LC_K, LC_FILL, LC_SKIP, LC_WORDS };
constexpr int LINE_KMAX = 256;  // most constant rows the unbalanced plan is made for: k prefix-min scans of n by one workgroup and O(k n) scratch (k = 256, n = 16 384: ~4 ms and 100 MB against > 400 ms for the general solver; 32 until round 3)

struct LineWs {
    Buf ctl, kin, kout, vin, vout, tmp, f, v64, v32, r2c;
    Buf band, P, E, ARG, Bk, L, mcol;   // the unbalanced plan (k_line_unbal)
};
LineWs g_lw;

typedef int v4i __attribute__((ext_vector_type(4)));

struct ArgMax {
    long long v;
    int i;
};
__device__ inline bool better(const ArgMax &a, const ArgMax &b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

// argmax over the block (ties: the smallest index); every thread gets the result
__device__ ArgMax block_argmax(ArgMax x, ArgMax *sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax y;
        y.v = __shfl_xor(x.v, o);
        y.i = __shfl_xor(x.i, o);
        if (better(y, x)) x = y;
    }
    __syncthreads();
    if (lane == 0) sh[w] = x;
    __syncthreads();
    ArgMax r = sh[0];
    for (int k = 1; k < nw; k++)
        if (better(sh[k], r)) r = sh[k];
    return r;
}

__device__ inline long long labs64(long long x) { return x < 0 ? -x : x; }

// (x, y) = distances of one point to two anchors that are D apart: inside the span x + y == D, outside |x - y| == D
__device__ inline bool span_ok(long long x, long long y, long long D) { return x + y == D || labs64(x - y) == D; }

// One workgroup, O(n) reads: anchors + a plausibility test, so that a matrix that is no line metric costs one
// small kernel and one host round trip (the verdict is written straight into the pinned host block).
//   0. the first 1024 cells of rows 0 and 1 must agree on ONE distance |a_0 - a_1| (inside the two rows' span the
//      distances add up to it, outside they differ by it): random matrices are refused here after two loads.
//      In the same round trip: is the last COLUMN constant over the sampled rows (model padded with dummy
//      requests: verdict 2, the host retries on the transpose), is the last ROW constant (dummy cabs)?
//   1. the same over the full rows; q = the column whose distance to row 0 differs most from column 0's
//      (columns p = 0 and q then lie at different positions; row n/2 is tried when row 0 sees no difference)
//   2. rows i1 = 0 and i2 = 1 if they lie at different positions, else the sampled row that differs most
//      (then all columns are tested against (i1, i2) again)
//   3. the sampled rows against (p, q) must agree on one distance |b_p - b_q|; constant rows are skipped and
//      counted (k of them: verdict 3 when 1 <= k <= LINE_KMAX)
//   rev: the column keys must be negated when the two key orders run in opposite directions
__global__ __launch_bounds__(1024) void k_line_probe(int n, const int32_t *__restrict__ c, long long *__restrict__ ctl,
                                                     long long *__restrict__ host_verdict)
{
    __shared__ ArgMax sh[16];
    __shared__ int s_cnt;
    const int t = threadIdx.x, T = blockDim.x;
    const int p = 0, i1 = 0;
    const int step = n > 1024 ? n / 1024 : 1;
    const int nsamp = (n + step - 1) / step;
    const long long u0 = c[0], w0 = c[n];
    const long long Ea = u0 + w0, Eb = labs64(u0 - w0);
    const long long B0 = c[(size_t)(n - 1) * n], B1 = c[n - 1];
    int mode = 0, rev = 0, q = -1, rq = 0, i2 = -1, kdummy = 0;
    int bad_a = 0, bad_b = 0, rowvar = 0, colvar = 0, far = 0;
    if (t == 0) s_cnt = 0;
    if (t < n) {
        const long long xx = c[t], yy = c[(size_t)n + t];
        bad_a = !span_ok(xx, yy, Ea);
        bad_b = !span_ok(xx, yy, Eb);
        rowvar = c[(size_t)(n - 1) * n + t] != B0;
        far = labs64(xx - u0) > 254;   // row 0 cannot be stored in one byte per cell
    }
    if (t < nsamp) colvar = c[(size_t)(t * step) * n + (n - 1)] != B1;
    bad_a = __syncthreads_or(bad_a);
    bad_b = __syncthreads_or(bad_b);
    const bool rows_dummy = !__syncthreads_or(rowvar);
    const bool cols_dummy = !__syncthreads_or(colvar);
    // a hint for td_assign's speculative 1-byte attempt: it will fail (row 0 is too wide) or is likely to ask for the
    // transpose (constant last column): worth one early look at its flag instead of ~30 launches that exit at once
    const int far_any = __syncthreads_or(far);
    const int suspicious = far_any || cols_dummy;
    // shape estimate for td_assign (only when the last column looks like a dummy request): how many of 1024 sampled
    // columns / rows are constant over 16 sampled cells — the same test as the shape probe of k_init_state, but
    // BEFORE any compress pass, so that a padded model goes straight to the transposed formulation
    int est_cc = 0, est_cr = 0;
    if (cols_dummy) {
        // 256 sampled columns / rows, 8 sampled cells each: the estimate only chooses between two exact formulations
        const int step4 = n > 256 ? n / 256 : 1;
        const int ns4 = (n + step4 - 1) / step4;
        int ccol = 0, crow = 0;
        if (t < ns4) {
            const int j = t * step4, i = t * step4;
            const int32_t v0 = c[j], w0r = c[(size_t)i * n];
            int same_c = 1, same_r = 1;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const size_t kk = (size_t)(((long long)k * n) >> 3);
                same_c &= c[kk * n + j] == v0;
                same_r &= c[(size_t)i * n + kk] == w0r;
            }
            ccol = same_c;
            crow = same_r;
        }
        est_cc = __syncthreads_count(ccol) * step4;
        est_cr = __syncthreads_count(crow) * step4;
    }
    if (cols_dummy) {
        // before the host pays for a transpose: columns 0 and 1 over the first 1024 ROWS must pass step 0 too
        // (a thresholded simulator model with dummy requests is refused here)
        // Against column 0 the test takes a column that some row sees at ANOTHER distance (two requests on one
        // stand are identical columns and would pass any such test): r = the first row with a real cell in
        // column 0, c1 = the first of columns 1..64 that differs from column 0 in row r.
        __shared__ int s_r, s_c1;
        if (t == 0) {
            s_r = INT_MAX;
            s_c1 = 1;
        }
        __syncthreads();
        if (t < n && c[(size_t)t * n] != B1) atomicMin(&s_r, t);
        __syncthreads();
        const int r = s_r;
        int bad_c = 0, bad_d = 0;
        if (r != INT_MAX) {
            if (t < 64) {
                const int j = 1 + t;
                const unsigned long long dm = __ballot(j < n && c[(size_t)r * n + j] != c[(size_t)r * n]);
                if (t == 0 && dm) s_c1 = 1 + (__ffsll((long long)dm) - 1);
            }
            __syncthreads();
            const int c1 = s_c1;
            const long long ur = c[(size_t)r * n], wr = c[(size_t)r * n + c1];
            const long long Ga = ur + wr, Gb = labs64(ur - wr);
            if (t < n) {
                const long long xx = c[(size_t)t * n], yy = c[(size_t)t * n + c1];
                bad_c = !span_ok(xx, yy, Ga);
                bad_d = !span_ok(xx, yy, Gb);
            }
            bad_c = __syncthreads_or(bad_c);
            bad_d = __syncthreads_or(bad_d);
        }
        const int real0 = r != INT_MAX;   // column 0 is a real request: not the fill value everywhere
        mode = (rows_dummy || (bad_c && bad_d) || !real0) ? 0 : 2;
    } else if (!(bad_a && bad_b)) {
        ArgMax x;
        x.v = LLONG_MIN, x.i = 0x7fffffff;
        bad_a = bad_b = 0;
        for (int j = t; j < n; j += T) {
            const long long xx = c[j], yy = c[(size_t)n + j];
            bad_a |= !span_ok(xx, yy, Ea);
            bad_b |= !span_ok(xx, yy, Eb);
            ArgMax y{labs64(xx - u0), j};
            if (better(y, x)) x = y;
        }
        bad_a = __syncthreads_or(bad_a);
        bad_b = __syncthreads_or(bad_b);
        ArgMax r = block_argmax(x, sh);
        if (r.v != 0) q = r.i;
        bool cols_ok = !(bad_a && bad_b);
        if (cols_ok && q < 0) {   // row 0 sees every column at one distance: ask row n/2 (a real row: dummies are at the end)
            rq = rows_dummy ? 1 : n / 2;
            const int32_t *row = c + (size_t)rq * n;
            const long long ref = row[p];
            x.v = LLONG_MIN, x.i = 0x7fffffff;
            for (int j = t; j < n; j += T) {
                ArgMax y{labs64((long long)row[j] - ref), j};
                if (better(y, x)) x = y;
            }
            r = block_argmax(x, sh);
            if (r.v != 0) q = r.i;
        }
        if (cols_ok && q >= 0) {
            if (w0 != u0 || c[(size_t)n + q] != c[q])
                i2 = 1;   // rows 0 and 1 lie at different positions: step 1 was the column test
            else {
                for (int attempt = 0; attempt < 2 && i2 < 0; attempt++) {
                    const int col = attempt == 0 ? p : q;
                    const long long ref = c[(size_t)i1 * n + col];
                    x.v = LLONG_MIN, x.i = 0x7fffffff;
                    for (int k = t; k < nsamp; k += T) {
                        const int i = k * step;
                        const bool dummy = rows_dummy && c[(size_t)i * n + p] == B0 && c[(size_t)i * n + q] == B0;
                        ArgMax y{dummy ? 0ll : labs64((long long)c[(size_t)i * n + col] - ref), i};
                        if (better(y, x)) x = y;
                    }
                    r = block_argmax(x, sh);
                    if (r.v != 0) i2 = r.i;
                }
                if (i2 >= 0) {
                    const long long uu = c[(size_t)i1 * n + p], ww = c[(size_t)i2 * n + p];
                    const long long Fa = uu + ww, Fb = labs64(uu - ww);
                    const int32_t *r1 = c + (size_t)i1 * n, *r2 = c + (size_t)i2 * n;
                    bad_a = bad_b = 0;
                    for (int j = t; j < n; j += T) {
                        const long long xx = r1[j], yy = r2[j];
                        bad_a |= !span_ok(xx, yy, Fa);
                        bad_b |= !span_ok(xx, yy, Fb);
                    }
                    bad_a = __syncthreads_or(bad_a);
                    bad_b = __syncthreads_or(bad_b);
                    cols_ok = !(bad_a && bad_b);
                }
            }
        }
        if (cols_ok && q >= 0 && i2 >= 0) {
            const long long x0 = c[(size_t)rq * n + p], y0 = c[(size_t)rq * n + q];
            const long long Da = x0 + y0, Db = labs64(x0 - y0);
            bad_a = bad_b = 0;
            for (int k = t; k < nsamp; k += T) {
                const int32_t *row = c + (size_t)(k * step) * n;
                const long long xx = row[p], yy = row[q];
                if (rows_dummy && xx == B0 && yy == B0) continue;
                bad_a |= !span_ok(xx, yy, Da);
                bad_b |= !span_ok(xx, yy, Db);
            }
            bad_a = __syncthreads_or(bad_a);
            bad_b = __syncthreads_or(bad_b);
            mode = !(bad_a && bad_b);
            if (mode && rows_dummy) {   // count the constant rows (anywhere, they are sorted to the end by their key)
                int cnt = 0;
                for (int i = t; i < n; i += T) cnt += c[(size_t)i * n + p] == B0 && c[(size_t)i * n + q] == B0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
                if ((t & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
                __syncthreads();
                kdummy = s_cnt;
                mode = (kdummy >= 1 && kdummy <= LINE_KMAX && n - kdummy >= 2) ? 3 : 0;
            }
            // K(i) = c[i][p]^2 - c[i][q]^2 = (b_q - b_p)(2 a_i - b_p - b_q);  K(i2) - K(i1) has the sign of (b_q - b_p)(a_i2 - a_i1)
            const long long a1 = c[(size_t)i1 * n + p], b1 = c[(size_t)i1 * n + q], a2 = c[(size_t)i2 * n + p], b2 = c[(size_t)i2 * n + q];
            rev = (a2 * a2 - b2 * b2) < (a1 * a1 - b1 * b1);
        }
    }
    if (t == 0) {
        ctl[LC_P] = p;
        ctl[LC_Q] = q < 0 ? 0 : q;
        ctl[LC_I1] = i1;
        ctl[LC_I2] = i2 < 0 ? 0 : i2;
        ctl[LC_PLAUS] = mode;
        ctl[LC_REV] = rev;
        ctl[LC_FAIL] = mode == 0;
        ctl[LC_FITS32] = 0;
        ctl[LC_TOTAL] = 0;
        ctl[LC_K] = kdummy;
        ctl[LC_FILL] = B0;
        ctl[LC_SKIP] = (mode != 0 || suspicious) ? 1 : 0;   // the speculative 1-byte compress pass queued behind this probe is void either way
        host_verdict[3] = est_cc;
        host_verdict[4] = est_cr;
        host_verdict[5] = far_any;
        host_verdict[6] = B1;   // the constant last column's value: the fill value of a padded model
        host_verdict[0] = mode;
        host_verdict[1] = kdummy;
        host_verdict[2] = suspicious;
        __threadfence_system();
    }
}

// sort keys: bit 63 = side (0 rows, 1 columns), bits 0..62 = key + 2^62 (|key| < 2^62 for int32 cells)
__global__ __launch_bounds__(256) void k_line_keys(int n, const int32_t *__restrict__ c, const long long *__restrict__ ctl,
                                                   unsigned long long *__restrict__ keys, int *__restrict__ vals)
{
    const int p = (int)ctl[LC_P], q = (int)ctl[LC_Q], i1 = (int)ctl[LC_I1], i2 = (int)ctl[LC_I2];
    const bool rev = ctl[LC_REV] != 0;
    const bool unbal = ctl[LC_PLAUS] == 3;
    const long long fill = ctl[LC_FILL];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += gridDim.x * blockDim.x) {
        long long x, y;
        if (t < n) {
            x = c[(size_t)t * n + p];
            y = c[(size_t)t * n + q];
            if (unbal && x == fill && y == fill) {   // a constant row: behind every real row
                keys[t] = 0x7fffffffffffffffull;
                vals[t] = t;
                continue;
            }
        } else {
            x = c[(size_t)i1 * n + (t - n)];
            y = c[(size_t)i2 * n + (t - n)];
        }
        long long key = x * x - y * y;
        if (t >= n && rev) key = -key;
        keys[t] = (unsigned long long)(key + (1ll << 62)) | (t < n ? 0ull : 1ull << 63);
        vals[t] = t < n ? t : t - n;
    }
}

// sorted[0..n) = rows by key (s), sorted[n..2n) = columns by key (tau): row_to_col, the adjacent differences
//   f[k] = c[s(k)][tau(k+1)] - c[s(k)][tau(k)]   (price step)      b[k] = c[s(k+1)][tau(k)] - c[s(k+1)][tau(k+1)]
// (f + b < 0 is a Monge violation on the diagonal: no prices can make both matched cells row minima), and the total.
__global__ __launch_bounds__(256) void k_line_gather(int n, const int32_t *__restrict__ c, const int *__restrict__ sorted,
                                                     long long *__restrict__ ctl, long long *__restrict__ f, int *__restrict__ r2c)
{
    __shared__ long long sh[4];
    const int *sig = sorted, *tau = sorted + n;
    long long sum = 0;
    int bad = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int s = sig[k], tk = tau[k];
        const int32_t *row = c + (size_t)s * n;
        const long long ckk = row[tk];
        long long d = 0;
        if (k + 1 < n) {
            const int tk1 = tau[k + 1];
            const int32_t *row1 = c + (size_t)sig[k + 1] * n;
            d = (long long)row[tk1] - ckk;
            const long long b = (long long)row1[tk] - (long long)row1[tk1];
            bad |= d + b < 0;
        }
        f[k] = d;
        r2c[s] = tk;
        sum += ckk;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        sum = sh[0] + sh[1] + sh[2] + sh[3];
        if (sum) atomicAdd((unsigned long long *)&ctl[LC_TOTAL], (unsigned long long)sum);
    }
    if (bad) ctl[LC_FAIL] = 1;
}

// one workgroup: prices = exclusive prefix sums of f along the sorted columns (CH tiles of 1024 loaded up front,
// so the serial chain of tile scans does not wait on memory)
__global__ __launch_bounds__(1024) void k_line_scan(int n, const int *__restrict__ sorted, long long *__restrict__ ctl,
                                                    const long long *__restrict__ f, long long *__restrict__ v64,
                                                    int32_t *__restrict__ v32)
{
    constexpr int CH = 16;
    __shared__ long long wsum[2][16];
    __shared__ int wide;
    const int t = threadIdx.x, T = blockDim.x, lane = t & 63, w = t >> 6, nw = T >> 6;
    const int *tau = sorted + n;
    if (t == 0) wide = 0;
    long long carry = 0;
    bool out = false;
    int par = 0;
    for (int base = 0; base < n; base += T * CH) {
        long long x[CH];
        int col[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int k = base + u * T + t;
            x[u] = k < n ? f[k] : 0;
            col[u] = k < n ? tau[k] : -1;
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            if (base + u * T >= n) break;
            long long inc = x[u];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const long long y = __shfl_up(inc, o);
                if (lane >= o) inc += y;
            }
            if (lane == 63) wsum[par][w] = inc;
            __syncthreads();   // (the other wsum buffer is written next: one barrier per tile)
            long long before = carry, all = 0;
            for (int j = 0; j < nw; j++) {
                const long long ws = wsum[par][j];
                if (j < w) before += ws;
                all += ws;
            }
            par ^= 1;
            const long long price = before + inc - x[u];   // exclusive
            if (col[u] >= 0) {
                v64[col[u]] = price;
                v32[col[u]] = (int32_t)price;
                out |= price < -(1ll << 30) || price > (1ll << 30);
            }
            carry += all;
        }
    }
    if (out) wide = 1;
    __syncthreads();
    if (t == 0) ctl[LC_FITS32] = !wide;
}

// -------------------------------------------------------------------------------------
// The unbalanced model (greedy_opt.py:88-90: n = max(cabs, requests), the missing cabs are rows of big_cost):
// m real rows, n = m + k columns, k constant rows.  The real rows take m of the n columns in sorted order
// (an optimal matching of points on a line does not cross), i.e. row i of the sorted order takes sorted column
// i + shift(i) with a non-decreasing shift in 0..k.  With P_d[t] = sum_{i<t} c[s(i)][tau(i+d)]:
//     E_0[t] = P_0[t],   E_d[t] = P_d[t] + min_{t' <= t} (E_{d-1}[t'] - P_d[t'])        (rows < t' keep a shift < d)
// is k prefix-min scans; the optimum is E_k[m] and the argmins give the rows t_1 <= ... <= t_k where the shift
// steps up.  Step d skips sorted column t_d + d - 1: those k columns go to the constant rows at price 0, the
// matched columns get  v = min(L, R) <= 0,  L_i = min(0, L_{i-1} + f_{i-1}),  R_i = min(0, R_{i+1} + b_i)
// (the cheapest alternating walk that ends in the column, walks may start anywhere because a skipped column
// reaches every column at no cost).  As in the balanced case nothing is taken on trust: k_line_cert checks
// the n rows of the actual matrix against these prices.
// -------------------------------------------------------------------------------------
struct P2 {
    long long a, b;
};
struct OpSum {   // a = running sum
    static __device__ P2 id() { return {0, 0}; }
    static __device__ P2 f(P2 x, P2 y) { return {x.a + y.a, 0}; }
};
struct OpMinArg {   // a = value, b = index; the EARLIER element wins a tie
    static __device__ P2 id() { return {LLONG_MAX, 0}; }
    static __device__ P2 f(P2 x, P2 y) { return y.a < x.a ? y : x; }
};
struct OpClamp {   // the map z -> min(a, z + b); f(x, y) = "x, then y"
    static __device__ P2 id() { return {LLONG_MAX / 4, 0}; }
    static __device__ P2 f(P2 x, P2 y) { return {y.a < x.a + y.b ? y.a : x.a + y.b, x.b + y.b}; }
};

// inclusive scan by ONE workgroup of 1024 threads over `len` elements (load(idx) -> store(idx, prefix)), front to
// back or back to front; CH tiles are loaded before the serial chain of tile scans starts
template <typename Op, typename Load, typename Store>
__device__ void wg_scan(int len, bool reverse, Load load, Store store, P2 *wsum)
{
    constexpr int CH = 8;
    const int t = threadIdx.x, T = blockDim.x, lane = t & 63, w = t >> 6, nw = T >> 6;
    P2 carry = Op::id();
    int par = 0;
    for (int base = 0; base < len; base += T * CH) {
        P2 x[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int pos = base + u * T + t;
            x[u] = pos < len ? load(reverse ? len - 1 - pos : pos) : Op::id();
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            if (base + u * T >= len) break;
            P2 inc = x[u];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                P2 y;
                y.a = __shfl_up(inc.a, o);
                y.b = __shfl_up(inc.b, o);
                if (lane >= o) inc = Op::f(y, inc);
            }
            if (lane == 63) wsum[par * 16 + w] = inc;
            __syncthreads();
            P2 before = carry, tot = carry;
            for (int j = 0; j < nw; j++) {
                const P2 ws = wsum[par * 16 + j];
                if (j < w) before = Op::f(before, ws);
                tot = Op::f(tot, ws);
            }
            par ^= 1;
            const int pos = base + u * T + t;
            if (pos < len) store(reverse ? len - 1 - pos : pos, Op::f(before, inc));
            carry = tot;
        }
    }
    __syncthreads();
}

// band[d][i] = c[s(i)][tau(i + d)], i < m, d <= k
__global__ __launch_bounds__(256) void k_line_band(int n, int k, const int32_t *__restrict__ c, const int *__restrict__ sorted,
                                                   long long *__restrict__ band)
{
    const int m = n - k;
    const int *sig = sorted, *tau = sorted + n;
    const long long cnt = (long long)(k + 1) * m;
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < cnt; x += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(x / m), i = (int)(x - (long long)d * m);
        band[x] = c[(size_t)sig[i] * n + tau[i + d]];
    }
}

// one workgroup: the plan above. Scratch (all long long): P (k+1) x (m+1), E m+1, ARG k x (m+1), F / B / L m each.
__global__ __launch_bounds__(1024) void k_line_unbal(int n, int k, const int32_t *__restrict__ c, const int *__restrict__ sorted,
                                                     long long *__restrict__ ctl, const long long *__restrict__ band,
                                                     long long *__restrict__ P, long long *__restrict__ E, long long *__restrict__ ARG,
                                                     long long *__restrict__ F, long long *__restrict__ Bk, long long *__restrict__ L,
                                                     int *__restrict__ mcol, long long *__restrict__ v64, int32_t *__restrict__ v32,
                                                     int *__restrict__ r2c)
{
    __shared__ P2 wsum[32];
    __shared__ int bounds[LINE_KMAX + 2];
    __shared__ long long s_tot[16];
    __shared__ int wide;
    const int t = threadIdx.x, T = blockDim.x;
    const int m = n - k, m1 = m + 1;
    const int *sig = sorted, *tau = sorted + n;
    if (t == 0) wide = 0;
    // P_d[t+1] = inclusive prefix sums of the band
    for (int d = 0; d <= k; d++) {
        long long *Pd = P + (size_t)d * m1;
        const long long *bd = band + (size_t)d * m;
        if (t == 0) Pd[0] = 0;
        wg_scan<OpSum>(m, false, [&](int i) { return P2{bd[i], 0}; }, [&](int i, P2 r) { Pd[i + 1] = r.a; }, wsum);
    }
    for (int i = t; i < m1; i += T) E[i] = P[i];
    __syncthreads();
    for (int d = 1; d <= k; d++) {
        const long long *Pd = P + (size_t)d * m1;
        long long *Ad = ARG + (size_t)(d - 1) * m1;
        wg_scan<OpMinArg>(m1, false, [&](int i) { return P2{E[i] - Pd[i], (long long)i}; },
                          [&](int i, P2 r) {
                              E[i] = Pd[i] + r.a;
                              Ad[i] = r.b;
                          },
                          wsum);
    }
    if (t == 0) {
        int pos = m;
        bounds[k + 1] = m;
        for (int d = k; d >= 1; d--) {
            pos = (int)ARG[(size_t)(d - 1) * m1 + pos];
            bounds[d] = pos;
        }
        bounds[0] = 0;
    }
    __syncthreads();
    // rows: shift, matched column, the adjacent differences along the matched chain
    long long sum = 0;
    for (int i = t; i < m; i += T) {
        int sh = 0;
        for (int d = 1; d <= k; d++) sh += bounds[d] <= i;
        mcol[i] = i + sh;
    }
    __syncthreads();
    for (int i = t; i < n; i += T) {
        const int row = sig[i];
        int col;
        if (i < m)
            col = tau[mcol[i]];
        else
            col = tau[bounds[i - m + 1] + (i - m)];   // constant row number d-1 takes the column skipped by step d
        r2c[row] = col;
        const int32_t *rp = c + (size_t)row * n;
        sum += rp[col];
        if (i < m) {
            long long f = 0, b = 0;
            if (i + 1 < m) {
                const int cn = tau[mcol[i + 1]];
                const int32_t *rn = c + (size_t)sig[i + 1] * n;
                f = (long long)rp[cn] - (long long)rp[col];
                b = (long long)rn[col] - (long long)rn[cn];
            }
            F[i] = f;
            Bk[i] = b;
        } else {
            v64[col] = 0;
            v32[col] = 0;
        }
    }
    __syncthreads();
    // L_0 = 0, L_{i+1} = min(0, L_i + F_i): inclusive scan of the clamp maps, applied to 0
    if (t == 0) L[0] = 0;
    wg_scan<OpClamp>(m - 1, false, [&](int i) { return P2{0, F[i]}; },
                     [&](int i, P2 r) { L[i + 1] = r.a < r.b ? r.a : r.b; }, wsum);
    // R_{m-1} = 0, R_i = min(0, R_{i+1} + B_i): the same from the back; price = min(L, R)
    if (t == 0) {
        const int col = tau[mcol[m - 1]];
        const long long pr = L[m - 1] < 0 ? L[m - 1] : 0;
        v64[col] = pr;
        v32[col] = (int32_t)pr;
        if (pr < -(1ll << 30)) wide = 1;
    }
    wg_scan<OpClamp>(m - 1, true, [&](int i) { return P2{0, Bk[i]}; },
                     [&](int i, P2 r) {
                         const long long R = r.a < r.b ? r.a : r.b;
                         const long long pr = R < L[i] ? R : L[i];
                         const int col = tau[mcol[i]];
                         v64[col] = pr;
                         v32[col] = (int32_t)pr;
                         if (pr < -(1ll << 30)) wide = 1;
                     },
                     wsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((t & 63) == 0) s_tot[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        long long tot = 0;
        for (int j = 0; j < (T >> 6); j++) tot += s_tot[j];
        ctl[LC_TOTAL] = tot;
        ctl[LC_FITS32] = !wide;
    }
}

// The certificate: row i passes when no cell of the row beats its matched cell, min_j (c[i][j] - v[j]) == c[i][m] - v[m].
// R rows per workgroup sweep so a price vector chunk is loaded once per R cost chunks; 16-byte loads.
// (nrows rows of n cells: the whole matrix, or one rank's row shard with r2c pointing at its first row's entry)
template <typename VT, int R, bool VEC>
__device__ void line_cert_body(int n, int nrows, const int32_t *__restrict__ c, const VT *__restrict__ v, const int *__restrict__ r2c,
                               long long *__restrict__ ctl, long long *sh)
{
    const int t = threadIdx.x, T = blockDim.x, lane = t & 63, w = t >> 6, nw = T >> 6;
    const int ngroups = (nrows + R - 1) / R;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int r0 = g * R;
        long long m[R];
#pragma unroll
        for (int r = 0; r < R; r++) m[r] = LLONG_MAX;
        if (VEC) {
            const int nq = n >> 2;
            for (int ch = t; ch < nq; ch += T) {
                long long pv[4];
                if (sizeof(VT) == 4) {
                    const int4 x = *reinterpret_cast<const int4 *>((const int32_t *)v + 4 * (size_t)ch);
                    pv[0] = x.x, pv[1] = x.y, pv[2] = x.z, pv[3] = x.w;
                } else {
                    const longlong2 x = *reinterpret_cast<const longlong2 *>((const long long *)v + 4 * (size_t)ch);
                    const longlong2 y = *reinterpret_cast<const longlong2 *>((const long long *)v + 4 * (size_t)ch + 2);
                    pv[0] = x.x, pv[1] = x.y, pv[2] = y.x, pv[3] = y.y;
                }
                v4i cv[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int row = std::min(r0 + r, nrows - 1);
                    cv[r] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(c + (size_t)row * n) + ch);
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    long long a = (long long)cv[r].x - pv[0], b = (long long)cv[r].y - pv[1];
                    long long d = (long long)cv[r].z - pv[2], e = (long long)cv[r].w - pv[3];
                    a = a < b ? a : b;
                    d = d < e ? d : e;
                    a = a < d ? a : d;
                    m[r] = a < m[r] ? a : m[r];
                }
            }
        } else {
            for (int j = t; j < n; j += T) {
                const long long pv = (long long)v[j];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int row = std::min(r0 + r, nrows - 1);
                    const long long a = (long long)c[(size_t)row * n + j] - pv;
                    m[r] = a < m[r] ? a : m[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const long long y = __shfl_xor(m[r], o);
                m[r] = y < m[r] ? y : m[r];
            }
        }
        __syncthreads();
        if (lane == 0)
            for (int r = 0; r < R; r++) sh[r * 16 + w] = m[r];
        __syncthreads();
        if (t < R && r0 + t < nrows) {
            long long mm = sh[t * 16];
            for (int k = 1; k < nw; k++) mm = sh[t * 16 + k] < mm ? sh[t * 16 + k] : mm;
            const int row = r0 + t, col = r2c[row];
            const long long tight = (long long)c[(size_t)row * n + col] - (long long)v[col];
            if (mm != tight) ctl[LC_FAIL] = 1;
        }
    }
}

template <int R, bool VEC>
__global__ __launch_bounds__(256) void k_line_cert(int n, int nrows, const int32_t *__restrict__ c, const long long *__restrict__ v64,
                                                   const int32_t *__restrict__ v32, const int *__restrict__ r2c,
                                                   long long *__restrict__ ctl)
{
    __shared__ long long sh[R * 16];
    if (ctl[LC_FAIL]) return;   // the gather pass already found a violation
    if (ctl[LC_FITS32])
        line_cert_body<int32_t, R, VEC>(n, nrows, c, v32, r2c, ctl, sh);
    else
        line_cert_body<long long, R, VEC>(n, nrows, c, v64, r2c, ctl, sh);
}

// -------------------------------------------------------------------------------------
// The same plan over ROW SHARDS (one rank = rows [row0, row0 + nrows) x all n columns): everything O(n) is
// replicated, the two O(rows) reads of the matrix (keys, differences) and the O(n^2 / world) certificate pass are
// local, and the ranks meet in four SUM all-reduces of disjointly written, otherwise zero segments of one
// workspace (so a sum IS the gather / broadcast, whatever the backend).  Workspace layout, 64-bit words,
// N2 = n rounded up to 8:
//   [0, 16)            ctl, indexed like the single-GPU path (LC_*)            | segment 0: written by the rank that
//   [16, 16 + N2)      rows i1 = 0 and i2 of the matrix as int32 (2 * N2)      | owns row 0
//   RK  n words        row keys, each rank its own rows                           segment 1
//   FB  2 N2 + 1       f[k], b[k] by sorted position, + the matching's total      segment 2
//   FAIL 1 word        certificate verdicts                                       segment 3
//   then the replicated working set: sorted (2 N2 int), inv, r2c (N2 int each), v64, v32, the 2n sort pairs
// -------------------------------------------------------------------------------------
struct LshLayout {
    size_t N2, ctl, anch, rk, fb, tot, failw, sorted, inv, r2c, v64, v32, kin, kout, vin, end;
    explicit LshLayout(int n)
    {
        N2 = ((size_t)n + 7) & ~(size_t)7;
        ctl = 0, anch = 16, rk = anch + N2, fb = rk + N2, tot = fb + 2 * N2, failw = tot + 1;
        sorted = (failw + 1 + 7) & ~(size_t)7;
        inv = sorted + N2, r2c = inv + N2 / 2, v64 = r2c + N2 / 2, v32 = v64 + N2, kin = v32 + N2 / 2, kout = kin + 2 * N2;
        vin = kout + 2 * N2, end = vin + N2;
    }
};

// the rank that owns row 0: anchors without a plausibility test (a wrong guess costs a refused certificate):
// p = 0, q = the column whose distance to row 0 differs most from column 0's, i1 = 0, i2 = the local row that
// differs most from row 0 in columns p and q
__global__ __launch_bounds__(1024) void k_lsh_anchor(int n, int nrows, const int32_t *__restrict__ c, long long *__restrict__ ctl,
                                                     int32_t *__restrict__ anch, size_t N2)
{
    __shared__ ArgMax sh[16];
    const int t = threadIdx.x, T = blockDim.x;
    if (nrows >= 2) {
        // FIRST the probe's first test (k_line_probe step 0), so that a matrix that is no line metric costs a few loads and
        // every rank ONE exchange: the first 1024 cells of rows 0 and 1 must agree on one distance |a_0 - a_1| (inside the
        // two rows' span the distances add up to it, outside they differ by it)
        const int m = n < 1024 ? n : 1024;
        long long x = 0, y = 0;
        ArgMax da = {-1, 0}, db = {LLONG_MIN, 0};
        if (t < m) {
            x = c[t], y = c[(size_t)n + t];
            da.v = labs64(x - y), da.i = t;
            db.v = -(x + y), db.i = t;
        }
        const long long Da = block_argmax(da, sh).v, Db = -block_argmax(db, sh).v;
        const int bad_a = __syncthreads_or(t < m && !span_ok(x, y, Da));
        const int bad_b = __syncthreads_or(t < m && !span_ok(x, y, Db));
        if (bad_a && bad_b) return;   // ctl stays zero: LC_PLAUS = 0 (the workspace was cleared in front of this kernel)
    }
    const long long c00 = c[0];
    ArgMax best = {-1, 0};
    for (int j = t; j < n; j += T) {
        const ArgMax x = {labs64((long long)c[j] - c00), j};
        if (better(x, best)) best = x;
    }
    best = block_argmax(best, sh);
    const int q = best.i;
    const bool okq = best.v > 0;
    const long long c0q = c[q];
    ArgMax b2 = {-1, 0};
    for (int i = t; i < nrows; i += T) {
        const ArgMax x = {labs64((long long)c[(size_t)i * n] - c00) + labs64((long long)c[(size_t)i * n + q] - c0q), i};
        if (better(x, b2)) b2 = x;
    }
    b2 = block_argmax(b2, sh);
    const int i2 = b2.i;
    const bool ok = okq && b2.v > 0;
    for (int j = t; j < n; j += T) {
        anch[j] = c[j];
        anch[N2 + j] = c[(size_t)i2 * n + j];
    }
    if (t == 0) {
        const long long a1 = c00, b1 = c0q, a2 = c[(size_t)i2 * n], b2v = c[(size_t)i2 * n + q];
        ctl[LC_P] = 0;
        ctl[LC_Q] = q;
        ctl[LC_I1] = 0;
        ctl[LC_I2] = i2;
        ctl[LC_PLAUS] = ok;
        ctl[LC_REV] = (a2 * a2 - b2v * b2v) < (a1 * a1 - b1 * b1);
    }
}

__global__ __launch_bounds__(256) void k_lsh_rowkeys(int n, int row0, int nrows, const int32_t *__restrict__ c,
                                                     const long long *__restrict__ ctl, long long *__restrict__ rk)
{
    const int q = (int)ctl[LC_Q];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nrows; t += gridDim.x * blockDim.x) {
        const long long x = c[(size_t)t * n], y = c[(size_t)t * n + q];
        rk[row0 + t] = x * x - y * y + (1ll << 62);
    }
}

__global__ __launch_bounds__(256) void k_lsh_keys(int n, const long long *__restrict__ ctl, const int32_t *__restrict__ anch, size_t N2,
                                                  const long long *__restrict__ rk, unsigned long long *__restrict__ keys,
                                                  int *__restrict__ vals)
{
    const bool rev = ctl[LC_REV] != 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += gridDim.x * blockDim.x) {
        if (t < n) {
            keys[t] = (unsigned long long)rk[t];
            vals[t] = t;
        } else {
            const long long x = anch[t - n], y = anch[N2 + t - n];
            long long key = x * x - y * y;
            if (rev) key = -key;
            keys[t] = (unsigned long long)(key + (1ll << 62)) | 1ull << 63;
            vals[t] = t - n;
        }
    }
}

__global__ __launch_bounds__(256) void k_lsh_inv(int n, const int *__restrict__ sorted, int *__restrict__ inv, int *__restrict__ r2c)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        inv[sorted[k]] = k;
        r2c[sorted[k]] = sorted[n + k];
    }
}

// each local row's matched cell and its two neighbours along the sorted columns: f[k] and b[k - 1] of k_line_gather
__global__ __launch_bounds__(256) void k_lsh_diffs(int n, int row0, int nrows, const int32_t *__restrict__ c, const int *__restrict__ sorted,
                                                   const int *__restrict__ inv, long long *__restrict__ F, long long *__restrict__ B,
                                                   long long *__restrict__ tot)
{
    __shared__ long long sh[4];
    const int *tau = sorted + n;
    long long sum = 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nrows; t += gridDim.x * blockDim.x) {
        const int k = inv[row0 + t];
        const int32_t *row = c + (size_t)t * n;
        const long long ckk = row[tau[k]];
        if (k + 1 < n) F[k] = (long long)row[tau[k + 1]] - ckk;
        if (k > 0) B[k - 1] = (long long)row[tau[k - 1]] - ckk;
        sum += ckk;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        sum = sh[0] + sh[1] + sh[2] + sh[3];
        if (sum) atomicAdd((unsigned long long *)tot, (unsigned long long)sum);
    }
}

__global__ __launch_bounds__(256) void k_lsh_monge(int n, const long long *__restrict__ F, const long long *__restrict__ B,
                                                   long long *__restrict__ ctl)
{
    int bad = ctl[LC_PLAUS] == 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k + 1 < n; k += gridDim.x * blockDim.x) bad |= F[k] + B[k] < 0;
    if (bad) ctl[LC_FAIL] = 1;
}

__global__ void k_lsh_publish(const long long *__restrict__ ctl, long long *__restrict__ failw) { *failw = ctl[LC_FAIL] != 0; }

}  // namespace

constexpr size_t VERDICT_OFF = 4096;   // byte offset of the probe's verdict in the pinned host block (clear of the other read-backs)
hipEvent_t g_probe_done = nullptr;

int line_probe_launch(int n, const int32_t *d_cost, const long long **skip_dev)
{
    Ctx &c = ctx();
    int rc;
    if ((rc = ensure(g_lw.ctl, 256))) return rc;
    if (!g_probe_done) TD_HIP(hipEventCreateWithFlags(&g_probe_done, hipEventDisableTiming));
    long long *ctl = (long long *)g_lw.ctl.p;
    {
        ProfScope ps(TD_K_LINE);
        k_line_probe<<<1, 1024, 0, c.stream>>>(n, d_cost, ctl, (long long *)((char *)c.pinned + VERDICT_OFF));
        TD_HIP(hipGetLastError());
    }
    TD_HIP(hipEventRecord(g_probe_done, c.stream));
    *skip_dev = ctl + LC_SKIP;
    return TD_OK;
}

int line_probe_wait(int *mode, int *k, int *suspicious, int *shape3)
{
    Ctx &c = ctx();
    TD_HIP(hipEventSynchronize(g_probe_done));
    const volatile long long *h = (const volatile long long *)((char *)c.pinned + VERDICT_OFF);
    *mode = (int)h[0];
    *k = (int)h[1];
    if (suspicious) *suspicious = (int)h[2];
    if (shape3) {   // estimated constant columns, estimated constant rows, row 0 too wide for one byte per cell, the last column's value
        shape3[0] = (int)h[3];
        shape3[1] = (int)h[4];
        shape3[2] = (int)h[5];
        shape3[3] = (int)h[6];
    }
    return TD_OK;
}

// The sorted matching on the n x n device matrix (after a "plausible" probe; k > 0: the last k rows of the sorted
// order are constant rows, verdict 3). *accepted = 1: *r2c_dev points at the library's device copy of row_to_col
// and *total is its cost, proven optimal by the certificate pass.
int line_finish(int n, int k, const int32_t *d_cost, const int32_t **r2c_dev, int64_t *total, int *accepted)
{
    Ctx &c = ctx();
    *accepted = 0;
    int rc;
    const size_t np = (size_t)n + 16;
    if ((rc = ensure(g_lw.kin, 16 * np))) return rc;
    if ((rc = ensure(g_lw.kout, 16 * np))) return rc;
    if ((rc = ensure(g_lw.vin, 8 * np))) return rc;
    if ((rc = ensure(g_lw.vout, 8 * np))) return rc;
    if ((rc = ensure(g_lw.f, 8 * np))) return rc;
    if ((rc = ensure(g_lw.v64, 8 * np))) return rc;
    if ((rc = ensure(g_lw.v32, 4 * np))) return rc;
    if ((rc = ensure(g_lw.r2c, 4 * np))) return rc;
    long long *ctl = (long long *)g_lw.ctl.p;
    auto *kin = (unsigned long long *)g_lw.kin.p, *kout = (unsigned long long *)g_lw.kout.p;
    int *vin = (int *)g_lw.vin.p, *vout = (int *)g_lw.vout.p;
    {
        ProfScope ps(TD_K_LINE);
        k_line_keys<<<std::max(1, std::min(c.n_cu * 4, (2 * n + 255) / 256)), 256, 0, c.stream>>>(n, d_cost, ctl, kin, vin);
        size_t bytes = 0;
        TD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, 2 * n, 0, 64, c.stream));
        if ((rc = ensure(g_lw.tmp, bytes + 256))) return rc;
        TD_HIP(hipcub::DeviceRadixSort::SortPairs(g_lw.tmp.p, bytes, kin, kout, vin, vout, 2 * n, 0, 64, c.stream));
        if (k == 0) {
            k_line_gather<<<std::max(1, std::min(c.n_cu * 4, (n + 255) / 256)), 256, 0, c.stream>>>(n, d_cost, vout, ctl, (long long *)g_lw.f.p,
                                                                                                (int *)g_lw.r2c.p);
            k_line_scan<<<1, 1024, 0, c.stream>>>(n, vout, ctl, (const long long *)g_lw.f.p, (long long *)g_lw.v64.p, (int32_t *)g_lw.v32.p);
        } else {
            const size_t n1 = (size_t)n + 1;
            if ((rc = ensure(g_lw.band, 8 * (size_t)(k + 1) * n1))) return rc;
            if ((rc = ensure(g_lw.P, 8 * (size_t)(k + 1) * n1))) return rc;
            if ((rc = ensure(g_lw.ARG, 8 * (size_t)k * n1))) return rc;
            if ((rc = ensure(g_lw.E, 8 * n1))) return rc;
            if ((rc = ensure(g_lw.Bk, 8 * n1))) return rc;
            if ((rc = ensure(g_lw.L, 8 * n1))) return rc;
            if ((rc = ensure(g_lw.mcol, 4 * n1))) return rc;
            k_line_band<<<std::max(1, std::min(c.n_cu * 4, (int)(((size_t)(k + 1) * n + 255) / 256))), 256, 0, c.stream>>>(
                n, k, d_cost, vout, (long long *)g_lw.band.p);
            k_line_unbal<<<1, 1024, 0, c.stream>>>(n, k, d_cost, vout, ctl, (const long long *)g_lw.band.p, (long long *)g_lw.P.p,
                                                    (long long *)g_lw.E.p, (long long *)g_lw.ARG.p, (long long *)g_lw.f.p,
                                                    (long long *)g_lw.Bk.p, (long long *)g_lw.L.p, (int *)g_lw.mcol.p,
                                                    (long long *)g_lw.v64.p, (int32_t *)g_lw.v32.p, (int *)g_lw.r2c.p);
        }
        TD_HIP(hipGetLastError());
    }
    {
        ProfScope ps(TD_K_CERT);
        constexpr int R = 4;
        const int ngroups = (n + R - 1) / R;
        const int grid = std::max(1, std::min(ngroups, c.n_cu * 16));
        const bool vec = (n % 4 == 0) && (((uintptr_t)d_cost & 15) == 0);
        if (vec)
            k_line_cert<R, true><<<grid, 256, 0, c.stream>>>(n, n, d_cost, (const long long *)g_lw.v64.p, (const int32_t *)g_lw.v32.p,
                                                             (const int *)g_lw.r2c.p, ctl);
        else
            k_line_cert<R, false><<<grid, 256, 0, c.stream>>>(n, n, d_cost, (const long long *)g_lw.v64.p, (const int32_t *)g_lw.v32.p,
                                                              (const int *)g_lw.r2c.p, ctl);
        TD_HIP(hipGetLastError());
    }
    TD_HIP(hipMemcpyAsync(c.pinned, ctl, LC_WORDS * sizeof(long long), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    const long long *h = (const long long *)c.pinned;
    if (h[LC_FAIL] == 0) {
        *accepted = 1;
        *total = h[LC_TOTAL];
        *r2c_dev = (const int32_t *)g_lw.r2c.p;
    }
    return TD_OK;
}

void line_release_workspace()
{
    if (g_probe_done) (void)hipEventDestroy(g_probe_done);
    g_probe_done = nullptr;
    Buf *bs[] = {&g_lw.ctl, &g_lw.kin,  &g_lw.kout, &g_lw.vin, &g_lw.vout, &g_lw.tmp, &g_lw.f,  &g_lw.v64, &g_lw.v32,
                 &g_lw.r2c, &g_lw.band, &g_lw.P,    &g_lw.E,   &g_lw.ARG,  &g_lw.Bk,  &g_lw.L,  &g_lw.mcol};
    for (Buf *b : bs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
}

}  // namespace td

using namespace td;

// ---- C ABI: the sorted matching over row shards (see the layout above k_lsh_anchor) ----
extern "C" int64_t td_line_shard_ws_words(int n) { return n < 2 ? 0 : (int64_t)LshLayout(n).end; }

extern "C" int td_line_shard_phase(int phase, int n, int row0, int nrows, const int32_t *cost_rows, int64_t *ws, int64_t *seg_off,
                                   int64_t *seg_len)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n < 2 || row0 < 0 || nrows < 0 || (long long)row0 + nrows > n) return fail(TD_EINVAL, "td_line_shard_phase: bad shard %d+%d of %d", row0, nrows, n);
    if (!ws || !seg_off || !seg_len) return fail(TD_EINVAL, "td_line_shard_phase: null argument");
    if (nrows > 0 && (!cost_rows || !is_device_ptr(cost_rows))) return fail(TD_EINVAL, "td_line_shard_phase: cost_rows must be device memory");
    if (!is_device_ptr(ws)) return fail(TD_EINVAL, "td_line_shard_phase: ws must be device memory");
    const LshLayout L(n);
    long long *w = (long long *)ws, *ctl = w + L.ctl;
    const int grid_rows = std::max(1, std::min(c.n_cu * 4, (nrows + 255) / 256));
    const int grid_n = std::max(1, std::min(c.n_cu * 4, (2 * n + 255) / 256));
    int rc;
    ProfScope ps(phase == 3 ? TD_K_CERT : TD_K_LINE);
    switch (phase) {
    case 0:   // anchors, from the rank that owns row 0
        TD_HIP(hipMemsetAsync(w, 0, L.end * 8, c.stream));
        if (row0 == 0 && nrows > 0) k_lsh_anchor<<<1, 1024, 0, c.stream>>>(n, nrows, cost_rows, ctl, (int32_t *)(w + L.anch), L.N2);
        *seg_off = 0, *seg_len = (int64_t)(L.anch + L.N2);
        break;
    case 1:   // row keys of the local rows
        if (nrows > 0) k_lsh_rowkeys<<<grid_rows, 256, 0, c.stream>>>(n, row0, nrows, cost_rows, ctl, w + L.rk);
        *seg_off = (int64_t)L.rk, *seg_len = n;
        break;
    case 2: {   // replicated: the sort; local: matched cells and their neighbours
        auto *kin = (unsigned long long *)(w + L.kin), *kout = (unsigned long long *)(w + L.kout);
        int *vin = (int *)(w + L.vin), *sorted = (int *)(w + L.sorted);
        k_lsh_keys<<<grid_n, 256, 0, c.stream>>>(n, ctl, (const int32_t *)(w + L.anch), L.N2, w + L.rk, kin, vin);
        size_t bytes = 0;
        TD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, sorted, 2 * n, 0, 64, c.stream));
        if ((rc = ensure(g_lw.tmp, bytes + 256))) return rc;
        TD_HIP(hipcub::DeviceRadixSort::SortPairs(g_lw.tmp.p, bytes, kin, kout, vin, sorted, 2 * n, 0, 64, c.stream));
        k_lsh_inv<<<grid_n, 256, 0, c.stream>>>(n, sorted, (int *)(w + L.inv), (int *)(w + L.r2c));
        if (nrows > 0)
            k_lsh_diffs<<<grid_rows, 256, 0, c.stream>>>(n, row0, nrows, cost_rows, sorted, (const int *)(w + L.inv), w + L.fb, w + L.fb + L.N2,
                                                         w + L.tot);
        *seg_off = (int64_t)L.fb, *seg_len = (int64_t)(2 * L.N2 + 1);
        break;
    }
    case 3: {   // replicated: Monge test and prices; local: the certificate pass over this rank's rows
        k_lsh_monge<<<grid_n, 256, 0, c.stream>>>(n, w + L.fb, w + L.fb + L.N2, ctl);
        k_line_scan<<<1, 1024, 0, c.stream>>>(n, (const int *)(w + L.sorted), ctl, w + L.fb, w + L.v64, (int32_t *)(w + L.v32));
        if (nrows > 0) {
            constexpr int R = 4;
            const int grid = std::max(1, std::min((nrows + R - 1) / R, c.n_cu * 16));
            const bool vec = (n % 4 == 0) && (((uintptr_t)cost_rows & 15) == 0);
            const int *r2c = (const int *)(w + L.r2c) + row0;
            if (vec)
                k_line_cert<R, true><<<grid, 256, 0, c.stream>>>(n, nrows, cost_rows, w + L.v64, (const int32_t *)(w + L.v32), r2c, ctl);
            else
                k_line_cert<R, false><<<grid, 256, 0, c.stream>>>(n, nrows, cost_rows, w + L.v64, (const int32_t *)(w + L.v32), r2c, ctl);
        }
        k_lsh_publish<<<1, 1, 0, c.stream>>>(ctl, w + L.failw);
        *seg_off = (int64_t)L.failw, *seg_len = 1;
        break;
    }
    default:
        return fail(TD_EINVAL, "td_line_shard_phase: phase %d (0..3)", phase);
    }
    TD_HIP(hipGetLastError());
    return TD_OK;
}

extern "C" int td_line_shard_result(int n, int row0, int nrows, const int64_t *ws, int32_t *row_to_col, int64_t *total, int32_t *accepted)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n < 2 || !ws || !total || !accepted || row0 < 0 || nrows < 0 || (long long)row0 + nrows > n)
        return fail(TD_EINVAL, "td_line_shard_result: bad argument");
    const LshLayout L(n);
    long long h[2];
    TD_HIP(hipMemcpyAsync(c.pinned, ws + L.tot, 16, hipMemcpyDeviceToHost, c.stream));   // total, summed verdicts
    TD_HIP(hipStreamSynchronize(c.stream));
    memcpy(h, c.pinned, 16);
    *accepted = h[1] == 0;
    *total = h[0];
    if (*accepted && row_to_col && nrows > 0) {
        TD_HIP(hipMemcpyAsync(row_to_col, (const int *)(ws + L.r2c) + row0, 4 * (size_t)nrows,
                              is_device_ptr(row_to_col) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
    }
    return TD_OK;
}
