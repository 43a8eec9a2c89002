// td_forest.h — sparse core + incremental shortest-path forest: the exact finisher for wide,
// tie-free rows (the |a-b| geometry of greedy_opt.py:86-99,122-133 and wide uniform costs).
// Included by td_assign.hip inside its anonymous namespace (uses Tr<>, unpack<>, wave_umin32).
//
// Why (measured with tools/forest_proto.c, |a-b| geometry): after the eps = 0 bidding rounds a
// quarter of the rows are free; single-source shortest augmenting paths then scan ~14 000 rows
// per augmentation at N = 16 384 (k_sapx: 811 ms for the last 242 rows).  The structure that is
// cheap instead:
//   * ONE multi-source Dijkstra over all free rows (a shortest-path FOREST).  When a tree reaches
//     a free column it is augmented and RELEASED at once (its lazy dual raises are materialised);
//     all other trees stay, the search continues from the same distance.  Columns whose best row
//     was in the released tree get their label recomputed over the remaining forest rows (a
//     column scan).  N = 4096: 5 100 distance levels, 19 N row scans, 37 N column scans in total
//     against 244 783 single steps / 11 800 levels + 250 N scans for restarted searches.
//   * the levels are a dependent chain (30-100 hops per path in this geometry), so a level must
//     cost ~1 us, not a grid barrier: the whole search runs in ONE workgroup with every label in
//     LDS — which needs rows of ~10^2 entries, not 16 384:
//   * the SPARSE CORE: per row the K entries of smallest reduced cost at the prices the bidding
//     rounds left (k_core_extract, a streaming pass at HBM rate), stored row-wise (CSR) and
//     column-wise (CSC, for the label repairs).  The forest runs on the core only.
//   * exactness comes from PRICING OUT the result against the dense matrix (k_core_extract in
//     check mode, one more streaming pass): every row's dense minimum of c + p must equal its
//     core dual u_i.  Rows that violate get their core re-extracted at the current prices and are
//     freed, the forest continues; no violation = the core optimum is the dense optimum (dual
//     feasible on every cell, matched cells tight).  |a-b| N = 16 384, K = 128..256: no violation.
//   * a search that runs out of reachable columns inside the core (status "stuck") re-extracts the
//     rows of the forest with twice as many entries.
// Everything is exact integer arithmetic; labels are 32-bit (distance from the start of the
// launch) and the kernel reports an error instead of wrapping.

struct FRow {
    long long u;     // row dual in compressed-cost units: min over the row's core of c' + p (not yet raised while in the forest)
    uint32_t a;      // distance at which the row joined the forest
    uint32_t fr;     // bit 31: in the forest; low 20 bits: root (row id) of its tree
};
constexpr uint32_t FR_IN = 0x80000000u;
constexpr uint32_t FR_ROOT = 0x000FFFFFu;
constexpr int FO_KFREE = 0;   // free columns appended to a row of a stuck search (0: none — direct cells to far free columns made thousands of rows fail the pricing pass)

// forest status words (int32) in their own small buffer
enum {
    FS_STATUS = 0,   // 0 = all rows matched, 1 = stuck (no reachable column inside the core), 2 = error
    FS_NFREE = 1,    // free rows at exit
    FS_LEVELS = 2,
    FS_JOINS = 3,
    FS_REPAIRS = 4,
    FS_EVENTS = 5,   // levels that released at least one tree
    FS_VIOL = 6,     // k_core_extract check mode: rows whose dense minimum is below their core dual
    FS_REFRESH = 7,  // rows re-extracted
    FS_ERRCODE = 8,
    FS_AUGS = 9,
    FS_D = 16,       // distance at which a stuck search stopped
    FS_NEGRED = 17,  // cells relaxed with a negative reduced cost (appended cells; priced out later)
    FS_WORDS = 20
};

// ---------------------------------------------------------------------------------------------
// k_core_extract: one 256-thread workgroup per row (grid-strided).  The row stays in registers.
//   mode 0  every row: u_i = min_j (c'_ij + p_j), K smallest reduced costs -> CSR
//   mode 1  check: rows with dense minimum < u_i (pricing violation) are re-extracted with their
//           current count and, when their matched cell is no longer tight, freed
//   mode 2  rows flagged FR_IN (the forest of a stuck search) are re-extracted with twice the count
// The matched cell of a row (always tight) is entry 0 of its list.
// ---------------------------------------------------------------------------------------------
template <typename CT, int CHK>
__global__ __launch_bounds__(256) void k_core_extract(int n, int nchunks, const CT *__restrict__ cc,
                                                       typename Tr<CT>::PT *__restrict__ pk, int *__restrict__ owner,
                                                       int *__restrict__ r2c, FRow *__restrict__ rowrec,
                                                       int *__restrict__ ccol, uint32_t *__restrict__ cval,
                                                       int *__restrict__ ccnt, int kcap, int K, int mode,
                                                       int *__restrict__ fs, const unsigned char *__restrict__ infc,
                                                       int *__restrict__ cold)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = Tr<CT>::E;
    constexpr int V = CHK * E;   // values per thread
    __shared__ long long s_m[4];
    __shared__ uint32_t s_c[4];
    __shared__ int s_scan[2][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t pitch = (size_t)nchunks * E;
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        FRow rr = rowrec[row];
        if (mode == 2 && !(rr.fr & FR_IN)) continue;
        const CT *rp = cc + (size_t)row * pitch;
        uint4 cv[CHK];
        long long m = LLONG_MAX;
#pragma unroll
        for (int k = 0; k < CHK; k++) {
            const int ch = tid + 256 * k;
            if (ch < nchunks) {
                cv[k] = *reinterpret_cast<const uint4 *>(rp + (size_t)ch * E);
                uint32_t c[E];
                unpack<CT>(cv[k], c);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const long long v = (long long)c[e] + (long long)(pk[(size_t)ch * E + e] >> 1);
                    const bool skip = (mode == 2) && (ch * E + e >= n || infc[ch * E + e]);
                    m = (!skip && v < m) ? v : m;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long om = __shfl_xor(m, o);
            m = om < m ? om : m;
        }
        __syncthreads();   // scratch reuse across rows
        if (lane == 0) s_m[w] = m;
        __syncthreads();
        m = min(min(s_m[0], s_m[1]), min(s_m[2], s_m[3]));
        int Krow = K, off0 = 0;
        if (mode == 1) {
            if (m >= rr.u) continue;   // dense minimum == core dual: the row prices out
            Krow = ccnt[row];
            if (tid == 0) {
                atomicAdd(&fs[FS_VIOL], 1);
                atomicAdd(&fs[FS_REFRESH], 1);
            }
        } else if (mode == 2) {
            // append: the K cheapest columns OUTSIDE the forest go behind the entries the row has.  The
            // duals and the matching are not touched (the interrupted search resumes): an appended cell
            // with a negative reduced cost is relaxed as 0 by the forest and caught by the pricing pass.
            off0 = ccnt[row];
            Krow = min(K, kcap - off0);
            if (tid == 0) {
                cold[row] = off0;
                atomicAdd(&fs[FS_REFRESH], 1);
                if (Krow <= 0) atomicMax(&fs[FS_ERRCODE], 9);   // list full: the caller gives up
            }
            if (Krow <= 0 || m == LLONG_MAX) continue;
        }
        Krow = min(Krow, min(kcap, n));
        // matched cell: keep it when it is tight at the new dual, else free the row
        int jm = r2c[row];
        __syncthreads();   // every thread has read r2c[row] before thread 0 may change it
        bool forced = false;
        const long long unew = (mode == 2) ? rr.u : m;
        if (jm >= 0 && mode != 2) {
            const long long vm = (long long)rp[jm] + (long long)(pk[jm] >> 1);
            if (vm == unew) forced = true;
            else if (tid == 0) {
                r2c[row] = -1;
                owner[jm] = -1;
                pk[jm] = (PT)((pk[jm] >> 1) << 1);
            }
        }
        // Selection passes.  Modes 0 / 1: one pass, the Krow cheapest cells.  Mode 2: the cheapest columns
        // outside the forest, then the cheapest FREE columns (the search then always has a direct cell to
        // a free column; whether that cell is on a shortest path of the dense problem is the pricing
        // pass's business).
        const size_t base = (size_t)row * kcap;
        int written = off0 + (forced ? 1 : 0);
        const int kfree = (mode == 2) ? min(FO_KFREE, max(0, Krow - 1)) : 0;
        for (int ps = 0; ps < (mode == 2 ? 2 : 1); ps++) {
        // reduced costs (saturated to 32 bits) in registers
        uint32_t r[V];
        uint32_t rmax = 0;
#pragma unroll
        for (int k = 0; k < CHK; k++) {
            const int ch = tid + 256 * k;
            uint32_t c[E];
            if (ch < nchunks) unpack<CT>(cv[k], c);
#pragma unroll
            for (int e = 0; e < E; e++) {
                uint32_t x = 0xFFFFFFFFu;
                const int j = ch * E + e;
                if (ch < nchunks && j < n) {
                    const long long d = (long long)c[e] + (long long)(pk[j] >> 1) - m;
                    x = d < 0xFFFFFFF0ll ? (uint32_t)d : 0xFFFFFFF0u;
                    if ((forced && j == jm) || (mode == 2 && ps == 0 && infc[j]) || (ps == 1 && owner[j] >= 0))
                        x = 0xFFFFFFFFu;   // entry 0 below / inside the forest / not a free column
                    else rmax = x > rmax ? x : rmax;
                }
                r[k * E + e] = x;
            }
        }
        const int want = (mode == 2) ? (ps == 0 ? Krow - kfree : kfree) : Krow - (forced ? 1 : 0);
        // threshold T = want-th smallest reduced cost: binary search on the value
        {
            uint32_t t = rmax;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t ot = __shfl_xor(t, o);
                t = ot > t ? ot : t;
            }
            __syncthreads();
            if (lane == 0) s_c[w] = t;
            __syncthreads();
            rmax = max(max(s_c[0], s_c[1]), max(s_c[2], s_c[3]));
        }
        uint32_t lo = 0, hi = rmax;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < V; q++) cnt += (r[q] <= mid) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            __syncthreads();
            if (lane == 0) s_c[w] = (uint32_t)cnt;
            __syncthreads();
            const int tot = (int)(s_c[0] + s_c[1] + s_c[2] + s_c[3]);
            if (tot >= want) hi = mid;
            else lo = mid + 1;
        }
        const uint32_t T = lo;
        // compaction: all entries < T, then entries == T in thread order until `want` are taken
        int clt = 0, ceq = 0;
#pragma unroll
        for (int q = 0; q < V; q++) {
            clt += (r[q] < T) ? 1 : 0;
            ceq += (r[q] == T) ? 1 : 0;
        }
        int ilt = clt, ieq = ceq;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(ilt, o), b = __shfl_up(ieq, o);
            if (lane >= o) {
                ilt += a;
                ieq += b;
            }
        }
        __syncthreads();
        if (lane == 63) {
            s_scan[0][w] = ilt;
            s_scan[1][w] = ieq;
        }
        __syncthreads();
        int blt = 0, beq = 0, tlt = 0, teq = 0;
        for (int k = 0; k < 4; k++) {
            if (k < w) {
                blt += s_scan[0][k];
                beq += s_scan[1][k];
            }
            tlt += s_scan[0][k];
            teq += s_scan[1][k];
        }
        int plt = blt + ilt - clt, peq = beq + ieq - ceq;
        // The cells tied at the threshold are taken starting from a row-dependent thread (rotated
        // order).  With thousands of tied cells per row (near-optimal prices in the |a-b| geometry:
        // a cab is tight with every request on one side of it) a fixed order makes every row keep
        // the SAME few columns and the core has no perfect matching; rotated, the rows of a tie class
        // spread over it like a random graph.
        {
            const int rot = (int)((((uint32_t)row + 1u) * 0x9E3779B1u) >> 24);   // 0..255
            __syncthreads();
            if (tid == rot) s_scan[0][0] = peq;
            __syncthreads();
            const int prot = s_scan[0][0];
            peq = (tid >= rot) ? peq - prot : peq + teq - prot;
        }
        const int need_eq = (T == 0xFFFFFFFFu) ? 0 : min(teq, max(0, want - tlt));
        const int off = written;
#pragma unroll
        for (int k = 0; k < CHK; k++) {
            const int ch = tid + 256 * k;
            uint32_t c[E];
            if (ch < nchunks) unpack<CT>(cv[k], c);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t x = r[k * E + e];
                const int j = ch * E + e;
                if (x < T) {
                    if (plt < want) {
                        ccol[base + off + plt] = j;
                        cval[base + off + plt] = c[e];
                    }
                    plt++;
                } else if (x == T && T != 0xFFFFFFFFu) {
                    if (peq < need_eq) {
                        ccol[base + off + tlt + peq] = j;
                        cval[base + off + tlt + peq] = c[e];
                    }
                    peq++;
                }
            }
        }
        written = off + min(want, tlt) + need_eq;
        __syncthreads();
        }
        if (tid == 0) {
            if (forced) {
                ccol[base] = jm;
                cval[base] = (uint32_t)rp[jm];
            }
            ccnt[row] = written;
            if (mode != 2) {
                rr.u = unew;
                rr.a = 0;
                rr.fr = 0;
                rowrec[row] = rr;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CSC of the core (column lists for the label repairs): count, scan, fill.  The order of a
// column's entries depends on atomics; the repair takes a lexicographic (label, row) minimum, so
// the result does not.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_csc_count(int n, int kcap, const int *__restrict__ ccol, const int *__restrict__ ccnt,
                                                   int *__restrict__ tcnt)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + w; row < n; row += gridDim.x * 4) {
        const int cnt = ccnt[row];
        for (int e = lane; e < cnt; e += 64) atomicAdd(&tcnt[ccol[(size_t)row * kcap + e]], 1);
    }
}

// exclusive scan of tcnt[0..n) into tptr[0..n], cursor copy in tcur; one workgroup
__global__ __launch_bounds__(1024) void k_csc_scan(int n, const int *__restrict__ tcnt, int *__restrict__ tptr, int *__restrict__ tcur)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int j = base + tid;
        const int v = j < n ? tcnt[j] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(inc, o);
            if (lane >= o) inc += a;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int b = s_carry;
        for (int k = 0; k < w; k++) b += s_w[k];
        if (j < n) {
            tptr[j] = b + inc - v;
            tcur[j] = b + inc - v;
        }
        __syncthreads();
        if (tid == 1023) s_carry = b + inc;
        __syncthreads();
    }
    if (tid == 0) tptr[n] = s_carry;
}

__global__ __launch_bounds__(256) void k_csc_fill(int n, int kcap, const int *__restrict__ ccol, const uint32_t *__restrict__ cval,
                                                  const int *__restrict__ ccnt, int *__restrict__ tcur, int *__restrict__ trow,
                                                  uint32_t *__restrict__ tval)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + w; row < n; row += gridDim.x * 4) {
        const int cnt = ccnt[row];
        for (int e = lane; e < cnt; e += 64) {
            const int j = ccol[(size_t)row * kcap + e];
            const int pos = atomicAdd(&tcur[j], 1);
            trow[pos] = row;
            tval[pos] = cval[(size_t)row * kcap + e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_core_cols: column coverage.  The K row-wise entries do not guarantee that every COLUMN occurs
// in some row's list (a request far from every cab would be unreachable inside the core), so each
// column also contributes its KC rows of smallest reduced cost c'_ij + p_j - u_i.  One workgroup
// per tile of 64 columns, wave g scans rows g, g+4, ...; fixed KC entries per column (stride KC).
// ---------------------------------------------------------------------------------------------
template <typename CT, int KC>
__global__ __launch_bounds__(256) void k_core_cols(int n, int nchunks, const CT *__restrict__ cc,
                                                   const typename Tr<CT>::PT *__restrict__ pk, const FRow *__restrict__ rowrec,
                                                   int *__restrict__ c2row, uint32_t *__restrict__ c2val, int *__restrict__ c2cnt)
{
    constexpr int E = Tr<CT>::E;
    constexpr int U = 8;
    __shared__ uint32_t s_r[4][KC][64];
    __shared__ int s_i[4][KC][64];
    const size_t pitch = (size_t)nchunks * E;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const bool live = j < n;
    const long long pj = live ? (long long)(pk[j] >> 1) : 0;
    uint32_t br[KC];
    int bi[KC];
#pragma unroll
    for (int t = 0; t < KC; t++) {
        br[t] = 0xFFFFFFFFu;
        bi[t] = -1;
    }
    for (int i0 = g; i0 < n; i0 += 4 * U) {
        uint32_t v[U];
        long long uu[U];
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int i = i0 + 4 * q;
            v[q] = 0;
            uu[q] = 0;
            if (i < n) {
                uu[q] = rowrec[i].u;
                if (live) v[q] = (uint32_t)cc[(size_t)i * pitch + j];
            }
        }
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int i = i0 + 4 * q;
            if (i < n && live) {
                long long d = (long long)v[q] + pj - uu[q];
                d = d < 0 ? 0 : d;
                uint32_t x = d < 0xFFFFFFF0ll ? (uint32_t)d : 0xFFFFFFF0u;
                if (x < br[KC - 1]) {   // rows come in ascending order, so the first of equal values stays
                    int xi = i;
#pragma unroll
                    for (int t = 0; t < KC; t++) {
                        if (x < br[t]) {
                            const uint32_t tr = br[t];
                            const int ti = bi[t];
                            br[t] = x;
                            bi[t] = xi;
                            x = tr;
                            xi = ti;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KC; t++) {
        s_r[g][t][c] = br[t];
        s_i[g][t][c] = bi[t];
    }
    __syncthreads();
    if (g == 0 && live) {
        int ptr[4] = {0, 0, 0, 0};
        int cnt = 0;
        for (int t = 0; t < KC; t++) {   // 4-way merge of the sorted partial lists; ties -> smaller row
            uint32_t best = 0xFFFFFFFFu;
            int bw = -1, brow = INT_MAX;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (ptr[q] < KC) {
                    const uint32_t x = s_r[q][ptr[q]][c];
                    const int xi = s_i[q][ptr[q]][c];
                    if (xi >= 0 && (x < best || (x == best && xi < brow))) {
                        best = x;
                        bw = q;
                        brow = xi;
                    }
                }
            }
            if (bw < 0) break;
            ptr[bw]++;
            c2row[(size_t)j * KC + cnt] = brow;
            c2val[(size_t)j * KC + cnt] = (uint32_t)cc[(size_t)brow * pitch + j];
            cnt++;
        }
        c2cnt[j] = cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// k_forest: the incremental multi-source Dijkstra on the core, ONE workgroup of 1024 threads.
// Thread t owns columns t, t + 1024, ...  LDS: label+1 per column (0 = column in the forest,
// 0xFFFFFFFF = unreached), pred row (u16), row -> column (u16), small lists.
// ---------------------------------------------------------------------------------------------
constexpr int FO_T = 1024;
constexpr int FO_LIST = 1024;    // joiners / new rows / ends handled per step
constexpr int FO_REP = 2048;     // repairs gathered per sweep
constexpr int FO_EPL = 4;        // core entries per lane and relax chunk (256 entries per row and chunk)
constexpr int FO_KC = 8;         // rows per column of the column coverage lists
constexpr uint32_t FO_INF = 0xFFFFFFFFu;
constexpr uint32_t FO_SAT = 0xFFFFFF00u;   // labels at or above this are treated as unreachable / error

struct ForestArgs {
    int n, kcap;
    long long *pr;            // plain prices (scratch, n)
    void *pk;                 // packed prices (PT)
    int *owner, *r2c;
    FRow *rowrec;
    // row lists (k_core_extract) and their column view
    const int *ccol;
    const uint32_t *cval;
    const int *ccnt;
    const int *tptr, *trow;
    const uint32_t *tval;
    // column coverage lists (k_core_cols, stride FO_KC) and their row view
    const int *c2row;
    const uint32_t *c2val;
    const int *c2cnt;
    const int *xptr, *xcol;
    const uint32_t *xval;
    int *rootr;               // root of a forest row (compact copy of FRow::fr's root)
    int *rootc;               // root of a forest column
    uint32_t *acol;           // distance at which a forest column joined
    int *claim;               // per root: smallest end column of the current level (INT_MAX = none)
    int *flist;               // free-row list scratch (n)
    unsigned char *infc;      // exit (stuck): 1 = column inside the forest
    uint32_t *sv_slack;       // exit (stuck): labels and preds, reloaded by the resuming launch
    uint16_t *sv_pred;
    const int *cold;          // per row: entries it had before the last append (k_core_extract mode 2)
    int resume;               // 1: continue the interrupted search
    int *fs;                  // status / counters
};

__device__ __forceinline__ void lds_min_u16(uint16_t *base, int idx, uint32_t v)
{
    uint32_t *wp = reinterpret_cast<uint32_t *>(base) + (idx >> 1);
    const int sh = (idx & 1) * 16;
    uint32_t old = *wp;
    for (;;) {
        const uint32_t cur = (old >> sh) & 0xFFFFu;
        if (cur <= v) break;
        const uint32_t nw = (old & ~(0xFFFFu << sh)) | (v << sh);
        const uint32_t got = atomicCAS(wp, old, nw);
        if (got == old) break;
        old = got;
    }
}

// optional section timing (build with -DTD_FOREST_PROF=1): cycles of thread 0 per section, in fs[10..15]
#if defined(TD_FOREST_PROF) && TD_FOREST_PROF
#define FO_TICK(k)                                   \
    do {                                             \
        if (tid == 0) {                              \
            const long long _t = (long long)clock64(); \
            prof[k] += _t - tlast;                   \
            tlast = _t;                              \
        }                                            \
    } while (0)
#else
#define FO_TICK(k) do { } while (0)
#endif

struct FoNew {       // a row that joined in this step, with what its relaxation needs (saves a round trip)
    long long u;
    uint32_t a;
    int row;
    int cnt1, x0, x1;
    int e0;          // first entry to relax (resume: only the appended ones)
};

template <typename PT, int NPT>
__global__ __launch_bounds__(FO_T) void k_forest(ForestArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = A.n, kcap = A.kcap;
    constexpr int npad = NPT * FO_T;   // LDS arrays cover whole slices of the column ownership
    uint32_t *slack = reinterpret_cast<uint32_t *>(smem);                       // npad
    uint16_t *pred = reinterpret_cast<uint16_t *>(slack + npad);                // npad
    uint16_t *r2cs = pred + npad;                                               // npad
    uint32_t *needrep = reinterpret_cast<uint32_t *>(r2cs + npad);              // npad / 32
    uint16_t *jl = reinterpret_cast<uint16_t *>(needrep + npad / 32);           // FO_LIST
    uint16_t *ends = jl + FO_LIST;                                              // FO_LIST
    uint32_t *endroot = reinterpret_cast<uint32_t *>(ends + FO_LIST);           // FO_LIST
    uint16_t *replist = reinterpret_cast<uint16_t *>(endroot + FO_LIST);        // FO_REP
    __shared__ FoNew s_new[FO_LIST >> 3];   // 128 rows per relax sweep
    __shared__ uint32_t s_red[16];
    __shared__ int s_njl, s_nnew, s_nends, s_nrep, s_nfree, s_keep, s_err, s_maxe, s_negred;
    __shared__ int s_fl_w[16];
    constexpr int NEWCAP = FO_LIST >> 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    PT *pk = reinterpret_cast<PT *>(A.pk);
    int levels = 0, joins = 0, repairs = 0, events = 0;
#if defined(TD_FOREST_PROF) && TD_FOREST_PROF
    long long prof[6] = {0, 0, 0, 0, 0, 0}, tlast = (long long)clock64();
#endif

    // ---- init: labels, mirrors, plain prices, free-row list = the roots (or the saved state)
    const bool resume = A.resume != 0;
    for (int j = tid; j < npad; j += FO_T) {
        uint32_t sl = FO_INF;
        uint16_t pd = 0xFFFFu;
        int c = -1;
        if (j < n) {
            c = A.r2c[j];
            A.pr[j] = (long long)(pk[j] >> 1);
            A.claim[j] = INT_MAX;
            if (resume) {
                sl = A.sv_slack[j];
                pd = A.sv_pred[j];
            }
        }
        slack[j] = sl;
        pred[j] = pd;
        r2cs[j] = c >= 0 ? (uint16_t)c : (uint16_t)0xFFFFu;
    }
    for (int j = tid; j < npad / 32; j += FO_T) needrep[j] = 0;
    if (tid == 0) {
        s_err = 0;
        s_nnew = 0;
        s_negred = 0;
    }
    if (!resume)
        for (int i = tid; i < n; i += FO_T) {   // no row is in a forest yet (flags of an abandoned search)
            FRow rr = A.rowrec[i];
            if (rr.fr | rr.a) {
                rr.fr = 0;
                rr.a = 0;
                A.rowrec[i] = rr;
            }
        }
    __syncthreads();
    int nroots;
    {   // ordered list of the free rows (resume: of all forest rows) by a block scan over contiguous slices
        const int per = (n + FO_T - 1) / FO_T;
        const int lo = tid * per, hi = min(n, lo + per);
        int cnt = 0, cfree = 0;
        for (int r = lo; r < hi; r++) {
            const bool fr = r2cs[r] == 0xFFFFu;
            cfree += fr ? 1 : 0;
            cnt += (resume ? (A.rowrec[r].fr & FR_IN) != 0 : fr) ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cfree += __shfl_xor(cfree, o);
        if (lane == 63) s_fl_w[w] = incl;
        if (lane == 0) s_red[w] = (uint32_t)cfree;
        __syncthreads();
        int base = 0, tot = 0, totfree = 0;
        for (int k = 0; k < 16; k++) {
            if (k < w) base += s_fl_w[k];
            tot += s_fl_w[k];
            totfree += (int)s_red[k];
        }
        int pos = base + incl - cnt;
        for (int r = lo; r < hi; r++) {
            if (resume) {
                if (A.rowrec[r].fr & FR_IN) A.flist[pos++] = r;
            } else if (r2cs[r] == 0xFFFFu) {
                A.flist[pos++] = r;
                FRow rr = A.rowrec[r];
                rr.a = 0;
                rr.fr = FR_IN | (uint32_t)r;
                A.rowrec[r] = rr;
                A.rootr[r] = r;
            }
        }
        nroots = tot;
        if (tid == 0) s_nfree = totfree;
        __syncthreads();
    }
    uint32_t D = resume ? (uint32_t)A.fs[FS_D] : 0u;
    int status = 0;

    // Relax the rows in s_new[0..cnt): one row per wave and pass, 256 entries per chunk.
    // Phase A: atomic min of the labels (a column inside the forest holds 0, which no min changes).
    // Phase B: the rows that reach the final label agree on the smallest row id (deterministic).
    // `last_sync` = false leaves the closing barrier to the caller's next barrier.
    auto relax_new = [&](int cnt) {
        for (int b = 0; b < cnt; b += 16) {
            const int k = b + w;
            const bool have = k < cnt;
            FoNew nr;
            int tot = 0;
            if (have) {
                nr = s_new[k];
                tot = (nr.cnt1 - nr.e0) + (nr.x1 - nr.x0);
            }
            // chunks of 256 entries; the count is made uniform over the workgroup only when some row is long
            int nch = (tot + 255) >> 8;
            if (nch > 1) atomicMax(&s_maxe, nch);
            __syncthreads();
            const int nchunks_all = max(1, s_maxe);
            for (int chn = 0; chn < nchunks_all; chn++) {
                int jj[FO_EPL];
                uint32_t hh[FO_EPL];
                if (have) {
                    // all loads of the chunk first (entries, then the prices they point at), then the atomics
                    const size_t base = (size_t)nr.row * kcap;
                    uint32_t cv[FO_EPL];
                    long long pj[FO_EPL];
#pragma unroll
                    for (int q = 0; q < FO_EPL; q++) {
                        const int e = nr.e0 + chn * 256 + lane + 64 * q;
                        jj[q] = -1;
                        cv[q] = 0;
                        if (e - nr.e0 < tot) {
                            if (e < nr.cnt1) {
                                jj[q] = A.ccol[base + e];
                                cv[q] = A.cval[base + e];
                            } else {
                                jj[q] = A.xcol[nr.x0 + (e - nr.cnt1)];
                                cv[q] = A.xval[nr.x0 + (e - nr.cnt1)];
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < FO_EPL; q++) pj[q] = jj[q] >= 0 ? A.pr[jj[q]] : 0;
#pragma unroll
                    for (int q = 0; q < FO_EPL; q++) {
                        const int j = jj[q];
                        jj[q] = -1;
                        if (j >= 0) {
                            // a negative reduced cost (a cell appended at a stuck search, see k_core_extract
                            // mode 2; inside the forest both duals are lazy and the value means nothing) counts
                            // as 0: the forest solves the core with that cell made dearer, the pricing pass
                            // finds the row afterwards
                            const long long red0 = (long long)cv[q] + pj[q] - nr.u;
                            const long long red = red0 < 0 ? 0 : red0;
                            const long long h64 = (long long)nr.a + red;
                            if (h64 < (long long)FO_SAT) {
                                const uint32_t h1 = (uint32_t)h64 + 1u;
                                const uint32_t old = atomicMin(&slack[j], h1);
                                if (old != 0u && red0 < 0) s_negred = 1;
                                if (old != 0u && h1 <= old) {
                                    if (h1 < old) pred[j] = 0xFFFFu;
                                    jj[q] = j;
                                    hh[q] = h1;
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < FO_EPL; q++) jj[q] = -1;
                }
                __syncthreads();
                if (tid == 0) s_maxe = 0;   // everybody has read it; the next sweep's atomicMax comes after the closing barrier
                if (have) {
#pragma unroll
                    for (int q = 0; q < FO_EPL; q++)
                        if (jj[q] >= 0 && slack[jj[q]] == hh[q]) lds_min_u16(pred, jj[q], (uint32_t)nr.row);
                }
                if (chn + 1 < nchunks_all) __syncthreads();   // the next chunk's phase A may lower a label phase B still compares
            }
            __syncthreads();
        }
    };
    auto stage_row = [&](int slot, int row, long long u, uint32_t a) {
        FoNew nr;
        nr.u = u;
        nr.a = a;
        nr.row = row;
        nr.cnt1 = A.ccnt[row];
        nr.x0 = A.xptr[row];
        nr.x1 = A.xptr[row + 1];
        nr.e0 = 0;
        s_new[slot] = nr;
    };
    if (tid == 0) s_maxe = 0;
    __syncthreads();
    // roots first
    for (int b = 0; b < nroots; b += NEWCAP) {
        const int cnt = min(NEWCAP, nroots - b);
        if (tid < cnt) {
            const int r = A.flist[b + tid];
            const FRow rr = A.rowrec[r];
            stage_row(tid, r, rr.u, resume ? rr.a : 0u);
            if (resume) {   // the interrupted search has relaxed everything but the appended entries
                s_new[tid].e0 = A.cold[r];
                s_new[tid].x1 = s_new[tid].x0;
            }
        }
        __syncthreads();
        relax_new(cnt);
    }
    if (!resume) joins += nroots;

    for (;;) {
        if (s_nfree == 0) break;
        if (s_err) {
            status = 2;
            break;
        }
        // ---- next distance: block minimum of the labels of the non-forest columns
        uint32_t mk = FO_INF;
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            const uint32_t x = slack[tid + FO_T * k] - 1u;   // forest (0) -> 0xFFFFFFFF, unreached -> 0xFFFFFFFE
            mk = x < mk ? x : mk;
        }
        mk = wave_umin32(mk);
        if (lane == 0) s_red[w] = mk;
        if (tid == 0) {
            s_njl = 0;
            s_nends = 0;
            s_nnew = 0;
        }
        __syncthreads();
        uint32_t m = s_red[lane & 15];
        m = wave_umin32(m);
        if (m >= FO_SAT) {
            status = (m >= 0xFFFFFFFEu) ? 1 : 2;   // nothing reachable inside the core / label overflow
            if (status == 2 && tid == 0) A.fs[FS_ERRCODE] = 1;
            break;
        }
        D = m;
        levels++;
        // ---- columns at distance D join.  Up to NEWCAP of them: one pass over an unordered list (the
        // SET is deterministic, and nothing below depends on the order inside a pass).  More: passes
        // over fixed (slice, thread range) groups of the column ownership, so that the grouping does
        // not depend on the order of atomics either.
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            if (slack[tid + FO_T * k] == m + 1u) {
                const int pos = atomicAdd(&s_njl, 1);
                if (pos < NEWCAP) jl[pos] = (uint16_t)(tid + FO_T * k);
            }
        }
        __syncthreads();
        const int total_j = s_njl;
        const bool listmode = total_j <= NEWCAP;
        const int npass = listmode ? 1 : NPT * (FO_T / NEWCAP);
        FO_TICK(0);
        for (int ps = 0; ps < npass; ps++) {
            {
                int j = -1;
                if (listmode) {
                    if (tid < total_j) j = jl[tid];
                } else {
                    const int k = ps / (FO_T / NEWCAP), c = ps % (FO_T / NEWCAP);
                    // (a column whose best row was released by an earlier pass has been repaired by now)
                    if ((tid / NEWCAP) == c && slack[tid + FO_T * k] == m + 1u) j = tid + FO_T * k;
                }
                if (tid == 0) {
                    s_nnew = 0;
                    s_nends = 0;
                }
                if (!__syncthreads_or(j >= 0)) continue;
                if (j >= 0) {
                    int ip = pred[j];
                    if (ip == 0xFFFF) {
                        s_err = 6;
                        ip = 0;
                    }
                    const int o = A.owner[j];
                    const uint32_t root = (uint32_t)A.rootr[ip];
                    if (o < 0) {
                        const int e = atomicAdd(&s_nends, 1);
                        ends[e] = (uint16_t)j;
                        endroot[e] = root;
                        atomicMin(&A.claim[root], j);
                    } else {
                        slack[j] = 0u;   // in the forest
                        A.acol[j] = D;
                        A.rootc[j] = (int)root;
                        FRow rr = A.rowrec[o];
                        rr.a = D;
                        rr.fr = FR_IN | root;
                        A.rowrec[o] = rr;
                        A.rootr[o] = (int)root;
                        stage_row(atomicAdd(&s_nnew, 1), o, rr.u, D);
                    }
                }
                __syncthreads();
                const int nends = s_nends;
                int nnew = s_nnew;
                joins += nnew;
                FO_TICK(1);
                if (nends > 0) {
                    events++;
                    // ---- sweep 1: released forest columns (materialise the price raise) and the columns
                    // whose best row sits in a released tree (label repair)
                    uint32_t relmask = 0;
#pragma unroll 4
                    for (int k = 0; k < NPT; k++) {
                        const int j = tid + FO_T * k;
                        if (j < n) {
                            const uint32_t s = slack[j];
                            if (s == 0u) {
                                const int root = A.rootc[j];
                                if (__hip_atomic_load(&A.claim[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != INT_MAX) {
                                    relmask |= 1u << k;
                                    A.pr[j] += (long long)(D - A.acol[j]);
                                    atomicOr(&needrep[j >> 5], 1u << (j & 31));
                                }
                            } else if (s != FO_INF) {
                                const uint32_t root = (uint32_t)A.rootr[pred[j]];
                                if (__hip_atomic_load(&A.claim[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != INT_MAX)
                                    atomicOr(&needrep[j >> 5], 1u << (j & 31));
                            }
                        }
                    }
                    __syncthreads();
                    // ---- sweep 2: rows of the released columns (owner before the paths are flipped)
#pragma unroll 4
                    for (int k = 0; k < NPT; k++) {
                        if (relmask & (1u << k)) {
                            const int j = tid + FO_T * k;
                            const int o = A.owner[j];
                            FRow rr = A.rowrec[o];
                            rr.u += (long long)(D - rr.a);
                            rr.a = 0;
                            rr.fr = 0;
                            A.rowrec[o] = rr;
                        }
                    }
                    __syncthreads();
                    // ---- winners: one end per tree (smallest column), flip its path, release its root
                    if (tid < nends) {
                        const int j = ends[tid];
                        const uint32_t root = endroot[tid];
                        if (__hip_atomic_load(&A.claim[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == j) {
                            FRow rr = A.rowrec[root];
                            rr.u += (long long)(D - rr.a);
                            rr.a = 0;
                            rr.fr = 0;
                            A.rowrec[root] = rr;
                            int jj = j, guard = 0;
                            for (;;) {
                                const int i = pred[jj];
                                const int jn = r2cs[i];
                                A.owner[jj] = i;
                                A.r2c[i] = jj;
                                r2cs[i] = (uint16_t)jj;
                                if (jn == 0xFFFF) break;
                                jj = jn;
                                if (++guard > n) {
                                    s_err = 4;
                                    break;
                                }
                            }
                            atomicSub(&s_nfree, 1);
                            atomicAdd(&A.fs[FS_AUGS], 1);
                        }
                    }
                    __syncthreads();
                    if (tid < nends) A.claim[endroot[tid]] = INT_MAX;
#pragma unroll
                    for (int k = 0; k < NPT; k++) {
                        if (relmask & (1u << k)) {   // out of the forest; label repaired below
                            slack[tid + FO_T * k] = FO_INF;
                            pred[tid + FO_T * k] = 0xFFFFu;
                        }
                    }
                    // rows that joined in this step but belong to a released tree were released above
                    if (tid == 0) s_keep = 0;
                    __syncthreads();
                    FoNew keep;
                    bool kp = false;
                    if (tid < nnew) {
                        keep = s_new[tid];
                        kp = (A.rowrec[keep.row].fr & FR_IN) != 0;
                    }
                    __syncthreads();
                    if (kp) s_new[atomicAdd(&s_keep, 1)] = keep;
                    __syncthreads();
                    nnew = s_keep;
                    FO_TICK(2);
                    // ---- label repairs over the remaining forest rows (column scans)
                    for (;;) {
                        if (tid == 0) s_nrep = 0;
                        __syncthreads();
                        for (int wd = tid; wd < npad / 32; wd += FO_T) {
                            uint32_t bits = needrep[wd];
                            uint32_t left = bits;
                            while (bits) {
                                const int b = __ffs((int)bits) - 1;
                                bits &= bits - 1;
                                const int pos = atomicAdd(&s_nrep, 1);
                                if (pos < FO_REP) {
                                    replist[pos] = (uint16_t)(wd * 32 + b);
                                    left &= ~(1u << b);
                                }
                            }
                            needrep[wd] = left;
                        }
                        __syncthreads();
                        const int nrep_all = s_nrep;
                        const int nrep = min(nrep_all, FO_REP);
                        repairs += nrep;
                        for (int k = w; k < nrep; k += 16) {
                            const int j = replist[k];
                            const int e0 = A.tptr[j], e1 = A.tptr[j + 1];
                            const int n1 = e1 - e0, n2 = A.c2cnt[j];
                            const long long pj = A.pr[j];
                            uint32_t bh = FO_INF, bi = 0xFFFFu;
                            for (int eb = 0; eb < n1 + n2; eb += 256) {
                                int ri[4];
                                uint32_t cv[4];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const int e = eb + lane + 64 * q;
                                    ri[q] = -1;
                                    cv[q] = 0;
                                    if (e < n1) {
                                        ri[q] = A.trow[e0 + e];
                                        cv[q] = A.tval[e0 + e];
                                    } else if (e < n1 + n2) {
                                        ri[q] = A.c2row[(size_t)j * FO_KC + (e - n1)];
                                        cv[q] = A.c2val[(size_t)j * FO_KC + (e - n1)];
                                    }
                                }
                                FRow rq[4];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    rq[q].fr = 0;
                                    if (ri[q] >= 0) rq[q] = A.rowrec[ri[q]];
                                }
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    if (rq[q].fr & FR_IN) {
                                        const long long red0 = (long long)cv[q] + pj - rq[q].u;
                                        const long long red = red0 < 0 ? 0 : red0;
                                        const long long h64 = (long long)rq[q].a + red;
                                        if (red0 < 0) s_negred = 1;
                                        if (h64 < (long long)FO_SAT) {
                                            const uint32_t h1 = (uint32_t)h64 + 1u;
                                            if (h1 < bh || (h1 == bh && (uint32_t)ri[q] < bi)) {
                                                bh = h1;
                                                bi = (uint32_t)ri[q];
                                            }
                                        }
                                    }
                                }
                            }
                            const uint32_t mh = wave_umin32(bh);
                            const uint32_t mi = wave_umin32(bh == mh ? bi : 0xFFFFFFFFu);
                            if (lane == 0) {
                                slack[j] = mh;
                                pred[j] = (uint16_t)(mh == FO_INF ? 0xFFFFu : mi);
                            }
                        }
                        __syncthreads();
                        if (nrep_all <= FO_REP) break;
                    }
                }
                // ---- the rows that joined relax their core entries
                FO_TICK(3);
                relax_new(nnew);
                FO_TICK(4);
                if (s_nfree == 0) break;
            }
        }
    }
    __syncthreads();
    if (s_err && status == 0) status = 2;
    // ---- exit.  status 0: every tree was released, nothing is lazy.  Stuck / error: the unfinished
    // trees are dropped WITHOUT their raises (the duals stay feasible on the core and the matched
    // cells tight: raises are only ever applied to released trees); the rows keep their forest flag
    // so that k_core_extract (mode 2) widens exactly them.
    if (status == 0) {
        for (int i = tid; i < n; i += FO_T) {
            FRow rr = A.rowrec[i];
            if (rr.fr & FR_IN) {   // cannot happen: a tree without its root
                s_err = 7;
                rr.fr = 0;
                A.rowrec[i] = rr;
            }
        }
    }
    __syncthreads();
    if (status != 0)
        for (int j = tid; j < n; j += FO_T) {
            A.infc[j] = slack[j] == 0u ? 1 : 0;
            A.sv_slack[j] = slack[j];
            A.sv_pred[j] = pred[j];
        }
    for (int j = tid; j < n; j += FO_T) pk[j] = (PT)((PT)A.pr[j] << 1) | (PT)(A.owner[j] >= 0 ? 1 : 0);
    if (tid == 0) {
        if (s_err && status == 0) status = 2;
        A.fs[FS_STATUS] = status;
        A.fs[FS_NFREE] = s_nfree;
        A.fs[FS_D] = (int)D;
        if (s_negred) A.fs[FS_NEGRED] += 1;
        A.fs[FS_LEVELS] += levels;
        A.fs[FS_JOINS] += joins;
        A.fs[FS_REPAIRS] += repairs;
        A.fs[FS_EVENTS] += events;
        if (s_err) A.fs[FS_ERRCODE] = s_err;
#if defined(TD_FOREST_PROF) && TD_FOREST_PROF
        for (int k = 0; k < 6; k++) A.fs[10 + k] += (int)(prof[k] >> 10);
#endif
    }
}
