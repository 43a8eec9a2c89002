/*
 * td_oracle.c — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's assignment hot path, used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under taxidispatcher_amd/ may import, link or call this file.
 *
 * What it restates (citations are into the reference repo, read as text):
 *   - cost-matrix build      procedure.py:6-12, greedy_opt.py:86-99,
 *                            simulate.py:17-33, Simulator.java:493-520
 *   - LCM greedy             greedy_opt.py:61-82, simulate.py:76-98,
 *                            heuristic.py:24-33, Simulator.java:523-549
 *   - optimal assignment     the value cvxopt.glpk.ilp returns at
 *                            solver.py:26 / procedure.py:27 / greedy_opt.py:117
 *                            (GLPK itself is a third-party dependency that is
 *                            not in the reference tree and is not installed
 *                            here; no version is pinned by the reference).
 *                            The assignment polytope is totally unimodular, so
 *                            the ILP optimum equals the LP optimum and ANY exact
 *                            min-cost perfect matching solver returns the same
 *                            integer total.  This file uses a dense shortest
 *                            augmenting path solver (Jonker-Volgenant style)
 *                            in int64 and returns the dual potentials, so every
 *                            answer carries an LP-duality certificate.
 *   - objective evaluation   greedy_opt.py:21-29 (count_sum)
 *
 * Parity pinning: see oracle/README.md — pinned by the reference's own
 * known-answer instances (python.py:7, procedure.py:32-51, julia.jl:5) and by
 * an independent exact solver (scipy.optimize.linear_sum_assignment) in
 * tests/test_oracle.py.  Per-cab parity vs GLPK is unpinned where the optimum
 * is not unique (no GLPK output vector is committed anywhere in the reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* cost build                                                          */
/* ------------------------------------------------------------------ */

/*
 * Positional cost build (greedy_opt.py:86-99; with threshold simulate.py:17-33;
 * with id != -1 guard Simulator.java:508-511).
 *   n = max(n_s, n_d); cost[n][n] pre-filled with `fill`;
 *   cost[c][d] = dist[cab_to[c]][dem_from[d]]  (dist == NULL => |a-b|)
 *   only if threshold < 0 or value < threshold, and both ids != -1 when the
 *   id arrays are given.
 */
ORACLE_API int oracle_cost_build(const int32_t *cab_to, const int32_t *cab_id, int n_s,
                                 const int32_t *dem_from, const int32_t *dem_id, int n_d,
                                 const int32_t *dist, int S, int32_t fill, int32_t threshold,
                                 int32_t *cost /* n*n */)
{
    int n = n_s > n_d ? n_s : n_d;
    for (int64_t k = 0; k < (int64_t)n * n; k++) cost[k] = fill;
    for (int c = 0; c < n_s; c++) {
        if (cab_id && cab_id[c] == -1) continue;
        for (int d = 0; d < n_d; d++) {
            if (dem_id && dem_id[d] == -1) continue;
            int32_t a = cab_to[c], b = dem_from[d];
            int32_t v = dist ? dist[(int64_t)a * S + b] : (a > b ? a - b : b - a);
            if (threshold < 0 || v < threshold) cost[(int64_t)c * n + d] = v;
        }
    }
    return n;
}

/*
 * procedure.py:6-12 — fill is n*n and cells are addressed BY ID:
 *   cost[c_id][d_id] = distances[c_to][d_frm]
 */
ORACLE_API int oracle_cost_build_by_id(const int32_t *cab_id, const int32_t *cab_to, int n_s,
                                       const int32_t *dem_id, const int32_t *dem_from, int n_d,
                                       const int32_t *dist, int S, int32_t *cost)
{
    int n = n_s > n_d ? n_s : n_d;
    for (int64_t k = 0; k < (int64_t)n * n; k++) cost[k] = n * n;
    for (int c = 0; c < n_s; c++)
        for (int d = 0; d < n_d; d++) {
            int32_t a = cab_to[c], b = dem_from[d];
            int32_t v = dist ? dist[(int64_t)a * S + b] : (a > b ? a - b : b - a);
            cost[(int64_t)cab_id[c] * n + dem_id[d]] = v;
        }
    return n;
}

/* ------------------------------------------------------------------ */
/* LCM — lowest cost method                                            */
/* ------------------------------------------------------------------ */

/*
 * One routine covering the reference's variants.  Every iteration takes the
 * FIRST minimum in row-major order (numpy argmin: greedy_opt.py:67; Java
 * strict '<' scan: Simulator.java:531-537), records (row, col) and overwrites
 * that row and column with `mask`.
 *
 *   max_iter         number of iterations at most (n in every reference variant)
 *   threshold >= 0   stop BEFORE taking a cell whose value is > threshold
 *                    (greedy_opt.py:68-69, simulate.py:84-85); < 0 : no test
 *   stop_value_on    stop BEFORE taking when min == stop_value
 *                    (Simulator.java:538: LCM_min_val == big_cost)
 *   stop_size >= 0   stop AFTER taking when remaining size == stop_size
 *                    (Simulator.java:544-545: MAX_NON_LCM)
 *   sum_below        a taken cell adds to the total only if value < sum_below
 *                    (greedy_opt.py:74: d[elem] < big_cost); heuristic.py:27
 *                    sums everything -> pass INT32_MAX.
 *   java_scan        1: the running minimum starts at stop_value and only a
 *                    strictly smaller cell replaces it (Simulator.java:529-537)
 *                    so cells >= big_cost are never found; 0: plain argmin.
 * Returns the number of pairs; *last_min is the last minimum looked at
 * (Simulator.java's global LCM_min_val, read by main at :188).
 */
ORACLE_API int oracle_lcm(int n, const int32_t *cost, int32_t mask, int max_iter,
                          int32_t threshold, int stop_value_on, int32_t stop_value,
                          int stop_size, int64_t sum_below, int java_scan,
                          int32_t *rows, int32_t *cols, int64_t *total, int32_t *last_min)
{
    int32_t *d = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * n);
    memcpy(d, cost, sizeof(int32_t) * (size_t)n * n);
    int k = 0, size = n;
    int64_t tot = 0;
    int32_t lm = stop_value;
    for (int it = 0; it < max_iter && it < n; it++) {
        int64_t best = -1;
        int32_t bv = 0;
        if (java_scan) {
            bv = stop_value;
            for (int64_t e = 0; e < (int64_t)n * n; e++)
                if (d[e] < bv) { bv = d[e]; best = e; }
            lm = bv;
            if (bv == stop_value) break;
        } else {
            best = 0; bv = d[0];
            for (int64_t e = 1; e < (int64_t)n * n; e++)
                if (d[e] < bv) { bv = d[e]; best = e; }
            lm = bv;
            if (threshold >= 0 && bv > threshold) break;
            if (stop_value_on && bv == stop_value) break;
        }
        int r = (int)(best / n), c = (int)(best - (int64_t)r * n);
        rows[k] = r; cols[k] = c; k++;
        if ((int64_t)bv < sum_below) tot += bv;
        for (int j = 0; j < n; j++) { d[(int64_t)r * n + j] = mask; d[(int64_t)j * n + c] = mask; }
        size--;
        if (stop_size >= 0 && size == stop_size) break;
    }
    free(d);
    if (total) *total = tot;
    if (last_min) *last_min = lm;
    return k;
}

/* ------------------------------------------------------------------ */
/* exact min-cost perfect matching (what glpk.ilp's optimum equals)    */
/* ------------------------------------------------------------------ */

/*
 * Dense shortest-augmenting-path assignment in int64 with dual potentials.
 * Initialisation: row reduction + greedy tight matching; then one Dijkstra per
 * free row with the "ready / scan / todo" column partition (all columns at the
 * current minimum distance are scanned before a new minimum is searched), which
 * keeps degenerate (heavily tied) instances fast.
 * Output: row_to_col[n]; u[n], v[n] with u[i]+v[j] <= c[i][j] everywhere and
 * equality on matched cells  =>  sum(u)+sum(v) == total  (LP certificate).
 * Returns the optimal total.
 */
ORACLE_API int64_t oracle_assign(int n, const int32_t *c, int32_t *row_to_col,
                                 int64_t *u_out, int64_t *v_out)
{
    if (n <= 0) return 0;
    int64_t *u = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int64_t *v = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int64_t *dd = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int32_t *r2c = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *c2r = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *pred = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *collist = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; i++) { r2c[i] = -1; c2r[i] = -1; }

    /* row reduction + greedy on tight cells */
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n;
        int32_t m = ci[0];
        for (int j = 1; j < n; j++) if (ci[j] < m) m = ci[j];
        u[i] = m;
        for (int j = 0; j < n; j++)
            if (ci[j] == m && c2r[j] < 0) { c2r[j] = i; r2c[i] = j; break; }
    }

    for (int f = 0; f < n; f++) {
        if (r2c[f] >= 0) continue;
        /* Dijkstra from free row f over columns, reduced costs c - u - v >= 0 */
        int low = 0, up = 0; /* [0,low) ready, [low,up) scan, [up,n) todo */
        for (int j = 0; j < n; j++) {
            collist[j] = j;
            dd[j] = (int64_t)c[(int64_t)f * n + j] - u[f] - v[j];
            pred[j] = f;
        }
        int endcol = -1;
        int64_t mind = 0;
        while (endcol < 0) {
            if (low == up) {
                mind = dd[collist[up]];
                int t = up; up++;
                (void)t;
                for (int k = up; k < n; k++) {
                    int j = collist[k];
                    int64_t h = dd[j];
                    if (h <= mind) {
                        if (h < mind) { up = low; mind = h; }
                        collist[k] = collist[up]; collist[up] = j; up++;
                    }
                }
                for (int k = low; k < up; k++)
                    if (c2r[collist[k]] < 0) { endcol = collist[k]; break; }
            }
            if (endcol >= 0) break;
            int j1 = collist[low]; low++;
            int i = c2r[j1];
            const int32_t *ci = c + (int64_t)i * n;
            int64_t base = mind - ((int64_t)ci[j1] - u[i] - v[j1]); /* = mind, since tight */
            for (int k = up; k < n; k++) {
                int j = collist[k];
                int64_t h = (int64_t)ci[j] - u[i] - v[j] + base;
                if (h < dd[j]) {
                    dd[j] = h; pred[j] = i;
                    if (h == mind) {
                        if (c2r[j] < 0) { endcol = j; break; }
                        collist[k] = collist[up]; collist[up] = j; up++;
                    }
                }
            }
        }
        /* dual update on ready columns */
        for (int k = 0; k < low; k++) {
            int j = collist[k];
            int i = c2r[j];
            int64_t delta = mind - dd[j];
            v[j] -= delta;
            u[i] += delta;
        }
        u[f] += mind;
        /* augment */
        int j = endcol;
        for (;;) {
            int i = pred[j];
            c2r[j] = i;
            int t = r2c[i]; r2c[i] = j; j = t;
            if (i == f) break;
        }
    }
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        total += c[(int64_t)i * n + r2c[i]];
        if (row_to_col) row_to_col[i] = r2c[i];
    }
    if (u_out) memcpy(u_out, u, sizeof(int64_t) * (size_t)n);
    if (v_out) memcpy(v_out, v, sizeof(int64_t) * (size_t)n);
    free(u); free(v); free(dd); free(r2c); free(c2r); free(pred); free(collist);
    return total;
}

/*
 * LP-duality check for an (assignment, potentials) pair:
 *   feasibility  u[i]+v[j] <= c[i][j] for all cells  (returns -1 if violated)
 *   returns sum(u)+sum(v) through *dual; primal through *primal.
 * primal == dual  <=>  the assignment is optimal (and so equals GLPK's value).
 */
ORACLE_API int oracle_certificate(int n, const int32_t *c, const int32_t *row_to_col,
                                  const int64_t *u, const int64_t *v,
                                  int64_t *primal, int64_t *dual)
{
    int64_t p = 0, d = 0;
    char *seen = (char *)calloc((size_t)n, 1);
    int ok = 1;
    for (int i = 0; i < n; i++) {
        int j = row_to_col[i];
        if (j < 0 || j >= n || seen[j]) { ok = 0; break; }
        seen[j] = 1;
        p += c[(int64_t)i * n + j];
    }
    free(seen);
    if (!ok) return -2; /* not a permutation */
    for (int i = 0; i < n; i++) d += u[i];
    for (int j = 0; j < n; j++) d += v[j];
    for (int i = 0; i < n && ok; i++)
        for (int j = 0; j < n; j++)
            if (u[i] + v[j] > (int64_t)c[(int64_t)i * n + j]) { ok = 0; break; }
    if (primal) *primal = p;
    if (dual) *dual = d;
    return ok ? 0 : -1;
}

/*
 * Dual lower bound from column prices alone (what the HIP auction reports):
 *   D(p) = sum_i min_j (c[i][j] - p[j]) + sum_j p[j]   <= OPT for ANY p.
 * `scale` lets prices be expressed in units of 1/scale (the auction works on
 * scale*c); the bound returned is floor-free: it is in scaled units.
 */
ORACLE_API int64_t oracle_dual_bound_scaled(int n, const int32_t *c, const int64_t *price,
                                            int64_t scale)
{
    int64_t d = 0;
    for (int i = 0; i < n; i++) {
        int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) {
            int64_t w = (int64_t)c[(int64_t)i * n + j] * scale + price[j];
            if (w < m) m = w;
        }
        d += m;
    }
    for (int j = 0; j < n; j++) d -= price[j];
    return d;
}

/*
 * Uniqueness of the optimum.  Under optimal potentials (u, v) every optimal
 * assignment uses only tight cells (c - u - v == 0).  Orient matched cells
 * col->row and unmatched tight cells row->col: another optimal assignment
 * exists iff this digraph has a directed cycle.  Returns 1 if unique, 0 if
 * not.  (Iterative DFS, O(n^2).)
 */
ORACLE_API int oracle_is_unique(int n, const int32_t *c, const int32_t *row_to_col,
                                const int64_t *u, const int64_t *v)
{
    /* contract each matched (row i, col r2c[i]) into node i; edge i -> k iff
       cell (i, r2c[k]) is tight and k != i.  Cycle <=> alternative optimum. */
    int8_t *state = (int8_t *)calloc((size_t)n, 1); /* 0 new, 1 on stack, 2 done */
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *iter = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *c2r = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; i++) c2r[row_to_col[i]] = i;
    int unique = 1;
    for (int s = 0; s < n && unique; s++) {
        if (state[s]) continue;
        int sp = 0;
        stack[sp++] = s; state[s] = 1;
        while (sp && unique) {
            int i = stack[sp - 1];
            int advanced = 0;
            while (iter[i] < n) {
                int j = iter[i]++;
                int k = c2r[j];
                if (k == i) continue;
                if ((int64_t)c[(int64_t)i * n + j] - u[i] - v[j] != 0) continue;
                if (state[k] == 1) { unique = 0; break; }
                if (state[k] == 0) { state[k] = 1; stack[sp++] = k; advanced = 1; break; }
            }
            if (!unique) break;
            if (!advanced) { state[i] = 2; sp--; }
        }
    }
    free(state); free(stack); free(iter); free(c2r);
    return unique;
}

/* ------------------------------------------------------------------ */
/* objective evaluation  (greedy_opt.py:21-29 count_sum)               */
/* ------------------------------------------------------------------ */
ORACLE_API int64_t oracle_count_sum(int n, const int32_t *cost, const int32_t *row_to_col,
                                    int64_t big_cost, int32_t *n_real)
{
    int64_t s = 0;
    int32_t k = 0;
    for (int i = 0; i < n; i++) {
        int j = row_to_col[i];
        if (j < 0) continue;
        int64_t cv = cost[(int64_t)i * n + j];
        if (cv < big_cost) { s += cv; k++; }
    }
    if (n_real) *n_real = k;
    return s;
}

/* ------------------------------------------------------------------ */
/* synthetic instance generator shared with the device (bench G1)      */
/* ------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* perf.jl:5  t = rand(10:40, n, n)  — uniform ints lo..hi inclusive, counter-based */
ORACLE_API void oracle_gen_uniform(int n, uint64_t seed, int32_t lo, int32_t hi,
                                   int row0, int nrows, int32_t *cost)
{
    uint32_t span = (uint32_t)(hi - lo + 1);
    for (int i = 0; i < nrows; i++)
        for (int j = 0; j < n; j++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + (uint64_t)(row0 + i) * (uint64_t)n + (uint64_t)j);
            cost[(int64_t)i * n + j] = lo + (int32_t)(((h >> 32) * (uint64_t)span) >> 32);
        }
}

/* ------------------------------------------------------------------ */
/* f-4: pools of up to 4 passengers (pool_n.c:101-207, Pool.java:32-113) */
/* ------------------------------------------------------------------ */
/* Own restatement, iterative.  A plan = an ordered pick-up sequence p[0..k) of distinct requests
 * (first pick-up in [first0, first1): findpool.c:138-141 gives each of its 8 children such a
 * slice, pool_n.c:243-246) and a drop-off order q = a permutation of 0..k-1 into p.
 *   wait rule (pool_n.c:174-179): for every level l >= 1 the pick-up path p[0]->..->p[l] must not be
 *     longer than WAIT of p[l]; a failing candidate is skipped, the enumeration goes on.
 *   happiness (pool_n.c:105-122): passenger x = p[q[d]] rides  pick-up path from its own pick-up to
 *     the last one + last pick-up -> first drop-off + drop-off path up to its own drop-off; that
 *     must not exceed  direct(x) * (1 + LOSS(x) / 100.0)  (double arithmetic, `>` rejects).
 *   cost of a plan (pool_n.c:129-137): whole pick-up path + last pick-up -> first drop + drop path.
 * Plans are recorded in enumeration order (p lexicographic with the level-0 slice, then q
 * lexicographic), sorted by cost — STABLY (glibc's qsort is a merge sort for arrays of this size,
 * so equal costs keep the enumeration order; pool_n.c:198) — and a plan is dropped when it shares
 * a request with an earlier kept plan (pool_n.c:200-215).  Output records: p[0..k), p[q[0..k)], cost.
 * dist == NULL means |a-b| (pool_n.c:186-192).  Returns the number of kept plans, or -1 when more
 * than `cap` happy plans exist (the reference's pool[MAX_ARR] would overflow there). */
static inline int pool_dist(const int32_t *dist, int S, int a, int b)
{
    return dist ? dist[(int64_t)a * S + b] : (a > b ? a - b : b - a);
}
typedef struct { int32_t r[9]; int64_t seq; } pool_rec;
static void pool_merge_sort(pool_rec *a, pool_rec *tmp, long n)
{   /* stable merge sort by r[8] (cost) */
    if (n < 2) return;
    long h = n / 2;
    pool_merge_sort(a, tmp, h);
    pool_merge_sort(a + h, tmp, n - h);
    long i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].r[8] < a[i].r[8]) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, sizeof(pool_rec) * (size_t)n);
}
ORACLE_API long oracle_pool_n(int k, int n, const int32_t *from, const int32_t *to, const int32_t *wait,
                              const int32_t *loss, const int32_t *dist, int S, int first0, int first1,
                              long cap, int32_t *out /* cap * (2k+1) */, long *n_happy)
{
    if (k < 1 || k > 4 || n < k) { if (n_happy) *n_happy = 0; return 0; }
    pool_rec *rec = malloc(sizeof(pool_rec) * (size_t)(cap > 0 ? cap : 1));
    long cnt = 0;
    int p[4] = {0, 0, 0, 0}, lim[4];
    int perm[24][4], np = 0;
    {   /* permutations of 0..k-1 in lexicographic order (the recursion order of pool_n.c:140-153) */
        int q[4];
        for (q[0] = 0; q[0] < k; q[0]++)
            for (q[1] = 0; q[1] < (k > 1 ? k : 1); q[1]++)
                for (q[2] = 0; q[2] < (k > 2 ? k : 1); q[2]++)
                    for (q[3] = 0; q[3] < (k > 3 ? k : 1); q[3]++) {
                        int ok = 1;
                        for (int a = 0; a < k && ok; a++)
                            for (int b = a + 1; b < k; b++)
                                if (q[a] == q[b]) { ok = 0; break; }
                        if (!ok) continue;
                        for (int a = 0; a < 4; a++) perm[np][a] = a < k ? q[a] : 0;
                        np++;
                    }
    }
    int overflow = 0;
    /* odometer over the pick-up sequence; level 0 runs over the slice, deeper levels over all requests */
    int level = 0;
    p[0] = first0 - 1;
    lim[0] = first1;
    while (level >= 0) {
        p[level]++;
        if (p[level] >= (level == 0 ? lim[0] : n)) { level--; continue; }
        int dup = 0;
        for (int l = 0; l < level; l++) if (p[l] == p[level]) { dup = 1; break; }
        if (dup) continue;
        int path = 0;
        for (int l = 0; l < level; l++) path += pool_dist(dist, S, from[p[l]], from[p[l + 1]]);
        if (path > wait[p[level]]) continue;
        if (level + 1 < k) { level++; p[level] = -1; continue; }
        /* a complete pick-up sequence: all drop-off orders */
        for (int qi = 0; qi < np; qi++) {
            const int *q = perm[qi];
            int happy = 1;
            for (int d = 0; d < k && happy; d++) {
                int c = 0;
                for (int ph = q[d]; ph < k - 1; ph++) c += pool_dist(dist, S, from[p[ph]], from[p[ph + 1]]);
                c += pool_dist(dist, S, from[p[k - 1]], to[p[q[0]]]);
                for (int ph = 0; ph < d; ph++) c += pool_dist(dist, S, to[p[q[ph]]], to[p[q[ph + 1]]]);
                const int x = p[q[d]];
                if (c > pool_dist(dist, S, from[x], to[x]) * (1 + loss[x] / 100.0)) happy = 0;
            }
            if (!happy) continue;
            if (cnt >= cap) { overflow = 1; cnt++; continue; }
            pool_rec *r = &rec[cnt++];
            memset(r->r, 0, sizeof(r->r));
            for (int i = 0; i < k; i++) { r->r[i] = p[i]; r->r[i + k] = p[q[i]]; }
            int c = 0;
            for (int i = 0; i < k - 1; i++) c += pool_dist(dist, S, from[p[i]], from[p[i + 1]]);
            c += pool_dist(dist, S, from[p[k - 1]], to[p[q[0]]]);
            for (int i = 0; i < k - 1; i++) c += pool_dist(dist, S, to[p[q[i]]], to[p[q[i + 1]]]);
            r->r[8] = c;
        }
    }
    if (n_happy) *n_happy = cnt;
    if (overflow) { free(rec); return -1; }
    pool_rec *tmp = malloc(sizeof(pool_rec) * (size_t)(cnt > 0 ? cnt : 1));
    pool_merge_sort(rec, tmp, cnt);
    free(tmp);
    char *used = calloc((size_t)n, 1);
    long kept = 0;
    for (long i = 0; i < cnt; i++) {
        int clash = 0;
        for (int a = 0; a < k; a++) if (used[rec[i].r[a]]) { clash = 1; break; }
        if (clash) continue;
        for (int a = 0; a < k; a++) used[rec[i].r[a]] = 1;
        for (int a = 0; a < 2 * k; a++) out[kept * (2 * k + 1) + a] = rec[i].r[a];
        out[kept * (2 * k + 1) + 2 * k] = rec[i].r[8];
        kept++;
    }
    free(used);
    free(rec);
    return kept;
}
