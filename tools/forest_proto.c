/* Exploration model #3 (dev tool): eps=0 Jacobi rounds, then PHASES of a multi-source shortest-path
 * forest (all free rows are roots, one common distance), one augmentation per tree that reaches a
 * free column at the end distance.  Counts phases / levels / row scans — the quantities that set
 * the GPU time (a level is a grid-wide step, a row scan is 4n bytes of HBM traffic).
 * build: gcc -O3 -fopenmp -o /tmp/forest_proto tools/forest_proto.c
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static int32_t *gen(const char *kind, int n, uint64_t seed)
{
    int32_t *c = malloc(sizeof(int32_t) * (size_t)n * n);
    if (!strcmp(kind, "g1") || !strcmp(kind, "g4") || !strcmp(kind, "wide")) {
        uint64_t lo = !strcmp(kind, "g1") ? 10 : (!strcmp(kind, "g4") ? 1 : 0);
        uint64_t span = !strcmp(kind, "g1") ? 31 : (!strcmp(kind, "g4") ? 39 : 1000000);
        for (int64_t k = 0; k < (int64_t)n * n; k++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + k);
            c[k] = (int32_t)(lo + (((h >> 32) * span) >> 32));
        }
    } else if (!strcmp(kind, "g2") || !strcmp(kind, "g3") || !strcmp(kind, "g2d")) {
        int g3 = !strcmp(kind, "g3");
        int S = g3 ? 50 : 10 * n;
        int nd = g3 ? (int)(n * 0.363) : n;
        int32_t *a = malloc(4 * n), *b = malloc(4 * n);
        for (int i = 0; i < n; i++) {
            a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)S);
            b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)S);
        }
        int two = !strcmp(kind, "g2d"); /* 2-D manhattan on a sqrt grid */
        int W = 1; while (W * W < S) W++;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int32_t v = abs(a[i] - b[j]);
                if (two) v = abs(a[i] % W - b[j] % W) + abs(a[i] / W - b[j] / W);
                if (g3) v = (j < nd && v < 10) ? v : 250000;
                c[(int64_t)i * n + j] = v;
            }
        free(a); free(b);
    } else { fprintf(stderr, "kind?\n"); exit(1); }
    return c;
}

static int n, shift;
static const int32_t *c;
#define CS(x) ((int64_t)((x) >> shift))
static int64_t *p;          /* column prices (>= 0) */
static int64_t *u;          /* row duals: u_i = min_j c_ij + p_j at CS */
static int32_t *r2c, *owner;
static long st_rounds, st_scans, st_levels, st_phases, st_augs;

static int32_t *list, *pick; static int64_t *bidv; static int32_t *bidr;
static int jacobi_round(int U, int64_t eps)
{
    st_rounds++; st_scans += U;
    for (int j = 0; j < n; j++) bidr[j] = -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < U; t++) {
        int i = list[t];
        const int32_t *ci = c + (int64_t)i * n;
        int off = (int)(splitmix64(i * 0x9E37ull + 12345 + st_rounds) % (uint64_t)n);
        int64_t k1 = INT64_MAX, k2 = INT64_MAX; int j1 = -1;
        for (int s = 0; s < n; s++) {
            int j = s + off; if (j >= n) j -= n;
            int64_t k = 2 * (CS(ci[j]) + p[j]) + (owner[j] >= 0);
            if (k < k1) { k2 = k1; k1 = k; j1 = j; } else if (k < k2) k2 = k;
        }
        int64_t w1 = k1 >> 1, w2 = (n == 1) ? w1 : (k2 >> 1);
        int64_t inc = w2 - w1 + eps;
        pick[t] = j1;
        bidv[i] = p[j1] + inc;
    }
    for (int t = 0; t < U; t++) {
        int i = list[t], j = pick[t];
        if (j < 0) continue;
        if (bidr[j] < 0 || bidv[i] > bidv[bidr[j]] || (bidv[i] == bidv[bidr[j]] && i > bidr[j])) bidr[j] = i;
    }
    for (int t = 0; t < U; t++) {
        int i = list[t], j = pick[t];
        if (j >= 0 && bidr[j] == i) {
            int o = owner[j];
            if (o >= 0) r2c[o] = -1;
            owner[j] = i; r2c[i] = j; p[j] = bidv[i];
        }
    }
    int U2 = 0;
    for (int i = 0; i < n; i++) if (r2c[i] < 0) list[U2++] = i;
    return U2;
}

/* serial reference */
static int64_t lap_serial(void)
{
    int64_t *pp = calloc(n, 8), *uu = calloc(n, 8), *dd = malloc(8 * n);
    int32_t *rc = malloc(4 * n), *ow = malloc(4 * n), *pred = malloc(4 * n), *cl = malloc(4 * n);
    for (int i = 0; i < n; i++) rc[i] = ow[i] = -1;
    for (int i = 0; i < n; i++) { const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX; for (int j = 0; j < n; j++) if (ci[j] < m) m = ci[j]; uu[i] = m; }
    for (int f = 0; f < n; f++) {
        int low = 0, up = 0, endcol = -1; int64_t mind = 0;
        for (int j = 0; j < n; j++) { cl[j] = j; dd[j] = c[(int64_t)f * n + j] + pp[j] - uu[f]; pred[j] = f; }
        while (endcol < 0) {
            if (low == up) {
                mind = dd[cl[up]]; up++;
                for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = dd[j];
                    if (h <= mind) { if (h < mind) { up = low; mind = h; } cl[k] = cl[up]; cl[up] = j; up++; } }
                for (int k = low; k < up; k++) if (ow[cl[k]] < 0) { endcol = cl[k]; break; }
            }
            if (endcol >= 0) break;
            int j1 = cl[low]; low++; int i = ow[j1]; const int32_t *ci = c + (int64_t)i * n;
            for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = ci[j] + pp[j] - uu[i] + mind;
                if (h < dd[j]) { dd[j] = h; pred[j] = i;
                    if (h == mind) { if (ow[j] < 0) { endcol = j; break; } cl[k] = cl[up]; cl[up] = j; up++; } } }
        }
        for (int k = 0; k < low; k++) { int j = cl[k]; int i = ow[j]; int64_t d = mind - dd[j]; pp[j] += d; uu[i] += d; }
        uu[f] += mind;
        int j = endcol; for (;;) { int i = pred[j]; ow[j] = i; int t = rc[i]; rc[i] = j; j = t; if (i == f) break; }
    }
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + rc[i]];
    free(pp); free(uu); free(dd); free(rc); free(ow); free(pred); free(cl);
    return tot;
}

/* Phases of a multi-source shortest path forest.
 * quant: distances are compared after >> qshift (0 = exact).  Exactness needs qshift == 0 in the
 * last phases; a coarse quantum makes many trees end in the same level (bigger batches). */
static int verbose, mode_all, allow_untight, pre_rounds; static int32_t *untight; static double frac_done = 1.0;
static int g_stop_free;
static void forest_phases(int window_mode);
static void forest_phases_until(int sf) { g_stop_free = sf; forest_phases(0); g_stop_free = 0; }
static void forest_phases(int window_mode)
{
    int64_t *slack = malloc(8 * n), *drow = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *fin = malloc(4 * n), *front = malloc(4 * n), *inforest = malloc(4 * n);
    int32_t *root_of = malloc(4 * n), *root_done = malloc(4 * n), *ends = malloc(4 * n);
    /* u from prices */
    st_scans += n;
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; }
        u[i] = m;
        if (r2c[i] >= 0 && CS(ci[r2c[i]]) + p[r2c[i]] != m) { if (!allow_untight) { fprintf(stderr, "non-tight match row %d\n", i); exit(2); } untight[i] = 1; } else untight[i] = 0;
    }
    { int nu = 0; for (int i = 0; i < n; i++) if (untight[i]) { owner[r2c[i]] = -1; r2c[i] = -1; nu++; } if (verbose) printf("  shift=%d untight=%d\n", shift, nu); }
    if (pre_rounds) { int U = 0; for (int i = 0; i < n; i++) if (r2c[i] < 0) list[U++] = i; int r = 0; while (U > 0 && r < pre_rounds) { U = jacobi_round(U, 0); r++; }
#pragma omp parallel for
      for (int i = 0; i < n; i++) { const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX; for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; } u[i] = m; }
      st_scans += n; if (verbose) printf("  after %d pre-rounds free=%d\n", r, U); }
    for (;;) {
        int nf = 0;
        for (int i = 0; i < n; i++) { inforest[i] = 0; if (r2c[i] < 0) { front[nf++] = i; drow[i] = 0; inforest[i] = 1; root_of[i] = i; } }
        if (nf <= g_stop_free) break;
        st_phases++;
        int nfree0 = nf;
        for (int j = 0; j < n; j++) { slack[j] = INT64_MAX; pred[j] = -1; fin[j] = 0; }
        int64_t delta = 0; int nends = 0; long levels = 0, scans = 0; int ends_seen = 0, trees_done = 0;
        for (int i = 0; i < n; i++) root_done[i] = 0;
        for (;;) {
            /* relax all frontier rows */
            scans += nf; levels++;
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; j++) {
                if (fin[j]) continue;
                int64_t s = slack[j]; int pr = pred[j];
                for (int t = 0; t < nf; t++) {
                    int i = front[t];
                    int64_t h = drow[i] + CS(c[(int64_t)i * n + j]) + p[j] - u[i];
                    if (h < s) { s = h; pr = i; }
                }
                slack[j] = s; pred[j] = pr;
            }
            /* next level */
            int64_t m = INT64_MAX;
            for (int j = 0; j < n; j++) if (!fin[j] && slack[j] < m) m = slack[j];
            if (m == INT64_MAX) { if (mode_all && nends) break; fprintf(stderr, "no path\n"); exit(3); }
            delta = m; nf = 0;
            for (int j = 0; j < n; j++) if (!fin[j] && slack[j] == m) {
                fin[j] = 1;
                if (owner[j] < 0) ends[nends++] = j;
                else { int i = owner[j]; if (inforest[i]) { fprintf(stderr, "row twice\n"); exit(4); } inforest[i] = 1; drow[i] = m; root_of[i] = root_of[pred[j]]; front[nf++] = i; }
            }
            if (!mode_all) { if (nends) break; }
            else {
                for (int e = ends_seen; e < nends; e++) { int r = root_of[pred[ends[e]]]; if (!root_done[r]) { root_done[r] = 1; trees_done++; } }
                ends_seen = nends;
                if (trees_done >= (int)(nfree0 * frac_done + 0.999) || nf == 0 && 0) break;
            }
        }
        st_levels += levels; st_scans += scans;
        /* dual update: finalised columns (slack <= delta) and forest rows */
        for (int j = 0; j < n; j++) if (fin[j]) p[j] += delta - slack[j];
        for (int i = 0; i < n; i++) if (inforest[i]) u[i] += delta - drow[i];
        /* one augmentation per tree */
        for (int i = 0; i < n; i++) root_done[i] = 0;
        int augs = 0;
        for (int e = 0; e < nends; e++) {
            int j = ends[e]; int r = root_of[pred[j]];
            if (root_done[r]) continue;
            root_done[r] = 1; augs++;
            for (;;) { int i = pred[j]; owner[j] = i; int t = r2c[i]; r2c[i] = j; j = t; if (t < 0) break; }
        }
        st_augs += augs;
        if (verbose) printf("   phase %ld: free=%d levels=%ld scans=%ld delta=%ld ends=%d augs=%d\n", st_phases, nfree0, levels, scans, (long)delta, nends, augs);
    }
    free(slack); free(drow); free(pred); free(fin); free(front); free(inforest); free(root_of); free(root_done); free(ends);
}


static int cmp64(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : x > y; }
/* windowed label-correcting variant: a step closes the B smallest open owned labels at once */
static void forest_phases_win(int B)
{
    int64_t *slack = malloc(8 * n), *drow = malloc(8 * n), *tmp = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *closed = malloc(4 * n), *front = malloc(4 * n), *root_done = malloc(4 * n);
    st_scans += n;
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; }
        u[i] = m;
    }
    for (;;) {
        int nf = 0;
        for (int i = 0; i < n; i++) if (r2c[i] < 0) { front[nf++] = i; drow[i] = 0; }
        if (!nf) break;
        st_phases++;
        int nfree0 = nf;
        for (int j = 0; j < n; j++) { slack[j] = INT64_MAX; pred[j] = -1; closed[j] = 0; }
        long levels = 0, scans = 0; int64_t delta;
        for (;;) {
            scans += nf; levels++;
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; j++) {
                int64_t s = slack[j]; int pr = pred[j];
                for (int t = 0; t < nf; t++) {
                    int i = front[t];
                    int64_t h = drow[i] + CS(c[(int64_t)i * n + j]) + p[j] - u[i];
                    if (h < s) { s = h; pr = i; }
                }
                if (s < slack[j]) { slack[j] = s; pred[j] = pr; closed[j] = 0; }
            }
            int64_t mfree = INT64_MAX;
            for (int j = 0; j < n; j++) if (owner[j] < 0 && slack[j] < mfree) mfree = slack[j];
            int no = 0;
            for (int j = 0; j < n; j++) if (owner[j] >= 0 && !closed[j] && slack[j] < mfree) tmp[no++] = slack[j];
            if (!no) { delta = mfree; break; }
            int64_t thr;
            if (no <= B) thr = INT64_MAX; else { qsort(tmp, no, 8, cmp64); thr = tmp[B - 1]; }
            nf = 0;
            for (int j = 0; j < n; j++) if (owner[j] >= 0 && !closed[j] && slack[j] < mfree && slack[j] <= thr) { closed[j] = 1; drow[owner[j]] = slack[j]; front[nf++] = owner[j]; }
        }
        st_levels += levels; st_scans += scans;
        for (int j = 0; j < n; j++) if (owner[j] >= 0 && slack[j] < delta) { int64_t d = delta - slack[j]; p[j] += d; u[owner[j]] += d; }
        for (int i = 0; i < n; i++) if (r2c[i] < 0) u[i] += delta;
        for (int i = 0; i < n; i++) root_done[i] = r2c[i] < 0 ? 0 : -1; /* -1: not a root */
        int augs = 0, nends = 0;
        for (int e = 0; e < n; e++) { tmp[e] = -1; if (owner[e] < 0 && slack[e] == delta) {
            nends++;
            int j = e, i; for (;;) { i = pred[j]; if (root_done[i] >= 0) break; j = r2c[i]; }
            if (root_done[i]) continue;
            root_done[i] = 1; tmp[e] = i; } }
        for (int e = 0; e < n; e++) if (tmp[e] >= 0) {
            augs++;
            int j = e; for (;;) { int ii = pred[j]; owner[j] = ii; int t = r2c[ii]; r2c[ii] = j; j = t; if (t < 0) break; }
        }
        st_augs += augs;
        if (verbose) printf("   phase %ld: free=%d steps=%ld scans=%ld delta=%ld ends=%d augs=%d\n", st_phases, nfree0, levels, scans, (long)delta, nends, augs);
    }
}

/* Incremental forest: ONE continuous multi-source Dijkstra; a tree that reaches a free column is
 * augmented and released at once (its lazy dual raises are materialised), the other trees stay.
 * Labels of columns that lose their best row are recomputed over the remaining forest rows
 * (column scans: needs the transposed matrix on the GPU). */
static long st_repairs, st_rowjoins, st_repair_entries, st_join_entries;
static uint8_t *incore; /* n*n bytes, NULL = dense */
#define INFC ((int64_t)1 << 50)
#define COSTC(i, j) ((incore && !incore[(int64_t)(i) * n + (j)]) ? INFC : CS(c[(int64_t)(i) * n + (j)]))
static int g_stuck;
static void forest_incremental(int stop_free)
{
    int64_t *slack = malloc(8 * n), *arow = malloc(8 * n), *acol = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *inFr = calloc(n, 4), *inFc = calloc(n, 4), *rootr = malloc(4 * n), *rootc = malloc(4 * n);
    int32_t *newrows = malloc(4 * n), *ends = malloc(4 * n), *rel = calloc(n, 4), *need = calloc(n, 4);
    st_scans += n;
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = COSTC(i, j) + p[j]; if (w < m) m = w; }
        (void)ci; u[i] = m;
    }
    int nfree = 0, nn = 0;
    for (int i = 0; i < n; i++) if (r2c[i] < 0) { nfree++; inFr[i] = 1; arow[i] = 0; rootr[i] = i; newrows[nn++] = i; }
    for (int j = 0; j < n; j++) { slack[j] = INT64_MAX; pred[j] = -1; }
    int64_t D = 0; long levels = 0;
    while (nfree > stop_free) {
        /* relax the rows that joined */
        st_rowjoins += nn; st_scans += nn;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < n; j++) {
            if (inFc[j]) continue;
            int64_t s = slack[j]; int pr = pred[j];
            for (int t = 0; t < nn; t++) { int i = newrows[t]; int64_t cc_ = COSTC(i, j); if (cc_ >= INFC) continue; int64_t h = arow[i] + cc_ + p[j] - u[i]; if (h < s) { s = h; pr = i; } }
            slack[j] = s; pred[j] = pr;
        }
        levels++;
        int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) if (!inFc[j] && slack[j] < m) m = slack[j];
        if (m == INT64_MAX) { printf("   STUCK with %d free rows\n", nfree); g_stuck = 1; break; }
        if (m < D) { fprintf(stderr, "label below D\n"); exit(6); }
        D = m; nn = 0; int ne = 0;
        for (int j = 0; j < n; j++) if (!inFc[j] && slack[j] == D) {
            rootc[j] = rootr[pred[j]];
            if (owner[j] < 0) ends[ne++] = j;
            else { int i = owner[j]; inFc[j] = 1; acol[j] = D; inFr[i] = 1; arow[i] = D; rootr[i] = rootc[j]; newrows[nn++] = i; }
        }
        if (!ne) continue;
        /* one end per tree */
        int nrel = 0;
        for (int e = 0; e < ne; e++) { int r = rootc[ends[e]]; if (rel[r]) { ends[e] = -1; continue; } rel[r] = 1; nrel++; }
        for (int e = 0; e < ne; e++) if (ends[e] >= 0) {
            int j = ends[e]; for (;;) { int ii = pred[j]; owner[j] = ii; int t = r2c[ii]; r2c[ii] = j; j = t; if (t < 0) break; }
            st_augs++; nfree--;
        }
        /* release the augmented trees */
        for (int i = 0; i < n; i++) if (inFr[i] && rel[rootr[i]]) { u[i] += D - arow[i]; inFr[i] = 0; }
        for (int j = 0; j < n; j++) if (inFc[j] && rel[rootc[j]]) { p[j] += D - acol[j]; inFc[j] = 0; need[j] = 1; }
        for (int j = 0; j < n; j++) if (!inFc[j] && !need[j] && pred[j] >= 0 && !inFr[pred[j]]) need[j] = 1;
        { int k = 0; for (int t = 0; t < nn; t++) if (inFr[newrows[t]]) newrows[k++] = newrows[t]; nn = k; }
        /* repair labels over the remaining forest rows (the new rows are relaxed next round anyway, include them now) */
        long rep = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+:rep)
        for (int j = 0; j < n; j++) if (need[j]) {
            rep++;
            int64_t s = INT64_MAX; int pr = -1;
            for (int i = 0; i < n; i++) if (inFr[i]) { int64_t cc_ = COSTC(i, j); if (cc_ >= INFC) continue; int64_t h = arow[i] + cc_ + p[j] - u[i]; if (h < s) { s = h; pr = i; } }
            slack[j] = s; pred[j] = pr; need[j] = 0;
        }
        st_repairs += rep;
        for (int e = 0; e < ne; e++) if (ends[e] >= 0) rel[rootc[ends[e]]] = 0;
        for (int i = 0; i < n; i++) rel[i] = 0;
        if (verbose) printf("   D=%ld level %ld: free=%d released %d trees, repaired %ld cols\n", (long)D, levels, nfree, nrel, rep);
    }
    /* materialise what is left of the forest */
    for (int i = 0; i < n; i++) if (inFr[i]) { u[i] += D - arow[i]; inFr[i] = 0; }
    for (int j = 0; j < n; j++) if (inFc[j]) { p[j] += D - acol[j]; inFc[j] = 0; }
    st_levels += levels;
}

static int cmp64b(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : x > y; }
static void core_solve(int K)
{
    incore = calloc((size_t)n * n, 1);
    long entries = 0;
#pragma omp parallel
    {
        int64_t *w = malloc(8 * n);
#pragma omp for reduction(+:entries)
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) w[j] = CS(c[(int64_t)i * n + j]) + p[j];
            int64_t *t = malloc(8 * n); memcpy(t, w, 8 * n); qsort(t, n, 8, cmp64b); int64_t thr = t[K - 1 < n ? K - 1 : n - 1]; free(t);
            int cnt = 0;
            for (int j = 0; j < n; j++) if (w[j] < thr) { incore[(int64_t)i * n + j] = 1; cnt++; }
            for (int j = 0; j < n && cnt < K; j++) if (w[j] == thr) { incore[(int64_t)i * n + j] = 1; cnt++; }
            if (r2c[i] >= 0 && !incore[(int64_t)i * n + r2c[i]]) { incore[(int64_t)i * n + r2c[i]] = 1; cnt++; }
            entries += cnt;
        }
        free(w);
    }
    printf("  core K=%d entries/row=%.1f\n", K, (double)entries / n);
    for (int it = 0; it < 50; it++) {
        long l0 = st_levels, j0 = st_rowjoins, r0 = st_repairs;
        g_stuck = 0;
        forest_incremental(0);
        if (g_stuck) {   /* no augmenting path inside the core: widen every row by K more entries */
            printf("  iter %d stuck: widening\n", it);
#pragma omp parallel for
            for (int i = 0; i < n; i++) {
                int64_t *t = malloc(8 * n); int m = 0;
                for (int j = 0; j < n; j++) if (!incore[(int64_t)i * n + j]) t[m++] = CS(c[(int64_t)i * n + j]) + p[j];
                if (m) { qsort(t, m, 8, cmp64b); int64_t thr = t[K - 1 < m ? K - 1 : m - 1];
                    for (int j = 0; j < n; j++) if (!incore[(int64_t)i * n + j] && CS(c[(int64_t)i * n + j]) + p[j] <= thr) incore[(int64_t)i * n + j] = 1; }
                free(t);
            }
            for (int i = 0; i < n; i++) { int64_t m = INT64_MAX; for (int j = 0; j < n; j++) if (incore[(int64_t)i * n + j]) { int64_t w = CS(c[(int64_t)i * n + j]) + p[j]; if (w < m) m = w; }
                int j = r2c[i]; if (j >= 0 && CS(c[(int64_t)i * n + j]) + p[j] != m) { owner[j] = -1; r2c[i] = -1; } }
            continue;
        }
        long viol = 0, vrows = 0, unm = 0;
        for (int i = 0; i < n; i++) {
            int64_t m = u[i]; int any = 0;
            for (int j = 0; j < n; j++) if (!incore[(int64_t)i * n + j]) { int64_t w = CS(c[(int64_t)i * n + j]) + p[j]; if (w < u[i]) { incore[(int64_t)i * n + j] = 1; viol++; any = 1; if (w < m) m = w; } }
            if (any) { vrows++; u[i] = m; int j = r2c[i]; if (j >= 0 && CS(c[(int64_t)i * n + j]) + p[j] != m) { owner[j] = -1; r2c[i] = -1; unm++; } }
        }
        printf("  iter %d: levels=%ld rowjoins/n=%.2f repairs/n=%.2f | violations=%ld in %ld rows, unmatched %ld\n", it, st_levels - l0, (double)(st_rowjoins - j0) / n, (double)(st_repairs - r0) / n, viol, vrows, unm);
        if (!viol) break;
    }
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "g2";
    n = argc > 2 ? atoi(argv[2]) : 1000;
    int arr_rounds = argc > 3 ? atoi(argv[3]) : 12;
    int warm = argc > 4 ? atoi(argv[4]) : 0;
    uint64_t seed = argc > 5 ? strtoull(argv[5], 0, 10) : 1;
    verbose = argc > 6 ? atoi(argv[6]) : 0;
    mode_all = argc > 7 ? atoi(argv[7]) : 0;
    frac_done = argc > 8 ? atof(argv[8]) : 1.0;
    c = gen(kind, n, seed);
    p = calloc(n, 8); u = calloc(n, 8); r2c = malloc(4 * n); owner = malloc(4 * n); list = malloc(4 * n); pick = malloc(4 * n);
    bidv = malloc(8 * n); bidr = malloc(4 * n);
    int64_t opt = n <= 4096 ? lap_serial() : -1;
    for (int i = 0; i < n; i++) r2c[i] = owner[i] = -1;
    int U = n; for (int i = 0; i < n; i++) list[i] = i;
    if (warm) {
        int32_t cmin = INT32_MAX, cmax = INT32_MIN;
        for (int64_t k = 0; k < (int64_t)n * n; k++) { if (c[k] < cmin) cmin = c[k]; if (c[k] > cmax) cmax = c[k]; }
        int64_t eps = (int64_t)(cmax - cmin) / 4; if (eps < 1) eps = 1;
        for (;;) {
            for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
            int r = 0; int cut = getenv("CUT") ? atoi(getenv("CUT")) : 64; int rmax = getenv("RMAX") ? atoi(getenv("RMAX")) : 256; while (U > (cut ? n / cut : 0) && r < rmax) { U = jacobi_round(U, eps); r++; }
            printf("  eps=%ld: %d rounds, U=%d\n", (long)eps, r, U);
            if (eps == 1) break;
            eps /= warm; if (eps < 1) eps = 1;
        }
        for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
    }
    if (warm < 0) {
        untight = calloc(n, 4); allow_untight = 1; pre_rounds = arr_rounds;
        int32_t cmax = 0; for (int64_t k = 0; k < (int64_t)n * n; k++) if (c[k] > cmax) cmax = c[k];
        int top = 0; while ((cmax >> top) > 1) top++;
        int step = -warm;
        for (shift = top; ; shift -= step) {
            if (shift < 0) shift = 0;
            long ph0 = st_phases, lv0 = st_levels, sc0 = st_scans, ro0 = st_rounds;
            forest_phases(0);
            printf(" scale shift=%d: phases=%ld levels=%ld scans/n=%.2f rounds=%ld\n", shift, st_phases - ph0, st_levels - lv0, (double)(st_scans - sc0) / n, st_rounds - ro0);
            if (shift == 0) break;
            int sh = shift - step < 0 ? shift : step;
            for (int j = 0; j < n; j++) p[j] <<= sh;
        }
        int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
        printf("%s n=%d bit-scaling step %d: total=%ld opt=%ld %s | phases=%ld levels=%ld scans/n=%.2f rounds=%ld\n", kind, n, step, (long)tot, (long)opt, (opt < 0 || tot == opt) ? "OK" : "MISMATCH", st_phases, st_levels, (double)st_scans / n, st_rounds);
        return 0;
    }
    untight = calloc(n, 4);
    long wr = st_rounds;
    int r = 0;
    while (U > 0 && r < arr_rounds) { U = jacobi_round(U, 0); r++; }
    printf("  warm rounds=%ld ARR: %d rounds, U=%d\n", wr, r, U);
    long s0 = st_scans;
    if (mode_all <= -1000) core_solve(-mode_all - 1000); else if (mode_all == -1) forest_incremental(0); else if (mode_all <= -2) { int sf = -mode_all; int save = mode_all; mode_all = 0; forest_phases_until(sf); forest_incremental(0); mode_all = save; } else if (mode_all >= 2) forest_phases_win(mode_all); else forest_phases(0);
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    printf("%s n=%d: total=%ld opt=%ld %s | phases=%ld levels=%ld forest_scans/n=%.2f augs=%ld\n", kind, n, (long)tot, (long)opt,
           (opt < 0 || tot == opt) ? "OK" : "MISMATCH", st_phases, st_levels, (double)(st_scans - s0) / n, st_augs);
    printf("   incremental: rowjoins/n=%.2f repairs/n=%.2f\n", (double)st_rowjoins / n, (double)st_repairs / n);
    { int64_t dual = 0; for (int i = 0; i < n; i++) dual += u[i]; for (int j = 0; j < n; j++) dual -= p[j]; printf("   dual=%ld\n", (long)dual); }
    return 0;
}
