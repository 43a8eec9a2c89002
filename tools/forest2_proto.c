/* Exploration model #4 (dev tool, not product): WINDOWED incremental shortest-path forest.
 * eps=0 Jacobi rounds (optionally after eps>0 warm phases), then ONE continuous multi-source
 * label-correcting search: all free rows are roots, a level closes every open owned column whose label
 * lies within a window W above the smallest open label (labels may be tentative; a column whose label
 * drops after it was scanned is re-opened), a tree whose free column carries the smallest label of all
 * is augmented and released at once, columns that lose their best row are repaired by column scans.
 * Counts what sets GPU time: levels (grid-wide steps), row scans, column scans, end events.
 * build: gcc -O3 -fopenmp -o /tmp/forest2 tools/forest2_proto.c
 * usage: forest2 kind n arr_rounds warm seed W [verbose]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    if (!strcmp(kind, "g1") || !strcmp(kind, "wide") || !strcmp(kind, "w40k")) {
        uint64_t lo = !strcmp(kind, "g1") ? 10 : 0;
        uint64_t span = !strcmp(kind, "g1") ? 31 : (!strcmp(kind, "w40k") ? 40001 : 1000001);
        for (int64_t k = 0; k < (int64_t)n * n; k++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + k);
            c[k] = (int32_t)(lo + (((h >> 32) * span) >> 32));
        }
    } else {
        int S = 10 * n;
        int32_t *a = malloc(4 * n), *b = malloc(4 * n);
        for (int i = 0; i < n; i++) {
            a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)S);
            b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)S);
        }
        int two = !strcmp(kind, "g2d");
        int W = 4000;
        if (two) for (int i = 0; i < n; i++) { a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)(W * W)); b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)(W * W)); }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int32_t v = abs(a[i] - b[j]);
                if (two) v = abs(a[i] % W - b[j] % W) + abs(a[i] / W - b[j] / W);
                c[(int64_t)i * n + j] = v;
            }
        free(a); free(b);
    }
    return c;
}

static int n, shift;
static const int32_t *c;
#define CS(x) ((int64_t)((x) >> shift))
static int64_t *p, *u;
static int32_t *r2c, *owner;
static long st_rounds, st_scans;
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
        pick[t] = j1;
        bidv[i] = p[j1] + (w2 - w1 + eps);
    }
    for (int t = 0; t < U; t++) {
        int i = list[t], j = pick[t];
        if (bidr[j] < 0 || bidv[i] > bidv[bidr[j]] || (bidv[i] == bidv[bidr[j]] && i > bidr[j])) bidr[j] = i;
    }
    for (int t = 0; t < U; t++) {
        int i = list[t], j = pick[t];
        if (bidr[j] == i) { int o = owner[j]; if (o >= 0) r2c[o] = -1; owner[j] = i; r2c[i] = j; p[j] = bidv[i]; }
    }
    int U2 = 0;
    for (int i = 0; i < n; i++) if (r2c[i] < 0) list[U2++] = i;
    return U2;
}

#define INF ((int64_t)1 << 60)
static long st_levels, st_rowscans, st_repairs, st_ends, st_augs, st_reopen, st_endsteps;
static int verbose;

static void forest_windowed(int64_t W0, int adaptive)
{
    int64_t *lab = malloc(8 * n), *arow = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *inFr = calloc(n, 4), *inFc = calloc(n, 4), *dirty = calloc(n, 4);
    int32_t *pend = malloc(4 * n), *rootr = malloc(4 * n), *need = calloc(n, 4), *rel = calloc(n, 4);
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; }
        u[i] = m;
    }
    int nfree = 0, np = 0;
    for (int i = 0; i < n; i++) if (r2c[i] < 0) { nfree++; inFr[i] = 1; arow[i] = 0; pend[np++] = i; }
    for (int j = 0; j < n; j++) { lab[j] = INF; pred[j] = -1; }
    int64_t W = W0;
    while (nfree > 0) {
        /* 1. relax the pending rows against ALL columns (in-forest ones may be re-opened) */
        if (np) {
            st_rowscans += np;
#pragma omp parallel for schedule(static)
            for (int j = 0; j < n; j++) {
                int64_t s = lab[j]; int pr = pred[j];
                for (int t = 0; t < np; t++) { int i = pend[t]; int64_t h = arow[i] + CS(c[(int64_t)i * n + j]) + p[j] - u[i]; if (h < s) { s = h; pr = i; } }
                if (s < lab[j]) { lab[j] = s; pred[j] = pr; if (inFc[j]) dirty[j] = 1; }
            }
        }
        /* 2. next window */
        int64_t mfree = INF, dlo = INF;
        for (int j = 0; j < n; j++) {
            if (owner[j] < 0) { if (lab[j] < mfree) mfree = lab[j]; }
            else if ((inFc[j] && dirty[j]) || (!inFc[j] && lab[j] < INF)) { if (lab[j] < dlo) dlo = lab[j]; }
        }
        np = 0;
        if (dlo < mfree) {
            int64_t thr = dlo + W; if (thr > mfree && !getenv("NOCAP")) thr = mfree;
            long cand = 0;
            for (int j = 0; j < n; j++) {
                if (owner[j] < 0) continue;
                int open = (inFc[j] && dirty[j]) || (!inFc[j] && lab[j] < INF);
                if (!open) continue;
                if (lab[j] < mfree) cand++;
                if (lab[j] >= thr) continue;
                if (inFc[j]) st_reopen++;
                inFc[j] = 1; dirty[j] = 0; int i = owner[j]; inFr[i] = 1; arow[i] = lab[j]; pend[np++] = i;
            }
            st_levels++;
            if (adaptive) { if (np < 32 && cand > np) W *= 2; else if (np > 256 && W > 1) W /= 2; }
            if (verbose > 1) printf("   level %ld: dlo=%ld mfree=%ld W=%ld joined %d (cand %ld)\n", st_levels, (long)dlo, (long)mfree, (long)W, np, cand);
            continue;
        }
        if (mfree >= INF) { fprintf(stderr, "no path\n"); exit(3); }
        /* 3. END at mfree: every label below it is exact */
        st_endsteps++;
        int64_t D = mfree;
        /* roots by pred chains */
        for (int i = 0; i < n; i++) rootr[i] = -1;
        for (int i = 0; i < n; i++) if (inFr[i] && rootr[i] < 0) {
            int k = i, depth = 0; while (r2c[k] >= 0 && rootr[k] < 0) { k = pred[r2c[k]]; if (++depth > n) { fprintf(stderr, "cycle\n"); exit(5); } }
            int r = rootr[k] >= 0 ? rootr[k] : k;
            k = i; while (rootr[k] < 0) { rootr[k] = r; if (r2c[k] < 0) break; k = pred[r2c[k]]; }
        }
        int nrel = 0;
        for (int j = 0; j < n; j++) if (owner[j] < 0 && lab[j] == D) {
            int r = rootr[pred[j]];
            if (rel[r]) continue;
            rel[r] = 1; nrel++;
            int jj = j; for (;;) { int ii = pred[jj]; owner[jj] = ii; int t = r2c[ii]; r2c[ii] = jj; jj = t; if (t < 0) break; }
            st_augs++; nfree--;
        }
        st_ends += nrel;
        /* release: materialise duals (raise = D - label when positive) */
        for (int i = 0; i < n; i++) if (inFr[i] && rel[rootr[i]]) {
            int64_t a = arow[i]; if (a < D) u[i] += D - a;
            inFr[i] = 0;
        }
        /* columns of released trees: those whose owner-before-augment row was released; after the augmentation
         * owner[] changed along paths, so decide by pred chain: a forest column belongs to the tree of pred */
        for (int j = 0; j < n; j++) if (inFc[j] && rel[rootr[pred[j]]]) {
            if (lab[j] < D) p[j] += D - lab[j];
            inFc[j] = 0; dirty[j] = 0; need[j] = 1;
        }
        for (int j = 0; j < n; j++) if (!inFc[j] && !need[j] && pred[j] >= 0 && rel[rootr[pred[j]]]) need[j] = 1;
        long rep = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+:rep)
        for (int j = 0; j < n; j++) if (need[j]) {
            rep++;
            int64_t s = INF; int pr = -1;
            for (int i = 0; i < n; i++) if (inFr[i]) { int64_t h = arow[i] + CS(c[(int64_t)i * n + j]) + p[j] - u[i]; if (h < s) { s = h; pr = i; } }
            lab[j] = s; pred[j] = pr; need[j] = 0;
        }
        st_repairs += rep;
        for (int i = 0; i < n; i++) rel[i] = 0;
        if (verbose) printf("   END D=%ld: released %d trees, repaired %ld cols, free=%d levels=%ld\n", (long)D, nrel, rep, nfree, st_levels);
    }
    /* what is left in the forest has no free root any more: cannot happen (every tree has a free root) */
    for (int i = 0; i < n; i++) if (inFr[i]) { fprintf(stderr, "forest not empty at the end\n"); exit(7); }
}

/* PHASE version: every phase starts from scratch (all free rows are roots), windowed label-correcting levels until the
 * smallest free-column label is below every open label, dual update, one augmentation per end column whose path is
 * vertex-disjoint from the paths taken before it in this phase.  No repairs, no transposed matrix. */
static long st_phases;
static void forest_phases_windowed(int64_t W0, int adaptive, int multi)
{
    int64_t *lab = malloc(8 * n), *arow = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *inFr = calloc(n, 4), *inFc = calloc(n, 4), *dirty = calloc(n, 4);
    int32_t *pend = malloc(4 * n), *mark = calloc(n, 4);
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; }
        u[i] = m;
    }
    int64_t W = W0;
    for (;;) {
        int nfree = 0, np = 0;
        for (int i = 0; i < n; i++) { inFr[i] = 0; if (r2c[i] < 0) { nfree++; inFr[i] = 1; arow[i] = 0; pend[np++] = i; } }
        if (!nfree) break;
        st_phases++;
        for (int j = 0; j < n; j++) { lab[j] = INF; pred[j] = -1; inFc[j] = 0; dirty[j] = 0; }
        int64_t D;
        for (;;) {
            if (np) {
                st_rowscans += np;
#pragma omp parallel for schedule(static)
                for (int j = 0; j < n; j++) {
                    int64_t s = lab[j]; int pr = pred[j];
                    for (int t = 0; t < np; t++) { int i = pend[t]; int64_t h = arow[i] + CS(c[(int64_t)i * n + j]) + p[j] - u[i]; if (h < s) { s = h; pr = i; } }
                    if (s < lab[j]) { lab[j] = s; pred[j] = pr; if (inFc[j]) dirty[j] = 1; }
                }
            }
            int64_t mfree = INF, dlo = INF;
            for (int j = 0; j < n; j++) {
                if (owner[j] < 0) { if (lab[j] < mfree) mfree = lab[j]; }
                else if ((inFc[j] && dirty[j]) || (!inFc[j] && lab[j] < INF)) { if (lab[j] < dlo) dlo = lab[j]; }
            }
            np = 0;
            if (dlo > mfree || dlo >= INF) { D = mfree; break; }
            int64_t thr = dlo + W; if (thr > mfree + 1) thr = mfree + 1;
            long cand = 0;
            for (int j = 0; j < n; j++) {
                if (owner[j] < 0) continue;
                int open = (inFc[j] && dirty[j]) || (!inFc[j] && lab[j] < INF);
                if (!open) continue;
                if (lab[j] <= mfree) cand++;
                if (lab[j] >= thr) continue;
                if (inFc[j]) st_reopen++;
                inFc[j] = 1; dirty[j] = 0; int i = owner[j]; inFr[i] = 1; arow[i] = lab[j]; pend[np++] = i;
            }
            st_levels++;
            if (adaptive) { if (np < 32 && cand > np) W *= 2; else if (np > 256 && W > 1) W /= 2; }
        }
        st_endsteps++;
        for (int j = 0; j < n; j++) if (lab[j] < D) p[j] += D - lab[j];
        for (int i = 0; i < n; i++) if (inFr[i] && arow[i] < D) u[i] += D - arow[i];
        int augs = 0;
        for (int j = 0; j < n; j++) if (owner[j] < 0 && lab[j] == D) {
            /* check the path is disjoint from earlier ones of this phase, and tight all the way */
            int ok = 1, jj = j, len = 0;
            for (;;) { int ii = pred[jj]; if (mark[ii] == st_phases) { ok = 0; break; } int t = r2c[ii]; if (t < 0) break; jj = t; if (++len > n) { fprintf(stderr, "cycle\n"); exit(5); } }
            if (!ok) continue;
            jj = j; for (;;) { int ii = pred[jj]; mark[ii] = (int)st_phases; owner[jj] = ii; int t = r2c[ii]; r2c[ii] = jj; jj = t; if (t < 0) break; }
            augs++; st_augs++;
            if (!multi) break;
        }
        if (verbose) printf("   phase %ld: free=%d D=%ld augs=%d levels=%ld\n", st_phases, nfree, (long)D, augs, st_levels);
    }
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "g2";
    n = argc > 2 ? atoi(argv[2]) : 1000;
    int arr_rounds = argc > 3 ? atoi(argv[3]) : 12;
    int warm = argc > 4 ? atoi(argv[4]) : 0;
    uint64_t seed = argc > 5 ? strtoull(argv[5], 0, 10) : 1;
    int64_t W = argc > 6 ? atoll(argv[6]) : 1;
    verbose = argc > 7 ? atoi(argv[7]) : 0;
    int adaptive = W < 0; if (adaptive) W = -W;
    c = gen(kind, n, seed);
    p = calloc(n, 8); u = calloc(n, 8); r2c = malloc(4 * n); owner = malloc(4 * n); list = malloc(4 * n); pick = malloc(4 * n);
    bidv = malloc(8 * n); bidr = malloc(4 * n);
    for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; }
    int U = n;
    if (warm) {
        int32_t cmin = INT32_MAX, cmax = INT32_MIN;
        for (int64_t k = 0; k < (int64_t)n * n; k++) { if (c[k] < cmin) cmin = c[k]; if (c[k] > cmax) cmax = c[k]; }
        int64_t eps = (int64_t)(cmax - cmin) / 4; if (eps < 1) eps = 1;
        for (;;) {
            for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
            int r = 0; while (U > n / 64 && r < 256) { U = jacobi_round(U, eps); r++; }
            if (eps == 1) break;
            eps /= warm; if (eps < 1) eps = 1;
        }
        for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
    }
    if (getenv("BITS")) {
        int step = atoi(getenv("BITS"));
        int32_t cmax = 0; for (int64_t k = 0; k < (int64_t)n * n; k++) if (c[k] > cmax) cmax = c[k];
        int top = 0; while ((cmax >> top) > 1) top++;
        for (shift = top;; shift -= step) {
            if (shift < 0) shift = 0;
            long l0 = st_levels, e0 = st_endsteps, r0 = st_rowscans, q0 = st_repairs, a0 = st_augs, s0 = st_scans;
            /* keep matches that are still tight under the doubled prices */
            int rel = 0;
            for (int i = 0; i < n; i++) { int64_t m = INT64_MAX; const int32_t *ci = c + (int64_t)i * n; for (int j = 0; j < n; j++) { int64_t w = CS(ci[j]) + p[j]; if (w < m) m = w; }
                int j = r2c[i]; if (j >= 0 && CS(ci[j]) + p[j] != m) { owner[j] = -1; r2c[i] = -1; rel++; } }
            U = 0; for (int i = 0; i < n; i++) if (r2c[i] < 0) list[U++] = i;
            int U0 = U; int r = 0; while (U > 0 && r < arr_rounds) { U = jacobi_round(U, 0); r++; }
            forest_windowed(W, adaptive);
            printf(" shift=%d: untight %d free %d -> after ARR %d | levels=%ld ends=%ld rowscans/n=%.1f repairs/n=%.1f arr scans/n=%.1f\n", shift, rel, U0, U, st_levels - l0, st_endsteps - e0,
                   (double)(st_rowscans - r0) / n, (double)(st_repairs - q0) / n, (double)(st_scans - s0) / n);
            if (shift == 0) break;
            int sh = shift - step < 0 ? shift : step;
            for (int j = 0; j < n; j++) p[j] <<= sh;
        }
        U = 0;
    }
    long wr = st_rounds, ws = st_scans;
    int r = 0;
    while (U > 0 && r < arr_rounds) { U = jacobi_round(U, 0); r++; }
    printf("  warm rounds=%ld (scans/n %.1f), ARR %d rounds, free=%d\n", wr, (double)ws / n, r, U);
    if (getenv("PHASES")) forest_phases_windowed(W, adaptive, atoi(getenv("PHASES"))); else forest_windowed(W, adaptive);
    shift = 0; int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    int64_t dual = 0; for (int i = 0; i < n; i++) dual += u[i]; for (int j = 0; j < n; j++) dual -= p[j];
    /* feasibility of the duals */
    long bad = 0;
#pragma omp parallel for reduction(+:bad)
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (CS(c[(int64_t)i * n + j]) + p[j] - u[i] < 0) bad++;
    printf("%s n=%d W=%ld%s: total=%ld dual=%ld %s infeasible=%ld | levels=%ld endsteps=%ld rowscans/n=%.2f reopen/n=%.2f repairs/n=%.2f augs=%ld\n",
           kind, n, (long)W, adaptive ? "(adaptive)" : "", (long)tot, (long)dual, tot == dual && !bad ? "OK" : "MISMATCH", bad,
           st_levels, st_endsteps, (double)st_rowscans / n, (double)st_reopen / n, (double)st_repairs / n, st_augs);
    return 0;
}
