/* Exploration model of the Jacobi eps-scaling auction (dev tool, not shipped, not the oracle).
 * Counts rounds and row scans for candidate schedules so the HIP design can be sized.
 * build: gcc -O3 -fopenmp -o /tmp/auction_proto tools/auction_proto.c
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

typedef struct {
    int n;
    int32_t *c;
} inst_t;

static int cmp_i32(const void *a, const void *b) { return (*(int32_t *)a > *(int32_t *)b) - (*(int32_t *)a < *(int32_t *)b); }

static void gen(inst_t *I, const char *kind, int n, uint64_t seed)
{
    I->n = n;
    I->c = malloc(sizeof(int32_t) * (size_t)n * n);
    if (!strcmp(kind, "g1")) {
        for (int64_t k = 0; k < (int64_t)n * n; k++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + k);
            I->c[k] = 10 + (int32_t)(((h >> 32) * 31ull) >> 32);
        }
    } else if (!strcmp(kind, "g4")) {
        for (int64_t k = 0; k < (int64_t)n * n; k++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + k);
            I->c[k] = 1 + (int32_t)(((h >> 32) * 39ull) >> 32);
        }
    } else if (!strcmp(kind, "g2")) { /* |a-b|, S = 10 n */
        int32_t *a = malloc(4 * n), *b = malloc(4 * n);
        for (int i = 0; i < n; i++) {
            a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)(10 * n));
            b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)(10 * n));
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) I->c[(int64_t)i * n + j] = abs(a[i] - b[j]);
        free(a); free(b);
    } else if (!strcmp(kind, "g3")) { /* simulator-like: S=50, thr 10, big 250000, n_d = 0.36 n */
        int nd = (int)(n * 0.363);
        int32_t *a = malloc(4 * n), *b = malloc(4 * n);
        for (int i = 0; i < n; i++) {
            a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % 50ull);
            b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % 50ull);
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int32_t v = 250000;
                if (j < nd && abs(a[i] - b[j]) < 10) v = abs(a[i] - b[j]);
                I->c[(int64_t)i * n + j] = v;
            }
        free(a); free(b);
    } else if (!strcmp(kind, "wide")) {
        for (int64_t k = 0; k < (int64_t)n * n; k++) {
            uint64_t h = splitmix64(seed * 0x100000001B3ull + k);
            I->c[k] = (int32_t)(((h >> 32) * 1000000ull) >> 32);
        }
    } else { fprintf(stderr, "kind?\n"); exit(1); }
}

/* exact solver for checking (same as oracle, compact) */
static int64_t lap(int n, const int32_t *c)
{
    int64_t *u = calloc(n, 8), *v = calloc(n, 8), *dd = malloc(8 * n);
    int32_t *r2c = malloc(4 * n), *c2r = malloc(4 * n), *pred = malloc(4 * n), *cl = malloc(4 * n);
    for (int i = 0; i < n; i++) r2c[i] = c2r[i] = -1;
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int32_t m = ci[0];
        for (int j = 1; j < n; j++) if (ci[j] < m) m = ci[j];
        u[i] = m;
        for (int j = 0; j < n; j++) if (ci[j] == m && c2r[j] < 0) { c2r[j] = i; r2c[i] = j; break; }
    }
    for (int f = 0; f < n; f++) {
        if (r2c[f] >= 0) continue;
        int low = 0, up = 0, endcol = -1; int64_t mind = 0;
        for (int j = 0; j < n; j++) { cl[j] = j; dd[j] = (int64_t)c[(int64_t)f * n + j] - u[f] - v[j]; pred[j] = f; }
        while (endcol < 0) {
            if (low == up) {
                mind = dd[cl[up]]; up++;
                for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = dd[j];
                    if (h <= mind) { if (h < mind) { up = low; mind = h; } cl[k] = cl[up]; cl[up] = j; up++; } }
                for (int k = low; k < up; k++) if (c2r[cl[k]] < 0) { endcol = cl[k]; break; }
            }
            if (endcol >= 0) break;
            int j1 = cl[low]; low++; int i = c2r[j1]; const int32_t *ci = c + (int64_t)i * n;
            for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = (int64_t)ci[j] - u[i] - v[j] + mind;
                if (h < dd[j]) { dd[j] = h; pred[j] = i;
                    if (h == mind) { if (c2r[j] < 0) { endcol = j; break; } cl[k] = cl[up]; cl[up] = j; up++; } } }
        }
        for (int k = 0; k < low; k++) { int j = cl[k]; int i = c2r[j]; int64_t d = mind - dd[j]; v[j] -= d; u[i] += d; }
        u[f] += mind;
        int j = endcol; for (;;) { int i = pred[j]; c2r[j] = i; int t = r2c[i]; r2c[i] = j; j = t; if (i == f) break; }
    }
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    free(u); free(v); free(dd); free(r2c); free(c2r); free(pred); free(cl);
    return tot;
}

typedef struct {
    int64_t rounds, scans, phases;
} stats_t;

/* dual bound in scaled units */
static int64_t dual_bound(int n, const int32_t *c, const int64_t *p, int64_t K)
{
    int64_t d = 0;
#pragma omp parallel for reduction(+ : d)
    for (int i = 0; i < n; i++) {
        int64_t m = INT64_MAX; const int32_t *ci = c + (int64_t)i * n;
        for (int j = 0; j < n; j++) { int64_t w = K * ci[j] + p[j]; if (w < m) m = w; }
        d += m;
    }
    for (int j = 0; j < n; j++) d -= p[j];
    return d;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "g1";
    int n = argc > 2 ? atoi(argv[2]) : 1000;
    int theta = argc > 3 ? atoi(argv[3]) : 8;
    int keep = argc > 4 ? atoi(argv[4]) : 0;      /* 0 reset at each phase, 1 keep eps-CS rows */
    double e0div = argc > 5 ? atof(argv[5]) : 2;  /* eps0 = K*range/e0div ; 0 => eps0=1 */
    int rot = argc > 6 ? atoi(argv[6]) : 1;
    int verbose = argc > 7 ? atoi(argv[7]) : 0;
    uint64_t seed = argc > 8 ? strtoull(argv[8], 0, 10) : 1;
    inst_t I; gen(&I, kind, n, seed);
    const int32_t *c = I.c;
    int32_t cmin = INT32_MAX, cmax = INT32_MIN;
    for (int64_t k = 0; k < (int64_t)n * n; k++) { if (c[k] < cmin) cmin = c[k]; if (c[k] > cmax) cmax = c[k]; }
    int64_t K = n + 1;
    int64_t range = (int64_t)(cmax - cmin);
    int64_t eps = e0div > 0 ? (int64_t)(K * range / e0div) : 1;
    if (eps < 1) eps = 1;
    int64_t *p = calloc(n, 8);
    int32_t *r2c = malloc(4 * n), *owner = malloc(4 * n), *list = malloc(4 * n), *nlist = malloc(4 * n);
    uint64_t *bidk = calloc(n, 8); int64_t *bidv = calloc(n, 8); int32_t *bidr = malloc(4 * n);
    for (int i = 0; i < n; i++) { r2c[i] = -1; owner[i] = -1; }
    stats_t S = {0, 0, 0};
    int64_t opt = (n <= 20000) ? lap(n, c) : -1;
    clock_t t0 = clock();
    int done = 0;
    while (!done) {
        S.phases++;
        int U = 0;
        if (!keep || S.phases == 1) {
            for (int i = 0; i < n; i++) { r2c[i] = -1; owner[i] = -1; list[U++] = i; }
        } else {
            /* keep rows whose current col is within eps of their best */
            S.scans += n;
            for (int i = 0; i < n; i++) {
                const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
                for (int j = 0; j < n; j++) { int64_t w = K * ci[j] + p[j]; if (w < m) m = w; }
                int j = r2c[i];
                if (j >= 0 && K * ci[j] + p[j] <= m + eps) continue;
                if (j >= 0) { owner[j] = -1; r2c[i] = -1; }
                list[U++] = i;
            }
        }
        int64_t ph_rounds = 0, ph_scans = 0;
        while (U > 0) {
            ph_rounds++; ph_scans += U;
            for (int j = 0; j < n; j++) bidr[j] = -1;
#pragma omp parallel for schedule(dynamic, 16)
            for (int t = 0; t < U; t++) {
                int i = list[t];
                const int32_t *ci = c + (int64_t)i * n;
                int off = rot ? (int)(splitmix64(i * 0x9E37ull + 12345) % (uint64_t)n) : 0;
                int64_t w1 = INT64_MAX, w2 = INT64_MAX; int j1 = -1;
                for (int s = 0; s < n; s++) {
                    int j = s + off; if (j >= n) j -= n;
                    int64_t w = K * ci[j] + p[j];
                    if (w < w1) { w2 = w1; w1 = w; j1 = j; } else if (w < w2) w2 = w;
                }
                int64_t b = (n == 1) ? p[j1] + eps : w2 - K * ci[j1] + eps;
                /* stash per-row bid; resolved serially below to stay deterministic */
                bidv[i] = b; nlist[t] = j1;
            }
            for (int t = 0; t < U; t++) {
                int i = list[t], j = nlist[t];
                if (bidr[j] < 0 || bidv[i] > bidv[bidr[j]] || (bidv[i] == bidv[bidr[j]] && i > bidr[j])) bidr[j] = i;
            }
            int U2 = 0;
            for (int t = 0; t < U; t++) {
                int i = list[t], j = nlist[t];
                if (bidr[j] == i) {
                    int o = owner[j];
                    if (o >= 0) { r2c[o] = -1; bidk[U2] = 0; nlist[U2] = 0; }
                    owner[j] = i; r2c[i] = j; p[j] = bidv[i];
                }
            }
            /* rebuild unassigned list in row order */
            U2 = 0;
            for (int i = 0; i < n; i++) if (r2c[i] < 0) list[U2++] = i;
            U = U2;
            if (verbose > 1) printf("   round %ld U=%d\n", (long)ph_rounds, U);
            if (ph_rounds > 2000000) { printf("ABORT rounds\n"); return 1; }
        }
        S.rounds += ph_rounds; S.scans += ph_scans;
        int64_t P = 0; for (int i = 0; i < n; i++) P += K * c[(int64_t)i * n + r2c[i]];
        int64_t D = dual_bound(n, c, p, K); S.scans += n;
        if (verbose) printf(" phase %ld eps=%ld rounds=%ld scans/n=%.2f P/K=%ld gap/K=%.4f\n", (long)S.phases, (long)eps, (long)ph_rounds,
                            (double)ph_scans / n, (long)(P / K), (double)(P - D) / (double)K);
        if (P - D < K) done = 1;
        else if (eps == 1) { printf("eps=1 but certificate open?!\n"); done = 1; }
        else { eps = eps / theta; if (eps < 1) eps = 1; }
        if (done) {
            double sec = (double)(clock() - t0) / CLOCKS_PER_SEC;
            printf("%s n=%d theta=%d keep=%d e0div=%g rot=%d: total=%ld opt=%ld %s phases=%ld rounds=%ld scans/n=%.2f  cpu %.2fs\n", kind, n, theta, keep,
                   e0div, rot, (long)(P / K), (long)opt, (opt < 0 || opt == P / K) ? "OK" : "MISMATCH", (long)S.phases, (long)S.rounds, (double)S.scans / n, sec);
        }
    }
    return 0;
}
