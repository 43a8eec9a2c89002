/* Exploration model #2 (dev tool): parallel-JV = Jacobi eps=0 auction rounds ("ARR") + shortest
 * augmenting path finish; optional eps>0 auction phases in front.  Counts rounds / scans / SAP steps.
 * build: gcc -O3 -fopenmp -o /tmp/jv_proto tools/jv_proto.c
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
    } else if (!strcmp(kind, "g2") || !strcmp(kind, "g3")) {
        int g3 = !strcmp(kind, "g3");
        int S = g3 ? 50 : 10 * n;
        int nd = g3 ? (int)(n * 0.363) : n;
        int32_t *a = malloc(4 * n), *b = malloc(4 * n);
        for (int i = 0; i < n; i++) {
            a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)S);
            b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)S);
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int32_t v = abs(a[i] - b[j]);
                if (g3) v = (j < nd && v < 10) ? v : 250000;
                c[(int64_t)i * n + j] = v;
            }
        free(a); free(b);
    } else { fprintf(stderr, "kind?\n"); exit(1); }
    return c;
}

static int n;
static const int32_t *c;
static int64_t K, *p;
static int32_t *r2c, *owner;
static long st_rounds, st_scans, st_steps, st_augs;

/* one Jacobi auction round with increment eps (eps=0: ARR rules) ; returns #unassigned after */
static int32_t *list, *pick; static int64_t *bidv; static int32_t *bidr;
static int jacobi_round(int U, int64_t eps)
{
    st_rounds++; st_scans += U;
    for (int j = 0; j < n; j++) bidr[j] = -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < U; t++) {
        int i = list[t];
        const int32_t *ci = c + (int64_t)i * n;
        int off = (int)(splitmix64(i * 0x9E37ull + 12345) % (uint64_t)n);
        /* key = 2*w + owned  => among equal w prefer a free column */
        int64_t k1 = INT64_MAX, k2 = INT64_MAX; int j1 = -1;
        for (int s = 0; s < n; s++) {
            int j = s + off; if (j >= n) j -= n;
            int64_t k = 2 * (K * ci[j] + p[j]) + (owner[j] >= 0);
            if (k < k1) { k2 = k1; k1 = k; j1 = j; } else if (k < k2) k2 = k;
        }
        int64_t w1 = k1 >> 1, w2 = (n == 1) ? w1 : (k2 >> 1);
        int64_t inc = w2 - w1 + eps;
        pick[t] = j1;
        if (eps == 0 && inc == 0 && owner[j1] >= 0) pick[t] = -1; /* tie on owned cols: leave to SAP */
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

/* SAP finish: exact; requires matched pairs tight w.r.t. u_i = min_k w_ik */
static void sap_finish(void)
{
    int64_t *u = malloc(8 * n), *dd = malloc(8 * n);
    int32_t *pred = malloc(4 * n), *cl = malloc(4 * n);
    for (int i = 0; i < n; i++) {
        const int32_t *ci = c + (int64_t)i * n; int64_t m = INT64_MAX;
        for (int j = 0; j < n; j++) { int64_t w = K * ci[j] + p[j]; if (w < m) m = w; }
        u[i] = m;
        if (r2c[i] >= 0 && K * ci[r2c[i]] + p[r2c[i]] != m) { fprintf(stderr, "non-tight match row %d\n", i); exit(2); }
    }
    st_scans += n;
    for (int f = 0; f < n; f++) {
        if (r2c[f] >= 0) continue;
        st_augs++;
        int low = 0, up = 0, endcol = -1; int64_t mind = 0;
        for (int j = 0; j < n; j++) { cl[j] = j; dd[j] = K * c[(int64_t)f * n + j] + p[j] - u[f]; pred[j] = f; }
        st_steps++;
        while (endcol < 0) {
            if (low == up) {
                mind = dd[cl[up]]; up++;
                for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = dd[j];
                    if (h <= mind) { if (h < mind) { up = low; mind = h; } cl[k] = cl[up]; cl[up] = j; up++; } }
                for (int k = low; k < up; k++) if (owner[cl[k]] < 0) { endcol = cl[k]; break; }
            }
            if (endcol >= 0) break;
            int j1 = cl[low]; low++; int i = owner[j1]; const int32_t *ci = c + (int64_t)i * n;
            st_steps++;
            for (int k = up; k < n; k++) { int j = cl[k]; int64_t h = K * ci[j] + p[j] - u[i] + mind;
                if (h < dd[j]) { dd[j] = h; pred[j] = i;
                    if (h == mind) { if (owner[j] < 0) { endcol = j; break; } cl[k] = cl[up]; cl[up] = j; up++; } } }
        }
        for (int k = 0; k < low; k++) { int j = cl[k]; int i = owner[j]; int64_t d = mind - dd[j]; p[j] += d; u[i] += d; }
        u[f] += mind;
        int j = endcol; for (;;) { int i = pred[j]; owner[j] = i; int t = r2c[i]; r2c[i] = j; j = t; if (i == f) break; }
    }
    free(u); free(dd); free(pred); free(cl);
}

/* plain exact for checking */
static int64_t lap(void)
{
    int64_t *sp = p; int32_t *sr = r2c, *so = owner; int64_t sK = K;
    p = calloc(n, 8); r2c = malloc(4 * n); owner = malloc(4 * n); K = 1;
    for (int i = 0; i < n; i++) r2c[i] = owner[i] = -1;
    long a = st_scans, b = st_steps, d = st_augs;
    sap_finish();
    st_scans = a; st_steps = b; st_augs = d;
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    free(p); free(r2c); free(owner); p = sp; r2c = sr; owner = so; K = sK;
    return tot;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "g1";
    n = argc > 2 ? atoi(argv[2]) : 1000;
    int arr_rounds = argc > 3 ? atoi(argv[3]) : 16;   /* max eps=0 rounds */
    int eps_mode = argc > 4 ? atoi(argv[4]) : 0;      /* 0 none; 1: eps=1 attempt with budget; 2: eps-scaling w/ budget per phase */
    int budget = argc > 5 ? atoi(argv[5]) : 64;
    int colred = argc > 6 ? atoi(argv[6]) : 0;        /* 1: init prices by column reduction */
    uint64_t seed = argc > 7 ? strtoull(argv[7], 0, 10) : 1;
    c = gen(kind, n, seed);
    K = (eps_mode ? n + 1 : 1);
    p = calloc(n, 8); r2c = malloc(4 * n); owner = malloc(4 * n); list = malloc(4 * n); pick = malloc(4 * n);
    bidv = malloc(8 * n); bidr = malloc(4 * n);
    int64_t opt = lap();
    for (int i = 0; i < n; i++) r2c[i] = owner[i] = -1;
    if (colred) {
        for (int j = 0; j < n; j++) { int32_t m = INT32_MAX; for (int i = 0; i < n; i++) if (c[(int64_t)i * n + j] < m) m = c[(int64_t)i * n + j]; p[j] = -K * m; }
        st_scans += n;
    }
    int U = n; for (int i = 0; i < n; i++) list[i] = i;
    if (eps_mode == 1) {
        int r = 0; while (U > 0 && r < budget) { U = jacobi_round(U, 1); r++; }
        printf("  eps=1 attempt: %d rounds, U=%d\n", r, U);
    } else if (eps_mode == 2) {
        int32_t cmin = INT32_MAX, cmax = INT32_MIN;
        for (int64_t k = 0; k < (int64_t)n * n; k++) { if (c[k] < cmin) cmin = c[k]; if (c[k] > cmax) cmax = c[k]; }
        int64_t eps = K * (int64_t)(cmax - cmin) / 4; if (eps < 1) eps = 1;
        for (;;) {
            for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
            int r = 0; while (U > 0 && r < budget) { U = jacobi_round(U, eps); r++; }
            printf("  eps=%ld: %d rounds, U=%d\n", (long)eps, r, U);
            if (eps == 1) break;
            eps /= 8; if (eps < 1) eps = 1;
        }
    }
    if (!(eps_mode && U == 0)) {
        /* re-match tightly under current prices with eps=0 rounds, then SAP */
        for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
        int r = 0, prevU = n + 1;
        while (U > 0 && r < arr_rounds) { prevU = U; U = jacobi_round(U, 0); r++; if (U == prevU && 0) break; }
        printf("  ARR: %d rounds, U=%d\n", r, U);
        long s0 = st_steps;
        sap_finish();
        printf("  SAP: augs=%ld steps=%ld\n", st_augs, st_steps - s0);
    }
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    printf("%s n=%d arr=%d epsmode=%d budget=%d colred=%d: total=%ld opt=%ld %s rounds=%ld scans/n=%.2f sap_augs=%ld sap_steps=%ld\n", kind, n, arr_rounds, eps_mode,
           budget, colred, (long)tot, (long)opt, tot == opt ? "OK" : "MISMATCH", st_rounds, (double)st_scans / n, st_augs, st_steps);
    return 0;
}
