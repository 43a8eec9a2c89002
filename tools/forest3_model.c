/* CPU model of csrc/td_forest.hip (dev tool, not product): the windowed incremental shortest-path forest
 * exactly as the cooperative kernel runs it — column slices owned by "workgroups", one board (entries +
 * header per workgroup) per level, selection gated by the PREVIOUS board's global minima (one barrier per
 * level), predecessor columns + tree roots carried with the labels, END = a board without entries whose
 * smallest open label is not below the smallest free-column label.  Used to validate the protocol
 * (exactness: dual == total, all reduced costs >= 0) before it runs on the GPU, and to count levels,
 * row scans and END events.
 * build: gcc -O3 -fopenmp -o /tmp/forest3 tools/forest3_model.c
 * usage: forest3 kind n arr_rounds warm seed W0 [CW] [CAP] [verbose]
 * Round-4 experiments behind environment variables (none of them is in the kernel; DESIGN.md 2.12 has the outcomes):
 *   SCALE=K        costs * K before the warm start, i.e. an eps ladder that ends at 1/K of a cost unit: the rows the
 *                  eps = 0 rounds leave for the forest do NOT get fewer (g2d n = 4096: 251 / 245 / 261 / 259 for K = 1 / 4 / 16 / 64)
 *   PURE=1 SCALE=n+1   eps-scaling run to the end (eps < 1/n, no exact finisher): the last phase's tail did not end
 *                  within 20 minutes at n = 4096
 *   QDIV=d RELMAX=m    deferred END: at a would-be END the forest grows on to the (free rows / d)-th smallest free label
 *                  the headers know and releases up to m trees at once (prices of a released tree's free columns are raised
 *                  too).  g2d n = 2048: ENDs 38 -> 16, levels 495 -> 438, exact; at n = 8192 it runs into the level guard
 *                  (unfinished: an END below the deferred label is needed when trees are left over).  Levels are dependency
 *                  depth, not END count — not pursued.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t LT;
#define INF ((LT)1 << 60)

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
    if (!strcmp(kind, "g1") || !strcmp(kind, "wide") || !strcmp(kind, "w40k") || !strcmp(kind, "w100")) {
        uint64_t lo = !strcmp(kind, "g1") ? 10 : 0;
        uint64_t span = !strcmp(kind, "g1") ? 31 : (!strcmp(kind, "w40k") ? 40001 : (!strcmp(kind, "w100") ? 101 : 1000001));
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
        int Wd = 4000;
        if (two) for (int i = 0; i < n; i++) { a[i] = (int32_t)(splitmix64(seed + 7919ull * i) % (uint64_t)(Wd * Wd)); b[i] = (int32_t)(splitmix64(seed + 104729ull * i + 13) % (uint64_t)(Wd * Wd)); }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                int32_t v = abs(a[i] - b[j]);
                if (two) v = abs(a[i] % Wd - b[j] % Wd) + abs(a[i] / Wd - b[j] / Wd);
                c[(int64_t)i * n + j] = v;
            }
        free(a); free(b);
    }
    return c;
}

static int n;
static const int32_t *c;
static int64_t *p;
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
            int64_t k = 2 * (ci[j] + p[j]) + (owner[j] >= 0);
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

/* ---------------------------------------------------------------- the model ---------------- */
typedef struct { int row, root, col; LT base; } Ent;
typedef struct { LT minopen, minfree; int nend; int ecol[2], eroot[2]; int cnt; } Hdr;

static int CW = 64, CAP = 4, G;
/* per column (owned by workgroup j / CW; "LDS / registers" of that workgroup) */
static LT *lab, *price, *cown;
static int32_t *pc, *root, *own, *inF, *dirty, *need, *urgent;
/* global arrays (other workgroups read them) */
static LT *g_base; static int32_t *g_root, *g_col, *g_pc;
static long st_levels, st_rowscans, st_ends, st_endsteps, st_repair_rows, st_reopen, st_emptylv;
static int verbose, LO = 32, HI = 256, QDIV = 0, RELMAX = 8; static long st_exts; static long dbg_urg, dbg_above, dbg_below, dbg_rel; static LT dbg_mfree;

static void relax(int w, const Ent *E, int m)
{
    const int j0 = w * CW, j1 = (j0 + CW < n) ? j0 + CW : n;
    for (int j = j0; j < j1; j++) {
        LT best = INF; int bk = -1;
        for (int k = 0; k < m; k++) {
            const LT h = E[k].base + c[(int64_t)E[k].row * n + j];
            const int same = (E[k].col == pc[j]) && (E[k].col >= 0 || E[k].root == root[j]);
            if (h < best || (h == best && same)) { best = h; bk = k; }
        }
        if (bk < 0) continue;
        const LT tl = (lab[j] >= INF) ? INF : lab[j] - price[j];
        if (best < tl) {
            if (inF[j]) { dirty[j] = 1; if (root[j] != E[bk].root) urgent[j] = 1; }
            lab[j] = best + price[j]; pc[j] = E[bk].col; root[j] = E[bk].root; g_pc[j] = pc[j];
        } else if (best == tl && E[bk].col == pc[j] && (E[bk].col >= 0 || E[bk].root == root[j]) && root[j] != E[bk].root) {
            root[j] = E[bk].root;        /* the predecessor moved to another tree at the same label */
            if (inF[j]) { dirty[j] = 1; urgent[j] = 1; }
        }
    }
}

/* relax the slice against every forest row (repairs and the initial pass) */
static void relax_all(int w)
{
    Ent E[256];
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        m = 0;
        for (int i = i0; i < i0 + 256 && i < n; i++)
            if (g_base[i] < INF) { E[m].row = i; E[m].base = g_base[i]; E[m].root = g_root[i]; E[m].col = g_col[i]; m++; }
        if (m) { relax(w, E, m);
#pragma omp atomic
            st_repair_rows += m; }
    }
}

static void select_wg(int w, int gate, LT thr, Ent *ent, Hdr *h)
{
    const int j0 = w * CW, j1 = (j0 + CW < n) ? j0 + CW : n;
    h->cnt = 0; h->minopen = INF; h->minfree = INF; h->nend = 0;
    /* the CAP smallest open labels below thr (urgent ones first), like CAP rounds of a workgroup argmin */
    for (int round = 0; round < CAP; round++) {
        int bj = -1; LT bl = INF; int bu = 0;
        for (int j = j0; j < j1; j++) {
            if (own[j] < 0) continue;
            const int open = (!inF[j] && lab[j] < INF) || (inF[j] && dirty[j]);
            if (!open) continue;
            if (!(((gate && lab[j] < thr)) || urgent[j])) continue;
            if (bj < 0 || urgent[j] > bu || (urgent[j] == bu && lab[j] < bl)) { bj = j; bl = lab[j]; bu = urgent[j]; }
        }
        if (bj < 0) break;
        const int j = bj;
        if (inF[j]) st_reopen++; else { if (urgent[j]) dbg_urg++; else if (lab[j] >= dbg_mfree) dbg_above++; else dbg_below++; }
        inF[j] = 1; dirty[j] = 0; urgent[j] = 0;
        Ent *e = &ent[h->cnt++];
        e->row = own[j]; e->base = lab[j] - (cown[j] + price[j]); e->root = root[j]; e->col = j;
        g_base[e->row] = e->base; g_root[e->row] = e->root; g_col[e->row] = j;
    }
    for (int j = j0; j < j1; j++) {
        if (own[j] < 0) { if (lab[j] < h->minfree) h->minfree = lab[j]; continue; }
        const int open = (!inF[j] && lab[j] < INF) || (inF[j] && dirty[j]);
        if (!open) continue;
        if (urgent[j]) h->minopen = -INF;   /* a tree change that is not published yet: no END may happen */
        else if (lab[j] < h->minopen) h->minopen = lab[j];
    }
    if (h->minfree < INF)
        for (int j = j0; j < j1 && h->nend < 2; j++)
            if (own[j] < 0 && lab[j] == h->minfree) { h->ecol[h->nend] = j; h->eroot[h->nend] = root[j]; h->nend++; }
}

static void forest(LT W0)
{
    G = (n + CW - 1) / CW;
    lab = malloc(sizeof(LT) * n); price = malloc(sizeof(LT) * n); cown = malloc(sizeof(LT) * n);
    pc = malloc(4 * n); root = malloc(4 * n); own = malloc(4 * n); inF = calloc(n, 4); dirty = calloc(n, 4); need = calloc(n, 4); urgent = calloc(n, 4);
    g_base = malloc(sizeof(LT) * n); g_root = malloc(4 * n); g_col = malloc(4 * n); g_pc = malloc(4 * n);
    Ent *board = malloc(sizeof(Ent) * (size_t)G * CAP), *flat = malloc(sizeof(Ent) * (size_t)G * CAP);
    Hdr *hdr = malloc(sizeof(Hdr) * G);
    int nfree = 0;
    for (int j = 0; j < n; j++) { lab[j] = INF; price[j] = p[j]; own[j] = owner[j]; pc[j] = -1; root[j] = -1; g_pc[j] = -1;
        cown[j] = own[j] >= 0 ? c[(int64_t)own[j] * n + j] : 0; }
    for (int i = 0; i < n; i++) {
        g_base[i] = INF; g_root[i] = -1; g_col[i] = -1;
        if (r2c[i] < 0) {   /* init kernel: row dual of a free row */
            LT m = INF; for (int j = 0; j < n; j++) { LT v = c[(int64_t)i * n + j] + p[j]; if (v < m) m = v; }
            g_base[i] = -m; g_root[i] = i; nfree++;
        }
    }
#pragma omp parallel for
    for (int w = 0; w < G; w++) relax_all(w);
    st_rowscans += nfree;
    LT W = W0, gdlo = 0, gmfree = INF, ext = 0; int gate = 0, tight = 0, ext_on = 0;   /* first board: headers only */
    LT *srt = malloc(sizeof(LT) * G);
    long guard = 0;
    int32_t *path = malloc(4 * (n + 1)), *rl = malloc(4 * n), *oldown = malloc(4 * n);
    while (nfree > 0) {
        if (++guard > 64l * n + 1000) { fprintf(stderr, "level guard\n"); exit(8); }
        LT thr = (gdlo >= INF) ? INF : (gdlo <= -INF ? -INF : gdlo + W);
        { static LT WX = -1; if (WX < 0) WX = getenv("WX") ? atoll(getenv("WX")) : 16; if (gmfree < INF && thr > gmfree + WX) thr = gmfree + WX; }
        if (tight) thr = gmfree;
        if (ext_on) { thr = (gdlo >= INF) ? INF : (gdlo <= -INF ? -INF : gdlo + W); if (thr > ext || tight) thr = ext; }   /* deferred END: everything below ext */
        dbg_mfree = INF; for (int j = 0; j < n; j++) if (own[j] < 0 && lab[j] < dbg_mfree) dbg_mfree = lab[j];
        for (int w = 0; w < G; w++) select_wg(w, gate, thr, board + (size_t)w * CAP, hdr + w);
        /* ---- barrier; everybody reads the board */
        int tot = 0; LT ndlo = INF, nmf = INF;
        for (int w = 0; w < G; w++) {
            for (int k = 0; k < hdr[w].cnt; k++) flat[tot++] = board[(size_t)w * CAP + k];
            if (hdr[w].minopen < ndlo) ndlo = hdr[w].minopen;
            if (hdr[w].minfree < nmf) nmf = hdr[w].minfree;
        }
        if (tot > 0) {
            st_levels++; st_rowscans += tot;
#pragma omp parallel for
            for (int w = 0; w < G; w++) relax(w, flat, tot);
            if (tot < LO && ndlo < nmf) W = W * 2 < ((LT)1 << 40) ? W * 2 : W; else if (tot > HI && W > 1) W /= 2;
            /* minima of the columns that REMAINED open: what the relax just done opened is not in them.  With
             * nothing left open the next selection takes whatever opened below the smallest free label + W */
            gdlo = ndlo; gmfree = nmf; gate = 1;
            tight = !(ndlo < nmf) && nmf < INF;   /* nothing known open below the free label: take only what opened below it */
            if (ext_on) tight = !(ndlo < ext);
            if (ndlo >= INF) gdlo = ext_on ? ext : nmf;
            continue;
        }
        st_emptylv++;
        if (ndlo < (ext_on ? ext : nmf)) { gdlo = ndlo; gmfree = nmf; gate = 1; tight = 0; continue; }   /* gate was closed, work appeared */
        if (nmf >= INF) { fprintf(stderr, "no path (nfree=%d)\n", nfree); exit(3); }
        if (!ext_on && QDIV > 0 && nfree > 1) {   /* would-be END at nmf: defer it to the q-th smallest free label the headers know */
            int q = nfree / QDIV; if (q < 1) q = 1; if (q > RELMAX) q = RELMAX;
            int m = 0; for (int w = 0; w < G; w++) if (hdr[w].minfree < INF) srt[m++] = hdr[w].minfree;
            for (int a = 1; a < m; a++) { LT v = srt[a]; int b = a - 1; while (b >= 0 && srt[b] > v) { srt[b + 1] = srt[b]; b--; } srt[b + 1] = v; }
            const LT tgt = srt[(q - 1 < m) ? q - 1 : m - 1];
            if (tgt > nmf) { ext = tgt; ext_on = 1; st_exts++; gdlo = (ndlo >= INF) ? ext : ndlo; gmfree = nmf; gate = 1; tight = !(ndlo < ext); continue; }
        }
        /* ---- END at D: workgroup 0 flips the paths */
        st_endsteps++;
        const LT D = ext_on ? ext : nmf;   /* every label below D is exact; free labels <= D are exact */
        ext_on = 0;
        int nrl = 0;
        for (int w = 0; w < G; w++) for (int e = 0; e < hdr[w].nend; e++) {
            if (hdr[w].minfree > D) continue;
            const int r = hdr[w].eroot[e]; int dup = 0;
            for (int k = 0; k < nrl; k++) if (rl[k] == r) dup = 1;
            if (dup || nrl >= RELMAX) continue;
            rl[nrl++] = r;
            int len = 0, j = hdr[w].ecol[e];
            while (j >= 0) { path[len++] = j; j = g_pc[j]; if (len > n) { fprintf(stderr, "path cycle at END D=%ld root %d\n", (long)D, r);
                for (int k = 0; k < 12; k++) { int q = path[k]; fprintf(stderr, "  col %d lab=%ld pc=%d root=%d inF=%d dirty=%d own=%d gbase(own)=%ld groot(own)=%d\n", q, (long)lab[q], pc[q], root[q], inF[q], dirty[q], own[q], own[q] >= 0 ? (long)g_base[own[q]] : -1, own[q] >= 0 ? g_root[own[q]] : -1); }
                exit(5); } }
            for (int k = 0; k < len; k++) oldown[k] = owner[path[k]];
            for (int k = 0; k < len; k++) { const int nr = (k + 1 < len) ? oldown[k + 1] : r; owner[path[k]] = nr; r2c[nr] = path[k]; }
            g_base[r] = INF;
            st_ends++; nfree--;
        }
        /* ---- barrier B1; every workgroup releases its part of the trees and re-reads the owners */
        for (int j = 0; j < n; j++) {
            int hit = 0;
            if (lab[j] < INF) for (int k = 0; k < nrl; k++) if (root[j] == rl[k]) hit = 1;
            if (hit) {
                if (inF[j]) { if (lab[j] < D) price[j] += D - lab[j]; g_base[own[j]] = INF; inF[j] = 0; dirty[j] = 0; urgent[j] = 0; dbg_rel++; }
                else if (own[j] < 0 && lab[j] < D) price[j] += D - lab[j];   /* a free column inside the settled part of a released tree */
                lab[j] = INF; pc[j] = -1; g_pc[j] = -1; root[j] = -1; need[j] = 1;
            }
            if (own[j] != owner[j]) { own[j] = owner[j]; cown[j] = c[(int64_t)own[j] * n + j]; }
        }
        /* ---- barrier B2; repairs */
        for (int w = 0; w < G; w++) {
            int any = 0; for (int j = w * CW; j < (w + 1) * CW && j < n; j++) if (need[j]) { any = 1; need[j] = 0; }
            if (any) relax_all(w);
        }
        gdlo = nmf; gmfree = nmf; gate = 1; tight = 0;   /* the next free label is not known yet: stay within WX of the last one */
        if (verbose) printf("   END D=%ld: %d trees, free=%d levels=%ld W=%ld\n", (long)D, nrl, nfree, st_levels, (long)W);
    }
    for (int j = 0; j < n; j++) p[j] = price[j];
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "g2";
    n = argc > 2 ? atoi(argv[2]) : 1000;
    int arr_rounds = argc > 3 ? atoi(argv[3]) : 12;
    int warm = argc > 4 ? atoi(argv[4]) : 0;
    uint64_t seed = argc > 5 ? strtoull(argv[5], 0, 10) : 1;
    LT W0 = argc > 6 ? atoll(argv[6]) : 16;
    if (argc > 7) CW = atoi(argv[7]);
    if (argc > 8) CAP = atoi(argv[8]);
    verbose = argc > 9 ? atoi(argv[9]) : 0;
    if (getenv("LO")) LO = atoi(getenv("LO")); if (getenv("HI")) HI = atoi(getenv("HI"));
    if (getenv("QDIV")) QDIV = atoi(getenv("QDIV")); if (getenv("RELMAX")) RELMAX = atoi(getenv("RELMAX"));
    c = gen(kind, n, seed);
    if (getenv("SCALE")) { const int K = atoi(getenv("SCALE")); int32_t *cw = (int32_t *)c; for (int64_t k = 0; k < (int64_t)n * n; k++) cw[k] *= K; }
    p = calloc(n, 8); r2c = malloc(4 * n); owner = malloc(4 * n); list = malloc(4 * n); pick = malloc(4 * n);
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
            if (eps == 1) {
                if (getenv("PURE")) {   /* eps-scaling to the end on costs * K, K > n: run the last phase out, no exact finisher */
                    long r0 = st_rounds, s0 = st_scans, tail = 0;
                    while (U > 0) { if (U <= n / 64) tail++; U = jacobi_round(U, eps); }
                    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
                    int64_t lb = 0; for (int i = 0; i < n; i++) { int64_t m = INT64_MAX; for (int j = 0; j < n; j++) { int64_t v = c[(int64_t)i * n + j] + p[j]; if (v < m) m = v; } lb += m; }
                    for (int j = 0; j < n; j++) lb -= p[j];
                    const int K = atoi(getenv("SCALE"));
                    printf("PURE: rounds=%ld (last phase %ld, of them %ld with <= n/64 bidders), row scans/n=%.2f (last phase %.2f) total=%ld (/K = %ld rem %ld) gap to the dual bound=%ld (K=%d) %s\n",
                           st_rounds, st_rounds - r0 + r, tail, (double)st_scans / n, (double)(st_scans - s0) / n, (long)tot, (long)(tot / K), (long)(tot % K), (long)(tot - lb), K, (tot - lb < K) ? "PROVEN" : "unproven");
                    return 0;
                }
                break;
            }
            eps /= warm; if (eps < 1) eps = 1;
        }
        for (int i = 0; i < n; i++) { r2c[i] = owner[i] = -1; list[i] = i; } U = n;
    }
    long wr = st_rounds;
    int r = 0;
    while (U > 0 && r < arr_rounds) { U = jacobi_round(U, 0); r++; }
    printf("  warm rounds=%ld, ARR %d rounds, free=%d\n", wr, r, U);
    forest(W0);
    int64_t tot = 0; for (int i = 0; i < n; i++) tot += c[(int64_t)i * n + r2c[i]];
    /* duals: u_i = c[i][r2c[i]] + p[r2c[i]] (tight pairs); certificate + feasibility */
    int64_t dual = 0; long bad = 0, perm = 0;
    for (int i = 0; i < n; i++) { if (r2c[i] < 0 || owner[r2c[i]] != i) perm++; }
#pragma omp parallel for reduction(+:bad, dual)
    for (int i = 0; i < n; i++) {
        const int64_t ui = c[(int64_t)i * n + r2c[i]] + p[r2c[i]];
        dual += ui - p[i < n ? i : 0] * 0;
        for (int j = 0; j < n; j++) if (c[(int64_t)i * n + j] + p[j] < ui) bad++;
    }
    for (int j = 0; j < n; j++) dual -= p[j];
    printf("%s n=%d W0=%ld CW=%d CAP=%d: total=%ld dual=%ld %s infeasible=%ld notperm=%ld | levels=%ld empty=%ld endsteps=%ld ends=%ld rowscans/n=%.2f reopen/n=%.2f repair_rows/n=%.2f\n",
           kind, n, (long)W0, CW, CAP, (long)tot, (long)dual, (tot == dual && !bad && !perm) ? "OK" : "MISMATCH", bad, perm,
           st_levels, st_emptylv, st_endsteps, st_ends, (double)st_rowscans / n, (double)st_reopen / n, (double)st_repair_rows / n / G);
    printf("   deferred ENDs: %ld\n", st_exts);
    printf("   fresh joins: below mfree %.1f/n, above %.1f/n ; released forest cols %.1f/n\n", (double)dbg_below / n, (double)dbg_above / n, (double)dbg_rel / n);
    return (tot == dual && !bad && !perm) ? 0 : 1;
}
