// td_forest.h — included by td_assign.hip (inside its anonymous namespace, after the finishers).
//
// k_forest: the finisher for wide, tie-free rows (4-byte cells: the |a-b| geometry, 2-D grids, uniform
// 0..10^6) as ONE cooperative launch over all CUs — an INCREMENTAL shortest-path FOREST:
//
//   * every free row is a root, all trees grow at once under one common label scale (multi-source
//     Dijkstra on the reduced costs c + p - u >= 0 that the eps = 0 bidding rounds leave);
//   * workgroup g owns the columns [g*CW, (g+1)*CW): labels, predecessor COLUMN, tree ROOT, price, owner
//     live in its LDS; nothing about a column is ever written by another workgroup;
//   * a LEVEL = every workgroup publishes up to FO_CAP of its open columns (smallest labels first, inside
//     a window W above the smallest open label of the previous level) + a header (smallest label that
//     stayed open, smallest label of a free column) on a board in global memory, ONE grid barrier, then
//     every workgroup relaxes its columns against all published rows.  Label-correcting: a column whose
//     label drops after it was published is re-opened; labels below the end label are exact when no
//     column is open below it (the argument of k_sapx / DESIGN.md §2);
//   * END: a board without entries whose smallest open label is not below the smallest free-column label
//     D.  Workgroup 0 flips the augmenting path of every tree that ends at D (predecessor columns in LDS),
//     the trees are RELEASED at once (prices raised by D - label where the label is below D), the other
//     trees stay; the columns that lose their label are repaired by relaxing them against the rows that
//     are still in the forest;
//   * a column whose TREE changes (re-parented at a lower or equal label) is "urgent": it is published
//     whatever the window, and no END may happen while one is pending — at every END the root stored
//     with a label is the root of its predecessor chain.
//
// tools/forest3_model.c is the CPU model of exactly this protocol (same board, gate, window and END
// rules); it was used to validate exactness (dual == total, all reduced costs >= 0) before this ran.
//
// Inter-workgroup data (board, g_base / g_root / g_col, g_pc, owner, r2c) is written and read with
// agent-scope relaxed atomics (sc1: L2-coherent, bypassing the per-CU L1) and ordered by the grid
// barrier (monotonic counter, release / acquire fences at agent scope, bounded spin -> abort).

constexpr int FO_CAP = 4;      // entries a workgroup publishes per level
constexpr int FO_T = 256;      // threads per workgroup
constexpr int FO_GMAX = 256;   // workgroups (one per CU)
constexpr int FO_RELMAX = 16;  // trees released per END
constexpr int FO_EMAX = FO_GMAX * FO_CAP;

struct FoBoardWg {             // one 128-byte line per workgroup and parity
    unsigned long long w[16];  // [0] minopen [1] minfree [2] end (col+1 | (root+1) << 20) [3] spare, [4 + 2i] base, [5 + 2i] packed entry
};
struct FoLine {                // one 128-byte line per counter: the pollers of one XCD do not disturb the others
    unsigned long long v;
    unsigned long long pad[15];
};
struct FoShared {
    unsigned long long bar;    // start-up barrier (flat)
    int abort;
    int pad0;
    long long stat[16];        // levels, entries, ENDs, empty levels, repair rows, released trees, spare, spare; cycles of workgroup 0: select, barrier, board, relax, END, repair
    unsigned long long rel[4 + FO_RELMAX];   // [0] count [1] D (as bits) then the released roots
    unsigned long long pad1[4];
    FoLine xcnt[8];            // members of every XCD (counted at start-up)
    FoLine xarr[8];            // arrivals per XCD
    FoLine xgen[8];            // generation per XCD, set by the XCD's last arriver once every XCD has arrived
    FoLine top;                // XCDs that have arrived
    FoBoardWg board[2][FO_GMAX];
};

#define FO_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FO_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

typedef unsigned int fo_v4u __attribute__((ext_vector_type(4)));
typedef const fo_v4u __attribute__((address_space(1))) *fo_gvec;

// 16-byte agent-coherent (sc1) load through a buffer descriptor: base must be wave-uniform
__device__ __forceinline__ fo_v4u fo_ld16(const void *base, uint32_t bytes, uint32_t byte_off)
{
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);
}

template <typename LT>
struct FoLim {
    static constexpr LT INF = sizeof(LT) == 4 ? (LT)(1 << 30) : (LT)((long long)1 << 60);
    static constexpr LT WMAX = sizeof(LT) == 4 ? (LT)(1 << 26) : (LT)((long long)1 << 40);
};

// row duals of the free rows: base = -u, root = the row itself, no column
template <typename CT>
__global__ __launch_bounds__(256) void k_forest_init(int n, int nchunks, const ShardTab tab, const typename Tr<CT>::PT *__restrict__ pk,
                                                     const int *__restrict__ list, const int *__restrict__ ctl, long long *g_base,
                                                     int *g_root, int *g_col)
{
    using PT = typename Tr<CT>::PT;
    typedef typename std::conditional<sizeof(PT) == 4, int, long long>::type LT;
    constexpr int E = Tr<CT>::E;
    if (ctl[CTL_FLAG]) return;
    const int nfree = ctl[CTL_NFREE];
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= nfree) return;
    const int f = list[wv];
    const size_t pitch = (size_t)nchunks * E;
    const CT *row = shard_row<CT>(tab, f, pitch);
    LT m = FoLim<LT>::INF;
    for (int ch = lane; ch < nchunks; ch += 64) {
        const uint4 cv = *reinterpret_cast<const uint4 *>(row + (size_t)ch * E);
        uint32_t c[E];
        unpack<CT>(cv, c);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int j = ch * E + e;
            if (j < n) {
                const LT v = (LT)c[e] + (LT)(pk[j] >> 1);
                m = v < m ? v : m;
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const LT o = __shfl_xor(m, s);
        m = o < m ? o : m;
    }
    if (lane == 0) {
        g_base[f] = -(long long)m;
        g_root[f] = f;
        g_col[f] = -1;
    }
}

__global__ void k_forest_fill(int n, long long *g_base, int *g_root, int *g_col, int *g_pc, long long inf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        g_base[i] = inf;
        g_root[i] = -1;
        g_col[i] = -1;
        g_pc[i] = -1;
    }
}

// CW columns per workgroup (64 or 128), 4-byte cells only
template <typename CT, int CW, int TB>
__global__ __launch_bounds__(TB) void k_forest(int n, int nchunks, const ShardTab tab, typename Tr<CT>::PT *__restrict__ pk, int *owner_g,
                                                 int *r2c_g, int *g_pc, long long *g_base, int *g_root, int *g_col,
                                                 int *__restrict__ ctl, FoShared *sh, long long w0, long long wx, int pc_in_lds)
{
    using PT = typename Tr<CT>::PT;
    typedef typename std::conditional<sizeof(PT) == 4, int, long long>::type LT;
    static_assert(Tr<CT>::E == 4, "4-byte cells");
    constexpr LT INF = FoLim<LT>::INF;
    constexpr int SEGL = CW / 4;          // lanes per row segment (16 bytes each)
    constexpr int NS = TB / SEGL;       // row slots relaxed at once
    constexpr int NWV = TB / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char fo_dyn[];   // predecessor columns (workgroup 0 at an END) / forest row list (repairs)
    // ---- per-column state of this workgroup
    __shared__ LT s_lab[CW], s_price[CW], s_cown[CW];
    __shared__ int s_pc[CW], s_root[CW], s_own[CW];
    __shared__ unsigned char s_inF[CW], s_dirty[CW], s_urgent[CW], s_need[CW];
    // ---- entries of the level being relaxed
    __shared__ LT s_ebase[FO_EMAX];
    __shared__ int s_erow[FO_EMAX], s_eroot[FO_EMAX], s_ecol[FO_EMAX];
    // ---- reductions
    __shared__ LT s_rh[TB / 64 * CW];
    __shared__ int s_rk[FO_EMAX];   // NWV * CW partial results of a relax; the path of an augmentation
    __shared__ LT s_wk[NWV], s_wk2[NWV];
    __shared__ int s_wj[NWV], s_wcnt[NWV + 1];
    __shared__ int s_ok, s_any, s_tot;
    __shared__ int s_wflag[TB / 64];
    __shared__ LT s_gd, s_gm;
    __shared__ int s_rel[FO_RELMAX], s_nrel;
    __shared__ LT s_hmf[FO_GMAX];
    __shared__ unsigned long long s_he[FO_GMAX];

    if (ctl[CTL_FLAG]) return;
    int nfree = ctl[CTL_NFREE];
    if (nfree <= 0) return;
    const int nfree0 = nfree;   // rows handed to the finisher (reported like the other finishers do)
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t pitch = (size_t)nchunks * 4;
    const CT *const cc0 = reinterpret_cast<const CT *>(tab.p[0]);   // one shard (host checks): no table lookup per row
    // Workgroups are dealt round-robin over the 8 XCDs (observed, speed only): give every XCD a CONTIGUOUS range of
    // column slices, so that the 32 CUs behind one L2 read one contiguous 8 KiB piece of every published row.
    const int slice = (G % 8 == 0) ? (wg % 8) * (G / 8) + wg / 8 : wg;
    const int j0 = slice * CW;
    const bool colthr = tid < CW;
    const int jc = j0 + tid;                       // the column of a column thread
    const bool cvalid = colthr && jc < n;
    unsigned long long epoch = 0;
    // Grid barrier.  Every inter-workgroup word is written with an agent-scope (sc1, write-through) store and read with
    // an agent-scope load, so no release / acquire fence is needed (MI355X_MICROARCH.md, hand-off table row 1): every
    // storing wave drains its stores, the workgroup barrier orders them before ONE lane's atomic add, the consumers
    // poll relaxed.  Bounded spin: a workgroup that waits too long raises `abort`, everybody leaves.
    int xcc = 0, xmembers = 0, nxcd = 0;
    auto grid_sync = [&]() __attribute__((always_inline)) -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int ok = 1;
            const unsigned long long e1 = epoch + 1ull;
            const unsigned long long old = atomicAdd(&sh->xarr[xcc].v, 1ull);
            if (old + 1ull == e1 * (unsigned long long)xmembers) {   // the last of this XCD
                atomicAdd(&sh->top.v, 1ull);
                const unsigned long long target = e1 * (unsigned long long)nxcd;
                for (long long spins = 0;; spins++) {
                    if (FO_LD(&sh->top.v) >= target) break;
                    if (spins > 4000000ll || FO_LD(&sh->abort)) {
                        FO_ST(&sh->abort, 1);
                        ok = 0;
                        break;
                    }
                }
                FO_ST(&sh->xgen[xcc].v, e1);
            } else {
                for (long long spins = 0;; spins++) {
                    if (FO_LD(&sh->xgen[xcc].v) >= e1) break;
                    if (spins > 4000000ll || FO_LD(&sh->abort)) {
                        FO_ST(&sh->abort, 1);
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            s_ok = ok;
        }
        epoch++;
        __syncthreads();
        return s_ok != 0;
    };
    {   // start-up: count the members of every XCD (placement is whatever the dispatcher did), one flat barrier
        if (tid == 0) {
            xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);   // HW_REG_XCC_ID[3:0]
            atomicAdd(&sh->xcnt[xcc].v, 1ull);
            atomicAdd(&sh->bar, 1ull);
            int ok = 1;
            for (long long spins = 0;; spins++) {
                if (FO_LD(&sh->bar) >= (unsigned long long)G) break;
                if (spins > 4000000ll || FO_LD(&sh->abort)) {
                    FO_ST(&sh->abort, 1);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            int nx = 0;
            for (int q = 0; q < 8; q++) nx += FO_LD(&sh->xcnt[q].v) ? 1 : 0;
            s_wj[0] = xcc;
            s_wj[1] = (int)FO_LD(&sh->xcnt[xcc].v);
            s_wj[2] = nx;
            s_ok = ok;
        }
        __syncthreads();
        xcc = s_wj[0];
        xmembers = s_wj[1];
        nxcd = s_wj[2];
        if (!s_ok) return;
        __syncthreads();
    }
    // ---- load the slice
    if (colthr) {
        LT pr = 0, co = 0;
        int ow = -2;
        if (cvalid) {
            pr = (LT)(pk[jc] >> 1);
            ow = owner_g[jc];
            if (ow >= 0) co = (LT)(uint32_t)cc0[(size_t)ow * pitch + jc];
        }
        s_lab[tid] = INF;
        s_price[tid] = pr;
        s_cown[tid] = co;
        s_pc[tid] = -1;
        s_root[tid] = -1;
        s_own[tid] = ow;
        s_inF[tid] = 0;
        s_dirty[tid] = 0;
        s_urgent[tid] = 0;
        s_need[tid] = 0;
    }
    __syncthreads();

    // ---- relax this workgroup's columns against the M entries in s_e*
    long long tr_main = 0, tr_comb = 0, te_lpc = 0, te_walk = 0;
    auto relax = [&](int M, bool only_need) __attribute__((always_inline)) {
        const long long ra = clock64();
        const int seg = tid % SEGL, slot = tid / SEGL;
        const int jb = j0 + seg * 4;
        LT best[4];
        int bk[4], mypc[4], myroot[4];
        bool bs[4];
        bool want = !only_need;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int q = seg * 4 + x;
            mypc[x] = s_pc[q];
            myroot[x] = s_root[q];
            const LT lb = s_lab[q];
            best[x] = (lb >= INF) ? INF : lb - s_price[q];   // only an improvement (or a tie) of the current label matters
            bk[x] = -1;
            bs[x] = false;
            want = want || s_need[q];
        }
        const bool segok = jb < n && want;   // the columns >= n of the last segment are masked when the result is applied
        constexpr int UNR = TB >= 1024 ? 4 : 8;
        // one batch: UNR rows in flight per lane; the next batch is issued before this one is looked at
        auto issue = [&](int k0, fo_v4u (&cv)[UNR]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int k = min(k0 + u * NS, M - 1);
                cv[u] = *(fo_gvec)(uintptr_t)(cc0 + (size_t)s_erow[k] * pitch + jb);   // global address space: a flat load would wait on lgkmcnt too
            }
        };
        auto eat = [&](int k0, const fo_v4u (&cv)[UNR]) __attribute__((always_inline)) {
            LT bb[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) bb[u] = (k0 + u * NS < M) ? s_ebase[k0 + u * NS] : INF;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                LT m = INF;
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    const LT h = bb[u] + (LT)cv[u][x];
                    m = h < m ? h : m;
                }
                if (m <= best[x]) {   // rare once the labels have settled: which entry, and is it the column's own predecessor
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const int k = k0 + u * NS;
                        const LT h = bb[u] + (LT)cv[u][x];
                        if (k < M && h <= best[x]) {
                            const int ec = s_ecol[k], er = s_eroot[k];
                            const bool same = (ec == mypc[x]) && (ec >= 0 || er == myroot[x]);
                            if (h < best[x] || (same && !bs[x])) {
                                best[x] = h;
                                bk[x] = k;
                                bs[x] = same;
                            }
                        }
                    }
                }
            }
        };
        if (segok) {
            fo_v4u ca[UNR], cb[UNR];
            int k0 = slot;
            if (k0 < M) issue(k0, ca);
            while (k0 < M) {
                const int k1 = k0 + NS * UNR;
                if (k1 < M) issue(k1, cb);
                eat(k0, ca);
                if (k1 >= M) break;
                const int k2 = k1 + NS * UNR;
                if (k2 < M) issue(k2, ca);
                eat(k1, cb);
                k0 = k2;
            }
        }
        const long long rb = clock64();
        // the slots of one wave first (lanes seg, seg + SEGL, ...), then the partial results of the waves that have any
        // through LDS; total order of the candidates: smaller label, then the column's own predecessor, then the
        // smaller entry index.  Late levels improve few labels: a wave without a candidate skips all of it.
        const bool anyc = (bk[0] & bk[1] & bk[2] & bk[3]) >= 0;   // some bk >= 0 (all -1 gives -1)
        const bool wany = __any(anyc);
        if (wany) {
#pragma unroll
            for (int sx = SEGL; sx < 64; sx <<= 1) {
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const LT ob = __shfl_xor(best[x], sx);
                    const int ok_ = __shfl_xor(bk[x], sx);
                    const bool os = __shfl_xor((int)bs[x], sx) != 0;
                    if (ob < best[x] || (ob == best[x] && ((os && !bs[x]) || (os == bs[x] && (unsigned)ok_ < (unsigned)bk[x])))) {
                        best[x] = ob;
                        bk[x] = ok_;
                        bs[x] = os;
                    }
                }
            }
            if (lane < SEGL) {
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    s_rh[wv * CW + seg * 4 + x] = best[x];
                    s_rk[wv * CW + seg * 4 + x] = bk[x];
                }
            }
        }
        if (lane == 0) s_wflag[wv] = wany ? 1 : 0;
        __syncthreads();
        if (cvalid) {
            LT b = INF;
            int k = -1;
            bool ksame = false;
            const int pcj = s_pc[tid], rtj = s_root[tid];
            for (int s2 = 0; s2 < NWV; s2++) {
                if (!s_wflag[s2]) continue;
                const LT h = s_rh[s2 * CW + tid];
                const int kk = s_rk[s2 * CW + tid];
                if (kk < 0) continue;
                const bool same = (s_ecol[kk] == pcj) && (s_ecol[kk] >= 0 || s_eroot[kk] == rtj);
                if (k < 0 || h < b || (h == b && same && !ksame) || (h == b && same == ksame && kk < k)) {
                    b = h;
                    k = kk;
                    ksame = same;
                }
            }
            if (k >= 0) {
                const LT lb = s_lab[tid];
                const LT tlj = (lb >= INF) ? INF : lb - s_price[tid];
                const int ec = s_ecol[k], er = s_eroot[k];
                if (b < tlj) {
                    if (s_inF[tid]) {
                        s_dirty[tid] = 1;
                        if (rtj != er) s_urgent[tid] = 1;
                    }
                    s_lab[tid] = b + s_price[tid];
                    s_pc[tid] = ec;
                    s_root[tid] = er;
                    FO_ST(&g_pc[jc], ec);
                } else if (b == tlj && ec == pcj && (ec >= 0 || er == rtj) && rtj != er) {
                    s_root[tid] = er;   // the predecessor moved to another tree at the same label
                    if (s_inF[tid]) {
                        s_dirty[tid] = 1;
                        s_urgent[tid] = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (!only_need) {
            tr_main += rb - ra;
            tr_comb += clock64() - rb;
        }
    };

    // ---- relax the slice against EVERY row of the forest (initial pass, repairs)
    auto relax_all = [&]() __attribute__((always_inline)) -> long long {
        unsigned short *flist = reinterpret_cast<unsigned short *>(fo_dyn);
        // stage 1: compact the forest rows in row order (deterministic): every thread takes a contiguous block of
        // rows, one block-wide exclusive scan of the counts
        {
            const int per = (n + TB - 1) / TB;   // <= 128 (n <= 32 768)
            const int lo = tid * per;
            unsigned long long m0 = 0, m1 = 0;
            for (int r = 0; r < per; r++) {
                const int i = lo + r;
                const bool in = i < n && (LT)FO_LD(&g_base[i]) < INF;
                if (in) {
                    if (r < 64) m0 |= 1ull << r;
                    else m1 |= 1ull << (r - 64);
                }
            }
            const int cnt = __popcll(m0) + __popcll(m1);
            int incl = cnt;
#pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) {
                const int v = __shfl_up(incl, s2);
                if (lane >= s2) incl += v;
            }
            if (lane == 63) s_wcnt[wv] = incl;
            __syncthreads();
            int off = incl - cnt;
            for (int q = 0; q < wv; q++) off += s_wcnt[q];
            for (int r = 0; r < per; r++) {
                const bool in = (r < 64) ? ((m0 >> r) & 1ull) : ((m1 >> (r - 64)) & 1ull);
                if (in) flist[off++] = (unsigned short)(lo + r);
            }
            if (tid == TB - 1) s_tot = off;
            __syncthreads();
        }
        const int F = s_tot;
        for (int f0 = 0; f0 < F; f0 += FO_EMAX) {
            const int M = min(FO_EMAX, F - f0);
            for (int k = tid; k < M; k += TB) {
                const int i = flist[f0 + k];
                s_erow[k] = i;
                s_ebase[k] = (LT)FO_LD(&g_base[i]);
                s_eroot[k] = FO_LD(&g_root[i]);
                s_ecol[k] = FO_LD(&g_col[i]);
            }
            __syncthreads();
            relax(M, true);
        }
        return (long long)F;
    };

    // ---- block-wide argmin over the column threads: (flag desc, key asc, col asc)
    auto block_argmin = [&](LT key, int col, LT &okey, int &ocol) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const LT k2 = __shfl_xor(key, s);
            const int c2 = __shfl_xor(col, s);
            if (k2 < key || (k2 == key && c2 < col)) {
                key = k2;
                col = c2;
            }
        }
        if (lane == 0) {
            s_wk[wv] = key;
            s_wj[wv] = col;
        }
        __syncthreads();
        LT bk_ = s_wk[0];
        int bc_ = s_wj[0];
#pragma unroll
        for (int q = 1; q < NWV; q++)
            if (s_wk[q] < bk_ || (s_wk[q] == bk_ && s_wj[q] < bc_)) {
                bk_ = s_wk[q];
                bc_ = s_wj[q];
            }
        __syncthreads();
        okey = bk_;
        ocol = bc_;
    };

    long long tc_sel = 0, tc_bar = 0, tc_board = 0, tc_relax = 0, tc_end = 0, tc_rep = 0;
    long long st_levels = 0, st_entries = 0, st_ends = 0, st_empty = 0, st_reprows = 0, st_trees = 0, st_need = 0;
    // ---- initial pass: every column against every free row
    if (colthr) s_need[tid] = 1;
    __syncthreads();
    st_reprows += relax_all();
    if (colthr) s_need[tid] = 0;
    __syncthreads();
    LT W = (LT)w0, gdlo = 0, gmfree = INF;
    const LT WX = (LT)wx;
    bool gate = false, tight = false;
    int par = 0;
    bool bad = false;
    const long long max_levels = 64ll * n + 4096;
    for (long long it = 0; nfree > 0; it++) {
        if (it > max_levels) {
            bad = true;
            break;
        }
        // ---------------- selection + header
        const long long t0 = clock64();
        LT thr = (gdlo >= INF) ? INF : (gdlo <= -INF ? -INF : (gdlo + W > INF ? INF : gdlo + W));
        if (gmfree < INF && thr > gmfree + WX) thr = gmfree + WX;
        if (tight) thr = gmfree;   // nothing known to be open below the free label: take only what the last relax opened below it
        FoBoardWg *mine = &sh->board[par][wg];
        if constexpr (CW == 64) {
            // one wave holds the whole slice: FO_CAP rounds of a wave argmin, no workgroup barrier inside
            if (wv == 0) {
                const LT NONE = INF + 1;
                const LT lb = s_lab[lane];
                const int ow = s_own[lane];
                bool urg = s_urgent[lane] != 0;
                bool isopen = cvalid && ow >= 0 && ((!s_inF[lane] && lb < INF) || (s_inF[lane] && s_dirty[lane]));
                int npub = 0;
                for (int r = 0; r < FO_CAP; r++) {
                    LT key = NONE;
                    if (isopen && urg) key = -INF - 1;
                    else if (isopen && gate && lb < thr) key = lb;
                    LT m = key;
#pragma unroll
                    for (int sx = 32; sx > 0; sx >>= 1) {
                        const LT o = __shfl_xor(m, sx);
                        m = o < m ? o : m;
                    }
                    if (m == NONE) break;
                    const int L = __ffsll((long long)__ballot(key == m)) - 1;
                    if (lane == L) {
                        const LT base = lb - (s_cown[lane] + s_price[lane]);
                        s_inF[lane] = 1;
                        s_dirty[lane] = 0;
                        s_urgent[lane] = 0;
                        isopen = false;
                        urg = false;
                        FO_ST(&mine->w[4 + 2 * r], (unsigned long long)(long long)base);
                        FO_ST(&mine->w[5 + 2 * r],
                              (unsigned long long)(uint32_t)ow | ((unsigned long long)(uint32_t)(s_root[lane] + 1) << 20) | ((unsigned long long)(uint32_t)(jc + 1) << 40));
                        FO_ST(&g_base[ow], (long long)base);
                        FO_ST(&g_root[ow], s_root[lane]);
                        FO_ST(&g_col[ow], jc);
                    }
                    npub++;
                }
                if (lane < FO_CAP && lane >= npub) FO_ST(&mine->w[5 + 2 * lane], ~0ull);
                LT mo = isopen ? (urg ? -INF : lb) : INF;
                const LT kf = (cvalid && ow == -1 && lb < INF) ? lb : INF;
                LT mf = kf;
#pragma unroll
                for (int sx = 32; sx > 0; sx >>= 1) {
                    const LT o1 = __shfl_xor(mo, sx), o2 = __shfl_xor(mf, sx);
                    mo = o1 < mo ? o1 : mo;
                    mf = o2 < mf ? o2 : mf;
                }
                const unsigned long long fb = __ballot(kf == mf && kf < INF);
                const int cf = fb ? (__ffsll((long long)fb) - 1) : -1;
                if (lane == 0) {
                    FO_ST(&mine->w[0], (unsigned long long)(long long)mo);
                    FO_ST(&mine->w[1], (unsigned long long)(long long)mf);
                    unsigned long long e = 0;
                    if (cf >= 0) e = (unsigned long long)(uint32_t)(j0 + cf + 1) | ((unsigned long long)(uint32_t)(s_root[cf] + 1) << 20);
                    FO_ST(&mine->w[2], e);
                }
            }
        } else {
        int npub = 0;
        for (int r = 0; r < FO_CAP; r++) {
            LT key = INF + 1;   // nothing
            int col = INT_MAX;
            if (cvalid && s_own[tid] >= 0) {
                const LT lb = s_lab[tid];
                const bool open = (!s_inF[tid] && lb < INF) || (s_inF[tid] && s_dirty[tid]);
                if (open && s_urgent[tid]) {
                    key = -INF - 1;   // urgent columns first
                    col = tid;
                } else if (open && gate && lb < thr) {
                    key = lb;
                    col = tid;
                }
            }
            LT k1;
            int c1;
            block_argmin(key, col, k1, c1);
            if (c1 == INT_MAX) break;   // uniform
            if (tid == c1) {
                const LT lb = s_lab[tid];
                const int row = s_own[tid];
                const LT base = lb - (s_cown[tid] + s_price[tid]);
                s_inF[tid] = 1;
                s_dirty[tid] = 0;
                s_urgent[tid] = 0;
                FO_ST(&mine->w[4 + 2 * r], (unsigned long long)(long long)base);
                FO_ST(&mine->w[5 + 2 * r],
                      (unsigned long long)(uint32_t)row | ((unsigned long long)(uint32_t)(s_root[tid] + 1) << 20) | ((unsigned long long)(uint32_t)(jc + 1) << 40));
                FO_ST(&g_base[row], (long long)base);
                FO_ST(&g_root[row], s_root[tid]);
                FO_ST(&g_col[row], jc);
            }
            npub++;
            __syncthreads();
        }
        if (tid < FO_CAP && tid >= npub) FO_ST(&mine->w[5 + 2 * tid], ~0ull);
        {   // header: smallest label that stayed open (-INF while an urgent column is unpublished), smallest free label + its column
            LT ko = INF, kf = INF;
            int cf = INT_MAX, co = INT_MAX;
            if (cvalid) {
                const LT lb = s_lab[tid];
                if (s_own[tid] < 0) {
                    if (s_own[tid] == -1 && lb < INF) {
                        kf = lb;
                        cf = tid;
                    }
                } else {
                    const bool open = (!s_inF[tid] && lb < INF) || (s_inF[tid] && s_dirty[tid]);
                    if (open) {
                        ko = s_urgent[tid] ? -INF : lb;
                        co = tid;
                    }
                }
            }
            LT mo, mf;
            int c_o, c_f;
            block_argmin(ko, co, mo, c_o);
            block_argmin(kf, cf, mf, c_f);
            if (tid == 0) {
                FO_ST(&mine->w[0], (unsigned long long)(long long)mo);
                FO_ST(&mine->w[1], (unsigned long long)(long long)mf);
                unsigned long long e = 0;
                if (c_f != INT_MAX) e = (unsigned long long)(uint32_t)(j0 + c_f + 1) | ((unsigned long long)(uint32_t)(s_root[c_f] + 1) << 20);
                FO_ST(&mine->w[2], e);
            }
        }
        }
        const long long t1 = clock64();
        if (!grid_sync()) {
            bad = true;
            break;
        }
        const long long t2 = clock64();
        // ---------------- everybody reads the board
        int mycnt = 0;
        unsigned long long eb[FO_CAP], ep[FO_CAP];
        LT hmo = INF, hmf = INF;
        if (tid < G) {
            const uint32_t boff = (uint32_t)tid * (uint32_t)sizeof(FoBoardWg);
            const void *bb = &sh->board[par][0];
            const fo_v4u q0 = fo_ld16(bb, sizeof(FoBoardWg) * FO_GMAX, boff);        // minopen, minfree
            const fo_v4u q1 = fo_ld16(bb, sizeof(FoBoardWg) * FO_GMAX, boff + 16);   // end, spare
            fo_v4u qe[FO_CAP];
#pragma unroll
            for (int r = 0; r < FO_CAP; r++) qe[r] = fo_ld16(bb, sizeof(FoBoardWg) * FO_GMAX, boff + 32 + 16 * r);
            hmo = (LT)(long long)(((unsigned long long)q0.y << 32) | q0.x);
            hmf = (LT)(long long)(((unsigned long long)q0.w << 32) | q0.z);
            const unsigned long long he = ((unsigned long long)q1.y << 32) | q1.x;
#pragma unroll
            for (int r = 0; r < FO_CAP; r++) {
                eb[r] = ((unsigned long long)qe[r].y << 32) | qe[r].x;
                ep[r] = ((unsigned long long)qe[r].w << 32) | qe[r].z;
            }
#pragma unroll
            for (int r = 0; r < FO_CAP; r++) mycnt += (ep[r] != ~0ull) ? 1 : 0;
            s_hmf[tid] = hmf;
            s_he[tid] = he;
        }
        {
            int incl = mycnt;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int v = __shfl_up(incl, s);
                if (lane >= s) incl += v;
            }
            LT a = hmo, b2 = hmf;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                const LT o1 = __shfl_xor(a, s), o2 = __shfl_xor(b2, s);
                a = o1 < a ? o1 : a;
                b2 = o2 < b2 ? o2 : b2;
            }
            if (lane == 63) s_wcnt[wv] = incl;
            if (lane == 0) {
                s_wk[wv] = a;
                s_wk2[wv] = b2;
            }
            __syncthreads();
            int off = incl - mycnt;
            for (int q = 0; q < wv; q++) off += s_wcnt[q];
            if (tid < G) {
                int o = off;
#pragma unroll
                for (int r = 0; r < FO_CAP; r++)
                    if (ep[r] != ~0ull) {
                        s_ebase[o] = (LT)(long long)eb[r];
                        s_erow[o] = (int)(ep[r] & 0xFFFFFu);
                        s_eroot[o] = (int)((ep[r] >> 20) & 0xFFFFFu) - 1;
                        s_ecol[o] = (int)((ep[r] >> 40) & 0xFFFFFu) - 1;
                        o++;
                    }
            }
            if (tid == 0) {
                int t = 0;
                LT x = INF, y = INF;
                for (int q = 0; q < NWV; q++) {
                    t += s_wcnt[q];
                    x = s_wk[q] < x ? s_wk[q] : x;
                    y = s_wk2[q] < y ? s_wk2[q] : y;
                }
                s_tot = t;
                s_gd = x;
                s_gm = y;
            }
            __syncthreads();
        }
        const int tot = s_tot;
        const LT ndlo = s_gd, nmf = s_gm;
        par ^= 1;
        const long long t3 = clock64();
        tc_sel += t1 - t0;
        tc_bar += t2 - t1;
        tc_board += t3 - t2;
        if (tot > 0) {
            st_levels++;
            st_entries += tot;
            relax(tot, false);
            tc_relax += clock64() - t3;
            if (tot < 32 && ndlo < nmf)
                W = (W * 2 < FoLim<LT>::WMAX) ? W * 2 : W;
            else if (tot > 256 && W > 1)
                W /= 2;
            // The minima are those of the columns that STAYED open; what the relax just opened is not in them.  So the
            // gate stays open after a level with entries: when nothing is known to be open below the free label, the
            // next selection takes exactly the labels below it (tight), and an empty board then IS the END.
            gdlo = ndlo;
            gmfree = nmf;
            gate = true;
            tight = !(ndlo < nmf) && nmf < INF;
            if (ndlo >= INF) gdlo = nmf;
            continue;
        }
        st_empty++;
        if (ndlo < nmf) {   // the gate was closed and lower labels appeared
            gdlo = ndlo;
            gmfree = nmf;
            gate = true;
            tight = false;
            continue;
        }
        if (nmf >= INF) {   // free rows but no path: cannot happen on a complete cost matrix
            bad = true;
            break;
        }
        // ---------------- END at D
        st_ends++;
        const LT D = nmf;
        if (wg == 0) {
            const int n4 = (n + 3) / 4;
            int *lpc = reinterpret_cast<int *>(fo_dyn);
            unsigned short *lown = reinterpret_cast<unsigned short *>(fo_dyn + (size_t)n4 * 16);
            const long long ea = clock64();
            if (pc_in_lds) {   // snapshots of the predecessor columns and of the owners (both arrays are padded to 4 ints)
                for (int i = tid; i < n4; i += TB) {
                    const fo_v4u t = fo_ld16(g_pc, (uint32_t)n4 * 16u, (uint32_t)i * 16u);
                    *reinterpret_cast<fo_v4u *>(lpc + 4 * i) = t;
                    const fo_v4u o = fo_ld16(owner_g, (uint32_t)n4 * 16u, (uint32_t)i * 16u);
                    lown[4 * i + 0] = (unsigned short)o.x;
                    lown[4 * i + 1] = (unsigned short)o.y;
                    lown[4 * i + 2] = (unsigned short)o.z;
                    lown[4 * i + 3] = (unsigned short)o.w;
                }
            }
            // the ends: headers whose smallest free label is D, in workgroup order
            const bool isc = tid < G && s_hmf[tid] == D && s_he[tid] != 0;
            const unsigned long long cb = __ballot(isc);
            if (lane == 0) s_wcnt[wv] = __popcll(cb);
            __syncthreads();
            {
                int off = 0;
                for (int q = 0; q < wv; q++) off += s_wcnt[q];
                if (isc) s_erow[off + __popcll(cb & ((1ull << lane) - 1ull))] = tid;   // s_e* is free between two levels
            }
            int ncand = 0;
            for (int q = 0; q < NWV; q++) ncand += s_wcnt[q];
            __syncthreads();
            if (tid == 0) {   // one end per tree, at most FO_RELMAX trees (the others end at the next END, same D)
                int nr = 0;
                for (int ci = 0; ci < ncand && nr < FO_RELMAX; ci++) {
                    const unsigned long long e = s_he[s_erow[ci]];
                    const int ecol = (int)(e & 0xFFFFFu) - 1, eroot = (int)((e >> 20) & 0xFFFFFu) - 1;
                    bool dup = eroot < 0;
                    for (int k = 0; k < nr; k++) dup = dup || (s_rel[k] == eroot);
                    if (dup) continue;
                    s_rel[nr] = eroot;
                    s_ecol[nr] = ecol;
                    nr++;
                }
                s_nrel = nr;
            }
            __syncthreads();
            te_lpc += clock64() - ea;
            const int nr = s_nrel;
            // the paths of different trees are vertex-disjoint: one lane walks and flips each of them, all at once
            if (pc_in_lds) {
                if (tid < nr) {
                    const int r = s_rel[tid];
                    int j = s_ecol[tid], hops = 0;
                    while (j >= 0) {
                        const int pj = lpc[j];
                        const int nrow = (pj >= 0) ? (int)lown[pj] : r;   // the predecessor column's owner BEFORE the flip
                        FO_ST(&owner_g[j], nrow);
                        FO_ST(&r2c_g[nrow], j);
                        j = pj;
                        if (++hops > n) {   // a cycle: broken predecessor chain
                            FO_ST(&sh->abort, 1);
                            break;
                        }
                    }
                    FO_ST(&g_base[r], (long long)INF);
                }
            } else if (tid == 0) {
                for (int t = 0; t < nr; t++) {
                    const int r = s_rel[t];
                    int j = s_ecol[t], prev_owner_of_pj = -1;
                    // serial walk in global memory: read the predecessor and ITS owner before writing this column
                    for (int hop = 0; hop <= n && j >= 0; hop++) {
                        const int pj = FO_LD(&g_pc[j]);
                        const int nrow = (pj >= 0) ? FO_LD(&owner_g[pj]) : r;
                        FO_ST(&owner_g[j], nrow);
                        FO_ST(&r2c_g[nrow], j);
                        j = pj;
                    }
                    (void)prev_owner_of_pj;
                    FO_ST(&g_base[r], (long long)INF);
                }
            }
            if (tid == 0) {
                FO_ST(&sh->rel[0], (unsigned long long)nr);
                FO_ST(&sh->rel[1], (unsigned long long)(long long)D);
                for (int k = 0; k < nr; k++) FO_ST(&sh->rel[4 + k], (unsigned long long)(uint32_t)s_rel[k]);
            }
        }
        if (!grid_sync()) {
            bad = true;
            break;
        }
        // ---------------- release
        if (tid == 0) s_nrel = (int)FO_LD(&sh->rel[0]);
        if (tid < FO_RELMAX) s_rel[tid] = (int)FO_LD(&sh->rel[4 + tid]);
        if (tid == 0) {
            s_any = 0;
            s_tot = 0;
        }
        __syncthreads();
        const int nrel = s_nrel;
        if (nrel <= 0 || nrel > FO_RELMAX) {
            bad = true;
            FO_ST(&sh->abort, 1);
        }
        st_trees += nrel;
        nfree -= nrel;
        if (cvalid) {
            bool hit = false;
            const int rt = s_root[tid];
            if (s_lab[tid] < INF)
                for (int k = 0; k < nrel; k++) hit = hit || (rt == s_rel[k]);
            if (hit) {
                if (s_inF[tid]) {
                    const LT lb = s_lab[tid];
                    if (lb < D) {
                        const LT np = s_price[tid] + (D - lb);
                        s_price[tid] = np;
                        if constexpr (IsNP<CT>::value) {
                            if (np >= (LT)NP_PLIMIT) {
                                atomicOr(&ctl[CTL_FLAG], 8);
                                FO_ST(&sh->abort, 1);
                            }
                        }
                    }
                    FO_ST(&g_base[s_own[tid]], (long long)INF);
                    s_inF[tid] = 0;
                    s_dirty[tid] = 0;
                    s_urgent[tid] = 0;
                }
                s_lab[tid] = INF;
                s_pc[tid] = -1;
                s_root[tid] = -1;
                FO_ST(&g_pc[jc], -1);
                s_need[tid] = 1;
                s_any = 1;
                atomicAdd(&s_tot, 1);
            }
            const int no = FO_LD(&owner_g[jc]);
            if (no != s_own[tid]) {   // a column of an augmenting path
                s_own[tid] = no;
                s_cown[tid] = (no >= 0) ? (LT)(uint32_t)cc0[(size_t)no * pitch + jc] : 0;
            }
        }
        if (!grid_sync()) {   // g_base of the released rows is now visible everywhere
            bad = true;
            break;
        }
        st_need += s_tot;
        const long long t4 = clock64();
        if (s_any) {
            st_reprows += relax_all();
            if (colthr) s_need[tid] = 0;
            __syncthreads();
        }
        tc_rep += clock64() - t4;
        tc_end += t4 - t3;
        gdlo = D;
        gmfree = D;
        gate = true;
        tight = false;
    }
    if (bad) FO_ST(&sh->abort, 1);
    // ---- write the prices back
    if (cvalid) pk[jc] = (PT)((PT)s_price[tid] << 1) | (PT)1;
    if (wg == 0 && tid == 0) {
        ctl[CTL_NFREE] = bad ? nfree : nfree0;
        ctl[CTL_STEPS] = (int)(st_levels > INT_MAX ? INT_MAX : st_levels);
        ctl[CTL_FOREST] = (int)(st_levels > INT_MAX ? INT_MAX : (st_levels > 0 ? st_levels : 1));
        if (bad) atomicOr(&ctl[CTL_ERR], 32);
        sh->stat[0] = st_levels;
        sh->stat[1] = st_entries;
        sh->stat[2] = st_ends;
        sh->stat[3] = st_empty;
        sh->stat[4] = st_reprows;
        sh->stat[5] = st_trees;
        sh->stat[8] = tc_sel;
        sh->stat[9] = tc_bar;
        sh->stat[10] = tc_board;
        sh->stat[11] = tc_relax;
        sh->stat[12] = tc_end;
        sh->stat[13] = tc_rep;
        sh->stat[6] = tr_main;
        sh->stat[7] = tr_comb;
        sh->stat[14] = te_lpc;
        sh->stat[15] = st_need;
    }
}
