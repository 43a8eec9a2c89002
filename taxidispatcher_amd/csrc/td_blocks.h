// td_blocks.h — included by td_assign.hip (inside its anonymous namespace, after k_assign / build_free_list).
//
// BLOCK-LOCAL START of a solve with 1-byte cells ("phase A") and the TWO-HOP AUGMENTATION on tight
// cells.  Both exist for the row-sharded solve of SURVEY §8e (N = 65 536 over 8 GPUs): every exchange step between
// ranks costs tens of microseconds, and the plain sequence (one MAX all-reduce of the bid keys per bidding round,
// twelve rounds, a serial finisher on rank 0) spends more time in exchanges and in the serial tail than in the
// streaming passes that actually shard.
//
// Phase A needs NO exchange.  The matrix is cut into V diagonal blocks: block b = rows [b*rpb, (b+1)*rpb) x columns
// [b*rpb, (b+1)*rpb) (rpb = n / V; a rank owns whole blocks).  Inside its block a row may only take a column whose
// reduced cell is 0 at price 0 ("zero cell").  Such a pair is tight whatever happens elsewhere (reduced cells and
// prices are >= 0, so 0 is the row's minimum), no price moves, and nobody outside the block looks at the block's
// columns — so ANY matching on the zero cells of the diagonal blocks is a valid starting state of the auction
// (complementary slackness exact, all prices 0), found without a word exchanged:
//   round 0      the 1-byte compress pass writes every row's first zero cell of its own column slice in the row's
//                rotated order (k_compress_reg<.., BID0> with zs_rpb > 0), k_zs_assign resolves the columns;
//   rounds 1..R  k_zs_bid (one workgroup per free row, 1/V of the row is read) + k_zs_assign: a free column is
//                preferred, else the row takes an owned zero column and the evicted row bids again (the tie eviction
//                of k_bid, restricted to the block);
//   two hops     k_hop_*: what the rounds leave free (a few dozen rows per block) is matched through paths
//                free row i -> tight column j -> its owner r' -> FREE column j' tight for r'.  On tie-heavy
//                instances (perf.jl: n / 31 zero cells per row) nearly every (i, j') pair is connected this way.
// After phase A the ranks exchange their owner slices ONCE (all-gather) and the ordinary global rounds and the
// finisher take whatever is still free — on the perf.jl instance nothing.
//
// The two-hop kernels are written for general prices (tight = reduced cost 0 against the row's dual), windowed to a
// block's columns only in phase A, where the zero-cell rule stands in for the row minimum.
//   k_hop_lists  per block: the free rows / free columns (block-relative, ordered), at most HOP_FMAX of each used
//   k_hop_esc    per ASSIGNED row r' (one wave): bit b set iff free column b is tight for r' (c[r'][j'] + p[j'] equals
//                c[r'][col(r')] + p[col(r')], the row's dual by complementary slackness)
//   k_hop_table  per free row i (one workgroup): tab[i][b] = the smallest r' such that col(r') is tight for i and
//                free column b is tight for r' (atomicMin in LDS: independent of scheduling)
//   k_hop_match  per block (one workgroup): rows in order take the first free column in their rotated order whose
//                r' is still unused; the winner thread rewires i -> col(r'), r' -> j'.  Deterministic.
// Nothing here changes a price, so every pair it creates is tight and the finishers' invariant holds.

// HOP_FMAX (free rows / columns of a block the two-hop pass looks at) and HOP_BMAX (blocks per shard) are defined
// with the tunables at the top of td_assign.hip.

struct HopCtl {                 // device words of the pass (one per shard)
    int nfr[HOP_BMAX], nfc[HOP_BMAX];   // free rows / columns per local block (true counts)
    int left;                           // free rows the pass left (over all local blocks)
    int matched;
    int pad[2];
};

__device__ __forceinline__ uint32_t zs_zero_bytes(uint32_t w)   // 0x80 in every byte of w that is zero (exact, no borrow)
{
    return ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w | 0x7F7F7F7Fu);
}

// ---- phase A, rounds >= 1: one workgroup per free row, zero cells of the row's own column slice only
// ob[j] = 1 once column j has an owner (phase A never changes a price: all are 0)
__global__ __launch_bounds__(256) void k_zs_bid(int nrows, int row0, int nchunks, int rpb, const uint8_t *__restrict__ cc,
                                                const uint8_t *__restrict__ ob, const int *__restrict__ r2c,
                                                unsigned long long *__restrict__ bid, const int *__restrict__ ctl, int round,
                                                int tie_evict)
{
    __shared__ int s_best[4];
    if (ctl[CTL_FLAG]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cpb = rpb >> 4;
    const size_t pitch = (size_t)nchunks * 16;
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
        if (r2c[lrow] != -1) continue;   // uniform
        const int row = row0 + lrow;
        const int c0 = (row / rpb) * cpb;   // first chunk of the row's column slice
        const uint32_t hsh = ((uint32_t)row + 1u) * 0x9E3779B1u + (uint32_t)round * 0x85EBCA6Bu;
        const int rot = (int)(((uint64_t)(hsh ^ (hsh >> 15)) * (uint64_t)cpb) >> 32);
        const uint8_t *rp = cc + (size_t)lrow * pitch;
        int best = INT_MAX;   // owned << 24 | rotated cell position
        for (int t = tid; t < cpb; t += 256) {
            int ch = t + rot;
            if (ch >= cpb) ch -= cpb;
            ch += c0;
            const uint4 cv = *reinterpret_cast<const uint4 *>(rp + (size_t)ch * 16);
            const uint4 ov = *reinterpret_cast<const uint4 *>(ob + (size_t)ch * 16);
            const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t z = zs_zero_bytes(cw[k]);
                if (!z) continue;
                const uint32_t zf = z & zs_zero_bytes(ow[k]);   // zero cell in a free column
                const uint32_t pick = zf ? zf : z;
                const int cand = (zf ? 0 : (1 << 24)) | (t * 16 + k * 4 + (__builtin_ctz(pick) >> 3));
                best = min(best, cand);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
        if (lane == 0) s_best[w] = best;
        __syncthreads();
        if (tid == 0) {
            best = min(min(s_best[0], s_best[1]), min(s_best[2], s_best[3]));
            if (best != INT_MAX) {
                const bool owned = (best >> 24) != 0;
                const int pos = best & 0xFFFFFF;
                int ch = (pos >> 4) + rot;
                if (ch >= cpb) ch -= cpb;
                const int j = (c0 + ch) * 16 + (pos & 15);
                if (!owned || tie_evict) atomicMax(&bid[j], (unsigned long long)(row + 1));   // price 0: the key is the row
            }
        }
        __syncthreads();
    }
}

// ---- phase A: resolve the bids on the columns [col_lo, col_hi) of this shard's blocks (prices stay 0)
__global__ __launch_bounds__(256) void k_zs_assign(int col_lo, int col_hi, int nrows, int row0, unsigned long long *__restrict__ bid,
                                                   int32_t *__restrict__ pk, int *__restrict__ owner, int *__restrict__ r2c,
                                                   uint8_t *__restrict__ ob, const int *__restrict__ ctl)
{
    if (ctl[CTL_FLAG]) return;
    const int j = col_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= col_hi) return;
    const unsigned long long k = bid[j];
    if (!k) return;
    const int row = (int)(k & ((1ull << ROW_BITS) - 1)) - 1;
    const int old = owner[j];
    if (old >= row0 && old < row0 + nrows) r2c[old - row0] = -1;
    owner[j] = row;
    if (row >= row0 && row < row0 + nrows) r2c[row - row0] = j;
    pk[j] = 1;   // price 0, owned
    ob[j] = 1;
    bid[j] = 0ull;
}

// ---- two hops: free rows / columns of every local block (block-relative indices, ascending)
__global__ __launch_bounds__(1024) void k_hop_lists(int rpb, int ncols_blk, int col_lo, const int *__restrict__ r2c,
                                                    const int *__restrict__ owner, int *__restrict__ frl, int *__restrict__ fcl,
                                                    HopCtl *__restrict__ hc, const int *__restrict__ ctl)
{
    if (ctl[CTL_FLAG]) return;
    const int lb = blockIdx.x;
    const int nfr = build_free_list(rpb, r2c + (size_t)lb * rpb, frl + (size_t)lb * rpb, -1);
    const int nfc = build_free_list(ncols_blk, owner + col_lo + (size_t)lb * ncols_blk, fcl + (size_t)lb * ncols_blk, -1);
    if (threadIdx.x == 0) {
        hc->nfr[lb] = nfr;
        hc->nfc[lb] = nfc;
        if (lb == 0) hc->left = 0, hc->matched = 0;
    }
}

// ---- two hops: which free columns of its block is an assigned row tight to (one wave per row)
// RAW: `cc` is the caller's int32 matrix (row pitch = nchunks * 4 = n) and a cell is c - rowmin[row]: the same value the
// narrow copy holds, for a solve whose compress pass stored the diagonal slices only (k_compress_reg diag_only).
template <typename CT, bool RAW = false>
__global__ __launch_bounds__(256) void k_hop_esc(int nrows, int nchunks, int rpb, int ncols_blk, int col_lo, int max_rows,
                                                 const CT *__restrict__ cc, const typename Tr<CT>::PT *__restrict__ pk,
                                                 const int *__restrict__ r2c, const int *__restrict__ fcl,
                                                 const HopCtl *__restrict__ hc, unsigned long long *__restrict__ esc,
                                                 const int *__restrict__ ctl, const int32_t *__restrict__ rowmin = nullptr)
{
    using PT = typename Tr<CT>::PT;
    if (ctl[CTL_FLAG]) return;
    const int lane = threadIdx.x & 63;
    const int lrow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lrow >= nrows) return;
    const int lb = lrow / rpb;
    const int nfr = hc->nfr[lb];
    if (nfr == 0 || nfr > max_rows) return;   // nothing to do for this block, or left to the rounds (uniform per wave)
    const int nfc = min(hc->nfc[lb], HOP_FMAX);
    const int j = r2c[lrow];
    unsigned long long m0 = 0, m1 = 0;
    if (j >= 0) {
        const size_t pitch = (size_t)nchunks * (RAW ? 4 : Tr<CT>::E);
        const CT *rp = cc + (RAW ? 0 : (size_t)lrow * pitch);
        const int32_t *rr = reinterpret_cast<const int32_t *>(cc) + (RAW ? (size_t)lrow * pitch : 0);
        const int32_t mn = RAW ? rowmin[lrow] : 0;
        auto cell = [&](int col) -> long long {
            if constexpr (RAW) return (long long)(uint32_t)(rr[col] - mn);
            else return (long long)(uint32_t)rp[col];
        };
        const long long u = cell(j) + (long long)(pk[j] >> 1);   // the row's dual (its pair is tight)
        const int *fc = fcl + (size_t)lb * ncols_blk;
        const int cb = col_lo + lb * ncols_blk;
        bool t0 = false, t1 = false;
        if (lane < nfc) {
            const int jb = cb + fc[lane];
            t0 = cell(jb) + (long long)(pk[jb] >> 1) == u;
        }
        if (lane + 64 < nfc) {
            const int jb = cb + fc[lane + 64];
            t1 = cell(jb) + (long long)(pk[jb] >> 1) == u;
        }
        m0 = __ballot(t0);
        m1 = __ballot(t1);
    }
    if (lane == 0) {
        esc[(size_t)lrow * 2] = m0;
        esc[(size_t)lrow * 2 + 1] = m1;
    }
}

// ---- two hops: per free row the table row tab[b] = smallest r' (global id) with col(r') tight for the row and free
// column b tight for r'.  window_zero = 1: only the block's column slice is read and "tight" means a zero cell at
// price 0 (phase A); 0: the whole row, tight against the row's minimum of c + p.
// The row is taken in segments of 256 chunks (one per thread): the tight cells of a segment go to a list in LDS first
// (a thread-private walk over its 16 cells with three dependent loads behind every hit cost 16 serialised load chains
// per wave: 19 us for a 2048-column slice), then all 256 threads take list entries — owner, escape masks, atomicMin into
// the table row.  A segment holds at most 256 * E (RAW: 256 * 4 * 4) candidates: the list cannot overflow, the result
// does not depend on the order of the appends.
template <typename CT, bool RAW = false>
__global__ __launch_bounds__(256) void k_hop_table(int n, int nrows, int row0, int nchunks, int rpb, int ncols_blk, int col_lo,
                                                   int max_rows, int window_zero, const CT *__restrict__ cc,
                                                   const typename Tr<CT>::PT *__restrict__ pk, const int *__restrict__ owner,
                                                   const int *__restrict__ frl, const HopCtl *__restrict__ hc,
                                                   const unsigned long long *__restrict__ esc, int *__restrict__ tab,
                                                   const int *__restrict__ ctl, const int32_t *__restrict__ rowmin = nullptr)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = RAW ? 4 : Tr<CT>::E;   // cells per 16 bytes (RAW: the int32 matrix itself, nchunks = n / 4, see k_hop_esc)
    __shared__ int s_tab[HOP_FMAX];
    constexpr int CHT = RAW ? 4 : 1;   // 16-byte pieces per thread and segment (RAW: 4 cells a piece, the same 4096-entry list as 1-byte cells)
    __shared__ int s_cand[256 * E * CHT];
    __shared__ int s_ncand;
    __shared__ long long s_v[4];
    if (ctl[CTL_FLAG]) return;
    const int lb = blockIdx.x / HOP_FMAX, a = blockIdx.x % HOP_FMAX;
    const int nfr = hc->nfr[lb];
    if (nfr > max_rows || a >= min(nfr, HOP_FMAX)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lrow = lb * rpb + frl[(size_t)lb * rpb + a];
    const size_t pitch = (size_t)nchunks * E;
    const unsigned char *rp = reinterpret_cast<const unsigned char *>(cc) + (size_t)lrow * pitch * (RAW ? 4 : sizeof(CT));
    const int32_t mn = RAW ? rowmin[lrow] : 0;
    auto cells = [&](const uint4 &raw, uint32_t *c) {
        if constexpr (RAW) {
            c[0] = (uint32_t)((int32_t)raw.x - mn), c[1] = (uint32_t)((int32_t)raw.y - mn);
            c[2] = (uint32_t)((int32_t)raw.z - mn), c[3] = (uint32_t)((int32_t)raw.w - mn);
        } else
            unpack<CT>(raw, c);
    };
    if (tid < HOP_FMAX) s_tab[tid] = INT_MAX;
    if (tid == 0) s_ncand = 0;
    // the row's columns in question: its block's slice (phase A) or all of them
    const int ch_lo = window_zero ? (col_lo + lb * ncols_blk) / E : 0;
    const int ch_n = window_zero ? (ncols_blk + E - 1) / E : nchunks;
    long long v = 0;
    if (!window_zero) {
        long long mv = LLONG_MAX;
        for (int t = tid; t < ch_n; t += 256) {
            uint32_t c[E];
            cells(*reinterpret_cast<const uint4 *>(rp + (size_t)(ch_lo + t) * 16), c);
            PT pv[E];
            const int j0 = (ch_lo + t) * E;
            if constexpr (sizeof(PT) == 4) {
                const int4 *pp = reinterpret_cast<const int4 *>(pk + j0);
#pragma unroll
                for (int q = 0; q < E / 4; q++) {
                    const int4 x = pp[q];
                    pv[4 * q + 0] = x.x, pv[4 * q + 1] = x.y, pv[4 * q + 2] = x.z, pv[4 * q + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) pv[e] = pk[j0 + e];
            }
#pragma unroll
            for (int e = 0; e < E; e++)
                if (j0 + e < n) mv = min(mv, (long long)c[e] + (long long)(pv[e] >> 1));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long ov = __shfl_xor(mv, o);
            mv = ov < mv ? ov : mv;
        }
        if (lane == 0) s_v[w] = mv;
        __syncthreads();
        v = min(min(s_v[0], s_v[1]), min(s_v[2], s_v[3]));
    }
    __syncthreads();
    for (int t0 = 0; t0 < ch_n; t0 += 256 * CHT) {
        uint4 raws[CHT];
#pragma unroll
        for (int u = 0; u < CHT; u++) {
            const int t = t0 + u * 256 + tid;
            if (t < ch_n) raws[u] = *reinterpret_cast<const uint4 *>(rp + (size_t)(ch_lo + t) * 16);
        }
#pragma unroll
        for (int u = 0; u < CHT; u++) {
        const int t = t0 + u * 256 + tid;
        if (t < ch_n) {
            const uint4 raw = raws[u];
            bool any = true;
            if constexpr (!RAW && sizeof(CT) == 1 && std::is_same<CT, uint8_t>::value)   // (a zero cell is a zero byte)
                if (window_zero) any = (zs_zero_bytes(raw.x) | zs_zero_bytes(raw.y) | zs_zero_bytes(raw.z) | zs_zero_bytes(raw.w)) != 0;
            if (any) {
                uint32_t c[E];
                cells(raw, c);
                PT pv[E];
                const int j0 = (ch_lo + t) * E;
                if constexpr (sizeof(PT) == 4) {
                    const int4 *pp = reinterpret_cast<const int4 *>(pk + j0);
#pragma unroll
                    for (int q = 0; q < E / 4; q++) {
                        const int4 x = pp[q];
                        pv[4 * q + 0] = x.x, pv[4 * q + 1] = x.y, pv[4 * q + 2] = x.z, pv[4 * q + 3] = x.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; e++) pv[e] = pk[j0 + e];
                }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool tight = j0 + e < n && (long long)c[e] + (long long)(pv[e] >> 1) == v && (!window_zero || c[e] == 0);
                    if (tight) s_cand[atomicAdd(&s_ncand, 1)] = j0 + e;
                }
            }
        }
        }
        __syncthreads();
        const int nc = s_ncand;
        for (int k = tid; k < nc; k += 256) {
            const int r = owner[s_cand[k]];
            if (r < row0 || r >= row0 + nrows) continue;   // a free column (then the rounds take it), or a row of another shard
            if ((r - row0) / rpb != lb) continue;          // (two hops inside the block)
            unsigned long long m0 = esc[(size_t)(r - row0) * 2], m1 = esc[(size_t)(r - row0) * 2 + 1];
            while (m0) {
                const int b = __builtin_ctzll(m0);
                m0 &= m0 - 1;
                atomicMin(&s_tab[b], r);
            }
            while (m1) {
                const int b = 64 + __builtin_ctzll(m1);
                m1 &= m1 - 1;
                atomicMin(&s_tab[b], r);
            }
        }
        __syncthreads();
        if (tid == 0) s_ncand = 0;
        __syncthreads();
    }
    if (tid < HOP_FMAX) tab[((size_t)lb * HOP_FMAX + a) * HOP_FMAX + tid] = s_tab[tid] == INT_MAX ? -1 : s_tab[tid];
}

// ---- two hops: one workgroup per block takes the table rows in order (deterministic greedy) and rewires
template <typename PT>
__global__ __launch_bounds__(HOP_FMAX) void k_hop_match(int nrows, int row0, int rpb, int ncols_blk, int col_lo, int max_rows,
                                                        PT *__restrict__ pk, int *__restrict__ owner, int *__restrict__ r2c,
                                                        uint8_t *__restrict__ ob, const int *__restrict__ frl,
                                                        const int *__restrict__ fcl, HopCtl *__restrict__ hc,
                                                        const int *__restrict__ tab, int *__restrict__ ctl)
{
    extern __shared__ uint32_t s_usedr[];   // one bit per row of the block
    __shared__ int s_usedc[HOP_FMAX];
    __shared__ int s_key[2][2];
    if (ctl[CTL_FLAG]) return;
    const int lb = blockIdx.x;
    const int nfr = hc->nfr[lb];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (nfr == 0) return;
    if (nfr > max_rows) {
        if (tid == 0) atomicAdd(&hc->left, nfr);
        return;
    }
    const int na = min(nfr, HOP_FMAX), nc = min(hc->nfc[lb], HOP_FMAX);
    for (int k = tid; k < (rpb + 31) / 32; k += HOP_FMAX) s_usedr[k] = 0u;
    s_usedc[tid] = 0;
    __syncthreads();
    const int *fr = frl + (size_t)lb * rpb, *fc = fcl + (size_t)lb * ncols_blk;
    const int blk_row0 = row0 + lb * rpb;
    const int my_fc = tid < nc ? fc[tid] : 0;
    int done = 0;
    // the table entry and the row id of the NEXT iteration are loaded while this one is decided (the loop is a chain of
    // dependent global loads otherwise: ~1 us per row)
    int r_nx = tid < nc ? tab[((size_t)lb * HOP_FMAX + 0) * HOP_FMAX + tid] : -1;
    int i_nx = blk_row0 + fr[0];
    for (int a = 0; a < na; a++) {
        const int r = r_nx, i_g = i_nx;
        if (a + 1 < na) {
            r_nx = tid < nc ? tab[((size_t)lb * HOP_FMAX + a + 1) * HOP_FMAX + tid] : -1;
            i_nx = blk_row0 + fr[a + 1];
        }
        const uint32_t hsh = ((uint32_t)i_g + 1u) * 0x9E3779B1u;
        const int st = nc ? (int)(((uint64_t)(hsh ^ (hsh >> 15)) * (uint64_t)nc) >> 32) : 0;
        int key = INT_MAX;
        if (r >= 0 && !s_usedc[tid]) {
            const int rb = r - blk_row0;
            if (!((s_usedr[rb >> 5] >> (rb & 31)) & 1u)) {
                int t = tid - st;
                t += t < 0 ? nc : 0;
                key = t;
            }
        }
        int best = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
        if (lane == 0) s_key[a & 1][w] = best;
        __syncthreads();
        best = min(s_key[a & 1][0], s_key[a & 1][1]);
        if (best != INT_MAX && key == best) {   // exactly one thread: the keys of a row are distinct
            const int rl = r - row0, il = i_g - row0;
            const int j = r2c[rl], jp = col_lo + lb * ncols_blk + my_fc;
            r2c[il] = j;
            owner[j] = i_g;
            r2c[rl] = jp;
            owner[jp] = r;
            pk[jp] = pk[jp] | (PT)1;
            if (ob) ob[jp] = 1;
            s_usedc[tid] = 1;
            const int rb = r - blk_row0;
            atomicOr(&s_usedr[rb >> 5], 1u << (rb & 31));
        }
        if (best != INT_MAX) done++;
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(&hc->left, nfr - done);
        atomicAdd(&hc->matched, done);
    }
}

// ---- after the exchange of the owner slices: the owned bit of the columns other ranks own (prices are all 0)
__global__ void k_zs_owned_bits(int n, const int *__restrict__ owner, int32_t *__restrict__ pk)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) pk[j] = (pk[j] & ~1) | (owner[j] >= 0 ? 1 : 0);
}
