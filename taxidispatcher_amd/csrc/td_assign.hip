// td_assign.hip — optimal N x N assignment on gfx950.
//
// Replaces the cvxopt.glpk.ilp call of solver.py:26 / procedure.py:27 / greedy_opt.py:117.
// The assignment polytope is totally unimodular, so the exact min-cost perfect matching
// computed here has the same integer total as GLPK's optimum.
//
// Pipeline (all on one HIP stream, no host round trip between kernels):
//   k_compress   read the int32 matrix ONCE; per row subtract the row minimum (= Jonker-
//                Volgenant row reduction, absorbed into the row dual) and store the row in
//                the narrowest of u8 / u16 / u32 that holds its range, padded to 16-byte
//                chunks.  Later passes stream 1 byte per cell instead of 4.
//   k_bid        Jacobi auction bidding round (Bertsekas): one wavefront per unassigned cab
//                streams its row in 16-byte chunks, keeps per-lane best / second-best
//                (cost + price), reduces with wave64 shuffles and publishes
//                bid = best price + (second - best) with a 64-bit atomicMax on the packed
//                (price, row) key of the column.  eps = 0 ("naive" auction = JV augmenting
//                row reduction): complementary slackness stays EXACT, ties on owned columns
//                are left to the finisher, ties prefer a free column (free bit folded into
//                the key).  Prices are staged through LDS in the wide early rounds.
//   k_assign     resolve winners per column, update price / owner / row_to_col.
//   k_sap        exact finisher for the few rows the bidding rounds leave free: shortest
//                augmenting paths (Dijkstra on reduced costs) run by ONE persistent
//                workgroup with all per-column state in registers / LDS — a step is one row
//                stream + one block-wide argmin, with no kernel launch in the loop.
//   k_final      total from the ORIGINAL int32 costs, permutation check, optional
//                LP-duality bound from the final prices (certificate).
//
// Why not the pure eps-scaling auction: measured with tools/auction_proto.c it needs
// 2 190 (perf.jl instance) to 101 611 (|a-b| instance) bidding rounds at N = 1000, almost all
// of them with 1-2 bidders — pure launch latency on a GPU.  eps = 0 rounds + SAP need <= 16
// rounds and 10^2..10^4 in-kernel steps for the same instances and are exact by construction
// (no (N+1) cost scaling needed).  See DESIGN.md.
#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "td_common.h"

using namespace td;

namespace {

template <typename CT>
struct Tr;
template <>
struct Tr<uint8_t> {
    using PT = int32_t;
    static constexpr int E = 16;
    static constexpr uint32_t SENT = 0xFFu;
    static constexpr int64_t LIMIT = 254;
    static constexpr int32_t BIG = 1 << 28;
    static constexpr int32_t KMAX = INT32_MAX;
};
template <>
struct Tr<uint16_t> {
    using PT = int64_t;
    static constexpr int E = 8;
    static constexpr uint32_t SENT = 0xFFFFu;
    static constexpr int64_t LIMIT = 65534;
    static constexpr int64_t BIG = 1ll << 56;
    static constexpr int64_t KMAX = INT64_MAX;
};
template <>
struct Tr<uint32_t> {
    using PT = int64_t;
    static constexpr int E = 4;
    static constexpr uint32_t SENT = 0xFFFFFFFFu;
    static constexpr int64_t LIMIT = 0xFFFFFFFEll;
    static constexpr int64_t BIG = 1ll << 56;
    static constexpr int64_t KMAX = INT64_MAX;
};

// 32-bit cells with 32-BIT prices, keys and labels ("narrow price" mode, bpc code 5): rows whose
// range needs 4-byte cells but is small (<= NP_RANGE, e.g. the |a-b| geometry: range 163 839 at
// N = 16 384).  Same storage as uint32_t rows; every price / bid key / Dijkstra label is an int32, which
// halves the registers and the ALU work of the finishers (k_sap<u32,4,SPEC> spilled 98 VGPRs with
// 64-bit labels; k_sapx's relax is issue-bound on 64-bit compares).  Safety: prices only ever grow,
// and every kernel that raises a price checks the new value against NP_PLIMIT — a breach sets
// CTL_FLAG bit 3, every later kernel exits, the host redoes the solve with 64-bit prices.  With
// prices < 2^27 and cells < 2^22 no label (price + cell + a path of raises) can wrap an int32.
struct u32n {
    uint32_t v;
    u32n() = default;
    __host__ __device__ constexpr u32n(uint32_t x) : v(x) {}
    __host__ __device__ constexpr operator uint32_t() const { return v; }
};
constexpr int64_t NP_RANGE = (1 << 22) - 2;
constexpr int32_t NP_PLIMIT = 1 << 27;
template <>
struct Tr<u32n> {
    using PT = int32_t;
    static constexpr int E = 4;
    static constexpr uint32_t SENT = 0xFFFFFFFFu;
    static constexpr int64_t LIMIT = NP_RANGE;
    static constexpr int32_t BIG = 1 << 28;
    static constexpr int32_t KMAX = INT32_MAX;
};
template <typename CT>
struct IsNP {
    static constexpr bool value = false;
};
template <>
struct IsNP<u32n> {
    static constexpr bool value = true;
};

// 1-byte cells with ONE escape value per matrix (bpc code 6): padded / thresholded models hold a small range of
// real costs plus the fill value big_cost (Simulator.java:493-520, simulate.py:17-33: {0..9} and 250000).  Codes
// 0..253 are value - base, 254 is the fill value (g_esc8 = fill - base on the device), 255 pads the row.  Prices
// and labels are 32-bit as for plain 1-byte rows; with big_cost in play they are guarded like the narrow-price
// mode (NP_PLIMIT, CTL_FLAG bit 3 -> the host redoes the solve with 4-byte cells).
constexpr uint32_t U8E_ESC = 254u;
__device__ uint32_t g_esc8;   // the escape code's value of the matrix being solved (one solver per process)
struct u8e {
    uint8_t v;
    __device__ __forceinline__ operator uint32_t() const { return v == U8E_ESC ? g_esc8 : (uint32_t)v; }   // single-cell reads
};
template <>
struct Tr<u8e> {
    using PT = int32_t;
    static constexpr int E = 16;
    static constexpr uint32_t SENT = 0xFFu;
    static constexpr int64_t LIMIT = 253;
    static constexpr int32_t BIG = 1 << 28;
    static constexpr int32_t KMAX = INT32_MAX;
};
template <>
struct IsNP<u8e> {
    static constexpr bool value = true;
};
template <typename CT>
struct IsLean8 {   // plain 1-byte rows: the lean finisher k_sap8 and the speculative first attempt
    static constexpr bool value = std::is_same<CT, uint8_t>::value;
};

// control block (int32 words) in device memory
enum {
    CTL_FLAG = 0,      // compress: row range did not fit
    CTL_ERR = 1,       // device-side consistency error
    CTL_NFREE = 2,     // rows handed to SAP
    CTL_STEPS = 3,     // SAP dijkstra steps
    CTL_NCONST = 4,    // constant rows seen by the compress pass (survives k_init_state)
    CTL_PACC = 5,      // augmentations committed by the parallel finisher
    CTL_RANGE = 6,     // [6..7] 64-bit max row range seen by a compress pass that did not fit
    CTL_PROG = 8,      // [CTL_PROG + r] bids applied in round r
    CTL_WORDS = 8 + 64,  // room for up to 48 rounds
    // shape probe (k_shape): [+0] constant-looking columns, [+1] rows, [+2] block ticket,
    // [+3] 1 = solve the transpose, [+4..5] 64-bit largest sampled column range
    CTL_SHAPE = CTL_WORDS,
    CTL_PSTOP = CTL_WORDS + 6,
    CTL_TIED = CTL_WORDS + 7,    // rows whose two smallest costs are equal (bidding round 0, every 16th row sampled)   // a speculative batch committed nothing: later batches of the group exit at once
    CTL_FOREST = CTL_WORDS + 8,  // levels of the incremental forest finisher (0: another finisher ran)
    CTL_RSEEN = CTL_WORDS + 10,  // [+10..11] 64-bit largest row range seen by the row-wise compress passes (whether it fits or not)
    CTL_COREMISS = CTL_WORDS + 12,  // warm start on the sparse core: [+0] bidders whose list did not prove their best column (they bid on the dense row), [+1] bidders seen, [+2..3] 64-bit smallest price of any column
    CTL_ALL = CTL_WORDS + 16
};

constexpr int ROW_BITS = 20;
constexpr int HOP_FMAX = 128;   // td_blocks.h: free rows / columns of a block the two-hop pass looks at
constexpr int HOP_BMAX = 16;    // td_blocks.h: blocks per shard

// tunables (env TD_MAX_ROUNDS / TD_TIE_EVICT / TD_LDS_ROUNDS, read once at td_assign)
// defaults are the measured best on MI355X (tools/gpu_sweep.py); every one has an env override
int g_max_rounds = 12;      // TD_MAX_ROUNDS    bidding rounds launched (converged rounds exit on the device)
int g_tie_evict = 1;        // TD_TIE_EVICT     rounds >= 1: a tie on owned columns still takes the column
int g_lds_rounds = 1;       // TD_LDS_ROUNDS    first rounds that stage the price vector in LDS
int g_lds_grid = 1;         // TD_LDS_GRID      workgroups per CU of the LDS-staged bidding kernel
int g_row_loop = 5;         // TD_ROW_LOOP      from this round on k_bid_row loops over the rows with a fixed grid
int g_row_rounds = 2;       // TD_ROW_ROUNDS    from this round on: one workgroup per row (k_bid_row)
int g_cgrid = 4;            // TD_CGRID         workgroups per CU of the compress pass
int g_creg = 1;             // TD_CREG          register-resident compress kernel
int g_speculate = 1;        // TD_SPECULATE     try u8 storage without waiting for the range flag
int g_sap8 = 1;             // TD_SAP8          lean u8 finisher
int g_psap8_batches = 1;    // TD_PSAP8         speculative batches of the lean u8 search
int g_psap8_grid = 64;      // TD_PSAP8_GRID    searches per such batch
int g_warm_tie_div = 2;     // TD_WARM_TIE_DIV  no warm start when more than n / this rows are tied at their minimum
int g_u16_redo_free = 64;    // TD_U16_REDO_FREE 2-byte rows that leave at least this many free rows after the eps = 0 rounds are redone as 4-byte cells (forest finisher) even when too many rows are tied for the warm start (0: off); uniform -5000..4999 at n = 16 384 (80 % of the rows tied at their minimum): 503 -> 35 ms
int g_wide_u16_n = 2048;    // TD_WIDE_U16_N    wide, tie-free 2-byte rows of n >= this are redone as 4-byte cells with 32-bit prices (0: never); 4096 until the forest finisher took n >= 2048 (2-D Manhattan n = 2500 / 3500: 31 -> 21 ms, 46 -> 21 ms; uniform 0..40 000 n = 3000: 6.6 -> 7.1 ms)
bool g_line = true;         // TD_LINE          0: skip the line-metric recogniser (td_line.hip), always run the general solver
int g_line_min_n = 2;       // TD_LINE_MIN_N    smallest n the recogniser is tried on
int g_psap_batches = 16;    // TD_PSAP          speculative batches per group of the generic search (u16 / u32 rows)
int g_psap_min = 12;        // TD_PSAP_MIN      free rows below which the generic batches are skipped
int g_psap_cap = 4096;      // TD_PSAP_CAP      total speculative batches per solve
int g_warm = 1;             // TD_WARM          eps > 0 auction phases as a price warm start for wide, tie-free rows
int g_warm_div = 4;         // TD_WARM_DIV      first eps = row range / this
int g_warm_theta = 2;       // TD_WARM_THETA    eps divided by this between phases
int g_warm_bits = 30;       // TD_WARM_BITS     last eps >= row range >> this
int g_warm_groups = 32;     // TD_WARM_GROUPS   groups of 8 rounds per phase at most
int g_warm_cut = 64;        // TD_WARM_CUT      a phase ends when <= n / this rows are free (0: none)
int g_warm_min_range = 256; // TD_WARM_MIN_RANGE rows narrower than this are never warmed
int g_warm_minfree = 32;    // TD_WARM_MINFREE  free rows after the eps = 0 rounds below which the finisher is cheaper
int g_warm_min_n = 512;     // TD_WARM_MIN_N    no warm start for smaller models
int g_sapx = 1;             // TD_SAPX          cooperative multi-workgroup serial finisher (k_sapx)
int g_sapx_min = 8;         // TD_SAPX_MIN      fewest workgroups (256 chunks each) for which it is used
int g_sapx_slim_chunks = INT_MAX; // TD_SAPX_SLIM  1- / 2-byte rows with at least this many 16-byte chunks: 64-thread column slices in k_sapx.  OFF: measured at n = 65 536 (perf.jl rows, 7 searches, 14 steps) 64 slim workgroups take 1.41 ms against 0.71 ms for 16 wide ones — the barrier and the publish phase grow with the workgroup count faster than the relax shrinks
int g_sapx_rows = 24;       // TD_SAPX_ROWS     ... half of that when at least this many rows are left for it
int g_sap512 = 1;           // TD_SAP512        512-thread generic finisher (double register budget) for n <= 8192
int g_psap_worth = 4;       // TD_PSAP_WORTH    rows a batch must commit on average for another group to be launched
int g_bid0 = 1;             // TD_BID0          round 0's bids come out of the register-resident compress pass (td_assign, 1-byte attempt; k_compress_reg<.., BID0>): perf.jl n = 16 384 step 0.870 -> 0.809 ms
int g_defer_const = 1;      // TD_DEFER_CONST   constant rows sit out the solve (k_place_const)
int g_shape = 1;            // TD_SHAPE         probe for constant columns and solve the transpose when they dominate
int g_shape_max_n = 1 << 20; // TD_SHAPE_MAX_N   largest n the probe runs for
int g_narrow_price = 1;     // TD_NARROW_PRICE  4-byte cells with a row range <= 2^22: 32-bit prices and labels first (redone in 64 bits if a price reaches 2^27)
long long g_np_plimit = NP_PLIMIT;   // TD_NP_PLIMIT  (tests) lower price limit of the narrow-price mode in k_assign / k_pcommit
int g_fuse_t = 2;           // TD_FUSE_T        padded models (dummy requests): one fused transpose + compress pass, no void 1-byte attempt (2: as 1-byte cells + escape when they fit, 1: 4-byte cells, 0: off)
int g_fuse_spec = 1;        // TD_FUSE_SPEC     the fused pass is speculative (flags read with the final result) when the probe saw a plausible fill value
int g_fused_rounds = 8;      // TD_FUSED_ROUNDS  bidding rounds launched for a padded model taken by the fused pass
int g_fuse_gen = 1;          // TD_FUSE_GEN      td_build_assign / td_tick: cells of a padded model made from the position arrays inside the fused pass (no int32 matrix)
int g_tick_rounds = 5;       // TD_TICK_ROUNDS   ... of td_tick's remainder (n < 2048): 5 rounds make progress on the bench tick, every further launch is ~4 us of nothing
int g_forest = 1;           // TD_FOREST        cooperative incremental shortest-path forest (k_forest) as the finisher of 4-byte rows
int g_forest_min_n = 2048;  // TD_FOREST_MIN_N  smallest n it is used for
long long g_forest_w0 = 16; // TD_FOREST_W0     first label window
long long g_forest_wx = 16; // TD_FOREST_WX     how far above the smallest free-column label a window may reach
int g_blocks = -1;          // TD_BLOCKS        block-local start of the 1-byte attempt (td_blocks.h): diagonal blocks of the matrix; 0: off, -1: by size (td_assign: off below TD_BLOCKS_MIN_N)
int g_blocks_min_n = 12288; // TD_BLOCKS_MIN_N  smallest n td_assign starts block-locally by itself (perf.jl solve, n = 12 288 / 16 384 / 32 768 / 65 536: 0.61 / 0.76 / 2.26 / 7.5 -> 0.41 / 0.52 / 1.69 / 6.2 ms)
int g_zs_rounds = 4;        // TD_ZS_ROUNDS     local bidding rounds of phase A after round 0
int g_zs_sep = 0;           // TD_ZS_SEP        rows of <= 16 384 columns: round 0 of phase A as its own launch behind the plain compress pass.  OFF: measured slower (td_assign n = 16 384 0.487 -> 0.543 ms: the plain 256 x 16 pass takes 0.29 ms in this sequence, the pass that also writes the bids 0.243)
int g_hop_passes = 2;       // TD_HOP_PASSES    two-hop passes at the end of phase A (the second one takes the rows the first pass's greedy left: 4 of 183 at n = 16 384)
int g_hop_max_rows = HOP_FMAX;   // TD_HOP_MAX_ROWS  a block with more free rows than this is left to the rounds
int g_lazy_cc = 1;          // TD_LAZY_CC       td_assign, block-local start: the compress pass stores the diagonal slices of the narrow copy only; the rest is written (k_compress_rest) only if phase A + the two-hop pass over the whole matrix leave rows
int g_hop_global = 1;       // TD_HOP_GLOBAL    td_assign: one two-hop pass over the whole matrix after phase A
int g_zs_global_rounds = 6; // TD_ZS_GLOBAL_ROUNDS  td_assign: bidding rounds launched for what the block-local start left
int g_core = 1;             // TD_CORE          the warm start's eps-phases bid on a sparse core of every row (td_core_warm.h)
int g_core_k = 64;          // TD_CORE_K        rank of the group minimum that becomes a row's threshold (about 256 * -ln(1 - k/256) cells per row: 74)
int g_core_min_n = 8192;    // TD_CORE_MIN_N    smallest n it is used for (uniform 0..10^6: n = 4096 9.1 -> 11.4 ms, the dense rounds of 16 KiB rows are cheaper than building the lists; n = 16 384 43.7 -> 32.3 ms)
int g_core_mode = 1;        // TD_CORE_MODE     1: the lists are taken on trust in the phases whose eps is at most TD_CORE_EPS x their average reach (the earlier phases bid on the dense rows); 0: every phase, a bid only where the list proves the row's best column, the other bidders on their dense rows
double g_core_eps = 1.0;    // TD_CORE_EPS
int g_rand_test = 1;        // TD_RAND_TEST     row-correlation test (k_row_corr): random-like matrices get the short eps ladder (TD_RAND_DIV) and TD_RAND_ROUNDS eps = 0 rounds after it
int g_rand_div = 8192;      // TD_RAND_DIV
int g_rand_rounds = 48;     // TD_RAND_ROUNDS
int g_rand_repeat = 1;      // TD_RAND_REPEAT   blocks of TD_RAND_ROUNDS eps = 0 rounds after the warm start
double g_rand_corr = 0.12;  // TD_RAND_CORR     mean |r| over 64 sampled row pairs below which the matrix counts as random-like (random: ~0.05; metric structure: 0.3 - 0.9)
int g_core_patience = 6;    // TD_CORE_PATIENCE groups of 8 rounds a phase on trusted lists may take before the lists are dropped for good
int g_solver_eps = 0;       // TD_SOLVER=eps    literal eps-scaling auction (comparison mode)
int g_eps_theta = 8;        // TD_EPS_THETA
long long g_eps0_mult = 4;  // TD_EPS0_MULT     eps0 = (n+1) * mult ; 0 = start at eps = 1
void read_tunables()
{
    static bool done = false;
    if (done) return;
    done = true;
    if (const char *e = getenv("TD_MAX_ROUNDS")) g_max_rounds = std::max(1, std::min(48, atoi(e)));
    if (const char *e = getenv("TD_TIE_EVICT")) g_tie_evict = atoi(e) != 0;
    if (const char *e = getenv("TD_LDS_ROUNDS")) g_lds_rounds = std::max(0, atoi(e));
    if (const char *e = getenv("TD_SAP8")) g_sap8 = atoi(e) != 0;
    if (const char *e = getenv("TD_ROW_LOOP")) g_row_loop = std::max(0, atoi(e));
    if (const char *e = getenv("TD_ROW_ROUNDS")) g_row_rounds = std::max(0, atoi(e));
    if (const char *e = getenv("TD_CGRID")) g_cgrid = std::max(1, atoi(e));
    if (const char *e = getenv("TD_CREG")) g_creg = atoi(e) != 0;
    if (const char *e = getenv("TD_PSAP")) g_psap_batches = std::max(0, std::min(64, atoi(e)));
    if (const char *e = getenv("TD_PSAP_MIN")) g_psap_min = std::max(1, atoi(e));
    if (const char *e = getenv("TD_PSAP_CAP")) g_psap_cap = std::max(0, atoi(e));
    if (const char *e = getenv("TD_SAP512")) g_sap512 = atoi(e) != 0;
    if (const char *e = getenv("TD_SAPX")) g_sapx = atoi(e) != 0;
    if (const char *e = getenv("TD_WARM")) g_warm = atoi(e) != 0;
    if (const char *e = getenv("TD_WARM_DIV")) g_warm_div = std::max(1, atoi(e));
    if (const char *e = getenv("TD_WARM_THETA")) g_warm_theta = std::max(2, atoi(e));
    if (const char *e = getenv("TD_WARM_BITS")) g_warm_bits = std::max(1, std::min(30, atoi(e)));
    if (const char *e = getenv("TD_WARM_GROUPS")) g_warm_groups = std::max(1, atoi(e));
    if (const char *e = getenv("TD_WARM_CUT")) g_warm_cut = std::max(0, atoi(e));
    if (const char *e = getenv("TD_WARM_MIN_RANGE")) g_warm_min_range = std::max(1, atoi(e));
    if (const char *e = getenv("TD_WARM_MINFREE")) g_warm_minfree = std::max(1, atoi(e));
    if (const char *e = getenv("TD_WARM_MIN_N")) g_warm_min_n = std::max(0, atoi(e));
    if (const char *e = getenv("TD_SAPX_MIN")) g_sapx_min = std::max(1, atoi(e));
    if (const char *e = getenv("TD_SAPX_SLIM")) g_sapx_slim_chunks = std::max(64, atoi(e));
    if (const char *e = getenv("TD_SAPX_ROWS")) g_sapx_rows = std::max(1, atoi(e));
    if (const char *e = getenv("TD_PSAP_WORTH")) g_psap_worth = std::max(1, atoi(e));
    if (const char *e = getenv("TD_SPECULATE")) g_speculate = atoi(e) != 0;
    if (const char *e = getenv("TD_LDS_GRID")) g_lds_grid = std::max(1, std::min(8, atoi(e)));
    if (const char *e = getenv("TD_NARROW_PRICE")) g_narrow_price = atoi(e) != 0;
    if (const char *e = getenv("TD_NP_PLIMIT")) g_np_plimit = std::max(1ll, std::min((long long)NP_PLIMIT, atoll(e)));
    if (const char *e = getenv("TD_FUSE_T")) g_fuse_t = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("TD_FUSE_SPEC")) g_fuse_spec = atoi(e) != 0;
    if (const char *e = getenv("TD_FUSED_ROUNDS")) g_fused_rounds = std::max(1, std::min(48, atoi(e)));
    if (const char *e = getenv("TD_FUSE_GEN")) g_fuse_gen = atoi(e) != 0;
    if (const char *e = getenv("TD_TICK_ROUNDS")) g_tick_rounds = std::max(1, std::min(48, atoi(e)));
    if (const char *e = getenv("TD_FOREST")) g_forest = atoi(e) != 0;
    if (const char *e = getenv("TD_FOREST_MIN_N")) g_forest_min_n = std::max(64, atoi(e));
    if (const char *e = getenv("TD_FOREST_W0")) g_forest_w0 = std::max(1ll, atoll(e));
    if (const char *e = getenv("TD_FOREST_WX")) g_forest_wx = std::max(0ll, atoll(e));
    if (const char *e = getenv("TD_SOLVER")) g_solver_eps = (strcmp(e, "eps") == 0);
    if (const char *e = getenv("TD_EPS0_MULT")) g_eps0_mult = std::max(0ll, atoll(e));
    if (const char *e = getenv("TD_EPS_THETA")) g_eps_theta = std::max(2, atoi(e));
    if (const char *e = getenv("TD_PSAP8")) g_psap8_batches = std::max(0, std::min(32, atoi(e)));
    if (const char *e = getenv("TD_DEFER_CONST")) g_defer_const = atoi(e) != 0;
    if (const char *e = getenv("TD_BID0")) g_bid0 = atoi(e) != 0;
    if (const char *e = getenv("TD_SHAPE")) g_shape = atoi(e) != 0;
    if (const char *e = getenv("TD_SHAPE_MAX_N")) g_shape_max_n = atoi(e);
    if (const char *e = getenv("TD_WIDE_U16_N")) g_wide_u16_n = std::max(0, atoi(e));
    if (const char *e = getenv("TD_U16_REDO_FREE")) g_u16_redo_free = std::max(0, atoi(e));
    if (const char *e = getenv("TD_WARM_TIE_DIV")) g_warm_tie_div = std::max(1, atoi(e));
    if (const char *e = getenv("TD_LINE")) g_line = atoi(e) != 0;
    if (const char *e = getenv("TD_LINE_MIN_N")) g_line_min_n = std::max(2, atoi(e));
    if (const char *e = getenv("TD_PSAP8_GRID")) g_psap8_grid = std::max(1, std::min(192, atoi(e)));
    if (const char *e = getenv("TD_CORE")) g_core = atoi(e) != 0;
    if (const char *e = getenv("TD_CORE_K")) g_core_k = std::max(2, std::min(200, atoi(e)));
    if (const char *e = getenv("TD_CORE_MIN_N")) g_core_min_n = std::max(256, atoi(e));
    if (const char *e = getenv("TD_CORE_MODE")) g_core_mode = atoi(e);
    if (const char *e = getenv("TD_CORE_EPS")) g_core_eps = atof(e);
    if (const char *e = getenv("TD_CORE_PATIENCE")) g_core_patience = std::max(1, atoi(e));
    if (const char *e = getenv("TD_RAND_TEST")) g_rand_test = atoi(e) != 0;
    if (const char *e = getenv("TD_RAND_DIV")) g_rand_div = std::max(1, atoi(e));
    if (const char *e = getenv("TD_RAND_ROUNDS")) g_rand_rounds = std::max(1, std::min(48, atoi(e)));
    if (const char *e = getenv("TD_RAND_CORR")) g_rand_corr = atof(e);
    if (const char *e = getenv("TD_RAND_REPEAT")) g_rand_repeat = std::max(1, atoi(e));
    if (const char *e = getenv("TD_BLOCKS")) g_blocks = std::max(-1, std::min(HOP_BMAX, atoi(e)));
    if (const char *e = getenv("TD_BLOCKS_MIN_N")) g_blocks_min_n = std::max(0, atoi(e));
    if (const char *e = getenv("TD_ZS_ROUNDS")) g_zs_rounds = std::max(0, std::min(32, atoi(e)));
    if (const char *e = getenv("TD_ZS_SEP")) g_zs_sep = atoi(e) != 0;
    if (const char *e = getenv("TD_HOP_PASSES")) g_hop_passes = std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("TD_HOP_MAX_ROWS")) g_hop_max_rows = std::max(1, atoi(e));
    if (const char *e = getenv("TD_HOP_GLOBAL")) g_hop_global = atoi(e) != 0;
    if (const char *e = getenv("TD_LAZY_CC")) g_lazy_cc = atoi(e) != 0;
    if (const char *e = getenv("TD_ZS_GLOBAL_ROUNDS")) g_zs_global_rounds = std::max(1, std::min(48, atoi(e)));
}

// ---- unpack one 16-byte chunk into E cost values -----------------------------------
template <typename CT>
__device__ __forceinline__ void unpack(const uint4 &v, uint32_t *out);
template <>
__device__ __forceinline__ void unpack<uint8_t>(const uint4 &v, uint32_t *o)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o[4 * k + 0] = w[k] & 0xFFu;
        o[4 * k + 1] = (w[k] >> 8) & 0xFFu;
        o[4 * k + 2] = (w[k] >> 16) & 0xFFu;
        o[4 * k + 3] = w[k] >> 24;
    }
}
template <>
__device__ __forceinline__ void unpack<u8e>(const uint4 &v, uint32_t *o)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const uint32_t esc = g_esc8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t x = (w[k] >> (8 * b)) & 0xFFu;
            o[4 * k + b] = (x == U8E_ESC) ? esc : x;
        }
    }
}
template <>
__device__ __forceinline__ void unpack<uint16_t>(const uint4 &v, uint32_t *o)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o[2 * k + 0] = w[k] & 0xFFFFu;
        o[2 * k + 1] = w[k] >> 16;
    }
}
template <>
__device__ __forceinline__ void unpack<uint32_t>(const uint4 &v, uint32_t *o)
{
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
    o[3] = v.w;
}
template <>
__device__ __forceinline__ void unpack<u32n>(const uint4 &v, uint32_t *o)
{
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
    o[3] = v.w;
}

// 16-byte streaming load: TD_NT bit 2 makes the bidding kernels' row stream nontemporal
__device__ __forceinline__ uint4 load16_stream(const void *p)
{
#if defined(TD_NT) && (TD_NT & 4)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int m)
{
    return __shfl_xor(v, m);
}

// =====================================================================================
// k_compress: one workgroup per row (grid-strided)
// =====================================================================================
template <typename CT, bool VEC>
__global__ __launch_bounds__(256) void k_compress(int n, int nrows, int nchunks, const int32_t *__restrict__ cost,
                                                  CT *__restrict__ cc, int32_t *__restrict__ rowmin,
                                                  int *__restrict__ ctl, int *__restrict__ rconst, const long long *__restrict__ skip)
{
    if (skip && *skip) return;   // the line-metric probe queued in front of this pass wants the matrix for itself (td_line.hip)
    constexpr int E = Tr<CT>::E;
    __shared__ int s_mn[4], s_mx[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t pitch = (size_t)nchunks * E;
    long long wmax = 0;   // largest row range of this workgroup's rows (thread 0)
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int32_t *src = cost + (int64_t)row * n;
        int mn = INT_MAX, mx = INT_MIN;
        if (VEC) {
            const int4 *s4 = reinterpret_cast<const int4 *>(src);
            for (int q = tid; q < (n >> 2); q += 256) {
                int4 v = s4[q];
                mn = min(min(mn, v.x), min(v.y, min(v.z, v.w)));
                mx = max(max(mx, v.x), max(v.y, max(v.z, v.w)));
            }
        } else {
            for (int j = tid; j < n; j += 256) {
                int v = src[j];
                mn = min(mn, v);
                mx = max(mx, v);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, __shfl_xor(mn, o));
            mx = max(mx, __shfl_xor(mx, o));
        }
        if (lane == 0) {
            s_mn[w] = mn;
            s_mx[w] = mx;
        }
        __syncthreads();
        mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        __syncthreads();
        if (tid == 0) {
            rowmin[row] = mn;
            rconst[row] = (mx == mn) ? 1 : 0;   // a constant row can take ANY column at the same cost
            if (mx == mn) atomicAdd(&ctl[CTL_NCONST], 1);
            wmax = std::max(wmax, (long long)mx - (long long)mn);
            if ((int64_t)mx - (int64_t)mn > Tr<CT>::LIMIT) {
                atomicOr(&ctl[CTL_FLAG], 1);
                atomicMax(reinterpret_cast<unsigned long long *>(&ctl[CTL_RANGE]), (unsigned long long)((int64_t)mx - (int64_t)mn));
            }
        }
        // second read of the row comes from L2 (the workgroup has just streamed it)
        CT *dst = cc + (size_t)row * pitch;
        for (int k = tid; k < nchunks; k += 256) {
            uint32_t o[E];
            const int j0 = k * E;
            if (VEC && j0 + E <= n) {
                const int4 *s4 = reinterpret_cast<const int4 *>(src + j0);
#pragma unroll
                for (int q = 0; q < E / 4; q++) {
                    int4 v = s4[q];
                    o[4 * q + 0] = (uint32_t)(v.x - mn);
                    o[4 * q + 1] = (uint32_t)(v.y - mn);
                    o[4 * q + 2] = (uint32_t)(v.z - mn);
                    o[4 * q + 3] = (uint32_t)(v.w - mn);
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) o[e] = (j0 + e < n) ? (uint32_t)(src[j0 + e] - mn) : Tr<CT>::SENT;
            }
            uint4 pk;
            if constexpr (sizeof(CT) == 1) {
                pk.x = (o[0] & 0xFF) | ((o[1] & 0xFF) << 8) | ((o[2] & 0xFF) << 16) | (o[3] << 24);
                pk.y = (o[4] & 0xFF) | ((o[5] & 0xFF) << 8) | ((o[6] & 0xFF) << 16) | (o[7] << 24);
                pk.z = (o[8] & 0xFF) | ((o[9] & 0xFF) << 8) | ((o[10] & 0xFF) << 16) | (o[11] << 24);
                pk.w = (o[12] & 0xFF) | ((o[13] & 0xFF) << 8) | ((o[14] & 0xFF) << 16) | (o[15] << 24);
            } else if constexpr (sizeof(CT) == 2) {
                pk.x = (o[0] & 0xFFFF) | (o[1] << 16);
                pk.y = (o[2] & 0xFFFF) | (o[3] << 16);
                pk.z = (o[4] & 0xFFFF) | (o[5] << 16);
                pk.w = (o[6] & 0xFFFF) | (o[7] << 16);
            } else {
                pk.x = o[0];
                pk.y = o[1];
                pk.z = o[2];
                pk.w = o[3];
            }
            reinterpret_cast<uint4 *>(dst)[k] = pk;
        }
    }
    if (tid == 0 && wmax > 0) atomicMax(reinterpret_cast<unsigned long long *>(&ctl[CTL_RSEEN]), (unsigned long long)wmax);
}

// The cells k_compress_reg<.., BID0>(diag_only = 1) did not store: every row's narrow cells OUTSIDE its own column slice,
// from the row minima that pass left (same bytes as the full pass writes).  Launched only when something after phase A
// needs whole rows of the narrow copy (bidding rounds over whole rows, a finisher, the dual bound).
__global__ __launch_bounds__(256) void k_compress_rest(int n, int nrows, int row0, int nchunks, int rpb, const int32_t *__restrict__ cost,
                                                       uint8_t *__restrict__ cc, const int32_t *__restrict__ rowmin,
                                                       const int *__restrict__ ctl)
{
    if (ctl[CTL_FLAG]) return;   // rows too wide for one byte: the attempt is abandoned
    const int nq = n >> 2, qpb = rpb >> 2;
    const size_t pitch = (size_t)nchunks * 16;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int mn = rowmin[row];
        const int q0 = ((row0 + row) / rpb) * qpb;
        const int4 *s4 = reinterpret_cast<const int4 *>(cost + (int64_t)row * n);
        uint32_t *dst = reinterpret_cast<uint32_t *>(cc + (size_t)row * pitch);
        for (int q = threadIdx.x; q < nq; q += 256) {
            if ((unsigned)(q - q0) < (unsigned)qpb) continue;
            const int4 x4 = s4[q];
            const uint32_t a = (uint32_t)(x4.x - mn), b = (uint32_t)(x4.y - mn), c = (uint32_t)(x4.z - mn), d = (uint32_t)(x4.w - mn);
            dst[q] = (a & 0xFF) | ((b & 0xFF) << 8) | ((c & 0xFF) << 16) | (d << 24);
        }
    }
}

// Register-resident variant: the row is read ONCE from HBM (all VPT 16-byte loads of a thread
// are issued back to back, so a 256-thread workgroup keeps 64 KiB in flight), reduced, and
// written back narrow.  Needs n % 4 == 0, a 16-byte aligned matrix and n/4 <= THREADS*VPT.
__device__ __forceinline__ void shape_decide(int n, int *__restrict__ ctl);

// BID0: the row is in registers with its minimum known, so the bid of round 0 (all prices 0: the first minimum in
// the row's rotated chunk order, raised by second smallest - smallest) is written here and the round's own pass over
// the narrow matrix (k_bid, round 0) is not launched — the same key, bit for bit.
// LH > 0: the LAST LH of a thread's VPT 16-byte pieces wait in LDS instead of registers between the load and the store
// phase (65 536-column rows: 1024 threads x 16 pieces hit the 128-VGPR cap of a 16-wave workgroup and spilled 200 bytes per
// lane — scratch traffic of the order of the row itself; 8 pieces in registers + 8 x 16 KiB in LDS do not spill).
template <typename CT, int VPT, int THREADS, bool BID0 = false, int LH = 0>
__global__ __launch_bounds__(THREADS) void k_compress_reg(
    int n, int nrows, int nchunks, const int32_t *__restrict__ cost, CT *__restrict__ cc, int32_t *__restrict__ rowmin, int *__restrict__ ctl,
    int *__restrict__ rconst, const long long *__restrict__ skip, unsigned long long *__restrict__ bid = nullptr, int row0 = 0,
    int *__restrict__ r2c = nullptr /* BID0: constant rows are deferred (-2) */,
    int probe_tickets = 0 /* BID0: > 0 = take a ticket of the shape probe at the end */,
    int zs_rpb = 0 /* BID0: > 0 = block-local start (td_blocks.h): rows per diagonal block; a row bids for the first ZERO cell of
                      its own column slice only, never raising a price */,
    int diag_only = 0 /* BID0 + zs_rpb: 1 = store the narrow cells of the row's OWN column slice only (1/V of the copy: all that
                         phase A reads); k_compress_rest writes the rest if anything after phase A needs whole rows */)
{
    static_assert(!BID0 || sizeof(CT) == 1, "round 0 out of the compress pass: 1-byte cells");
    if (skip && *skip) return;
    constexpr int E = Tr<CT>::E;
    constexpr int NW = THREADS / 64;
    __shared__ int s_mn[2][NW], s_mx[2][NW];
    __shared__ int s_fp[2][BID0 ? NW : 1], s_c0[2][BID0 ? NW : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t pitch = (size_t)nchunks * E;
    const int nq = n >> 2;
    int par = 0;
    // BID0: the bid of a row is issued one iteration LATER, behind the next row's barrier (the waves' summaries of the
    // zero bytes they stored travel through LDS): no second barrier per row, which would also wait for the row's stores
    int prev_row = -1, prev_rot = 0;
    long long wmax = 0;   // largest row range of this workgroup's rows (thread 0)
    auto flush_bid = [&](int slot) {   // wave 0, after a barrier that follows the summaries of `prev_row` in s_fp / s_c0 [slot]
        if constexpr (BID0) {
            if (prev_row < 0 || w != 0) return;
            int fp = INT_MAX, c0 = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                fp = min(fp, s_fp[slot][k]);
                c0 += s_c0[slot][k];
            }
            if (fp == INT_MAX) return;   // the row did not bid (deferred, or too wide for one byte)
            unsigned m2 = 255;
            if (c0 == 1 && !zs_rpb) {   // (wave-uniform, rare in rows this wide) a single cell at the minimum: second smallest = the smallest non-zero byte of the row just written
                // (agent-scope loads: the words were stored by the other waves of this workgroup before the barrier, their
                // stores acknowledged by L2; the CU's L1 is not trusted to have seen them)
                const uint32_t *rp = reinterpret_cast<const uint32_t *>(cc + (size_t)prev_row * pitch);
                for (int q = lane; q < nchunks * 4; q += 64) {
                    const uint32_t x = __hip_atomic_load(rp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const unsigned b = (x >> (8 * e)) & 0xFFu;
                        m2 = (b != 0 && b < m2) ? b : m2;   // (pad cells hold the sentinel 255)
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m2 = min(m2, (unsigned)__shfl_xor((int)m2, o));
            }
            if (lane == 0) {
                const int grow = row0 + prev_row;
                const int t1 = fp / E;
                int ch = t1 + prev_rot;
                if (zs_rpb) {   // the rotation runs inside the row's column slice
                    const int zcpb = zs_rpb / E;
                    if (ch >= zcpb) ch -= zcpb;
                    ch += (grow / zs_rpb) * zcpb;
                } else if (ch >= nchunks)
                    ch -= nchunks;
                const int j1 = ch * E + (fp - t1 * E);
                const unsigned long long inc = (c0 >= 2 || m2 == 255) ? 0ull : (unsigned long long)m2;
                if (inc == 0 && (grow & 15) == 0) atomicAdd(&ctl[CTL_TIED], 1);
                if (j1 < n) atomicMax(&bid[j1], (inc << ROW_BITS) | (unsigned long long)(grow + 1));
            }
        }
    };
    constexpr int VR = VPT - LH;   // pieces that stay in registers
    extern __shared__ int4 s_rowh[];   // [LH][THREADS]: every thread reads back what it wrote itself
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int4 *s4 = reinterpret_cast<const int4 *>(cost + (int64_t)row * n);
        int4 v[VPT];
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int q = k * THREADS + tid;
#if defined(TD_NT) && (TD_NT & 2)
            if (q < nq) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i t4 = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(s4) + q);
                v[k] = make_int4(t4.x, t4.y, t4.z, t4.w);
            }
#else
            if (q < nq) v[k] = s4[q];
#endif
        }
        int mn = INT_MAX, mx = INT_MIN;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int q = k * THREADS + tid;
            if (q < nq) {
                mn = min(min(mn, v[k].x), min(v[k].y, min(v[k].z, v[k].w)));
                mx = max(max(mx, v[k].x), max(v[k].y, max(v[k].z, v[k].w)));
            }
        }
        if constexpr (LH > 0) {   // park the upper pieces in LDS: their registers are free for the scan below
#pragma unroll
            for (int k = VR; k < VPT; k++)
                if (k * THREADS + tid < nq) s_rowh[(k - VR) * THREADS + tid] = v[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, __shfl_xor(mn, o));
            mx = max(mx, __shfl_xor(mx, o));
        }
        if (lane == 0) {
            s_mn[par][w] = mn;
            s_mx[par][w] = mx;
        }
        __syncthreads();
        flush_bid(par ^ 1);   // the previous row's summaries were written before this barrier
#pragma unroll
        for (int k = 0; k < NW; k++) {
            mn = min(mn, s_mn[par][k]);
            mx = max(mx, s_mx[par][k]);
        }
        const bool fits = (int64_t)mx - (int64_t)mn <= Tr<CT>::LIMIT;
        if (tid == 0) {
            rowmin[row] = mn;
            rconst[row] = (mx == mn) ? 1 : 0;   // a constant row can take ANY column at the same cost
            if (mx == mn) atomicAdd(&ctl[CTL_NCONST], 1);
            wmax = std::max(wmax, (long long)mx - (long long)mn);
            if (!fits) {
                atomicOr(&ctl[CTL_FLAG], 1);
                atomicMax(reinterpret_cast<unsigned long long *>(&ctl[CTL_RANGE]), (unsigned long long)((int64_t)mx - (int64_t)mn));
            }
        }
        // BID0 (row-uniform): the narrow words are scanned for zero bytes (= cells at the row minimum) as they are stored
        const bool deferred = BID0 && r2c && mx == mn;
        const bool bids = BID0 && fits && !deferred;
        int fp = INT_MAX, c0 = 0, rot = 0, zc0 = 0;
        const int zcpb = BID0 ? zs_rpb / E : 0;
        if constexpr (BID0) {
            const uint32_t hsh = ((uint32_t)(row0 + row) + 1u) * 0x9E3779B1u;   // k_bid's rotation, round 0
            rot = (int)(((uint64_t)(hsh ^ (hsh >> 15)) * (uint64_t)(zs_rpb ? zcpb : nchunks)) >> 32);
            if (zs_rpb) zc0 = ((row0 + row) / zs_rpb) * zcpb;
        }
        CT *dst = cc + (size_t)row * pitch;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int q = k * THREADS + tid;
            if (q < nq) {
                int4 x4;
                if constexpr (LH > 0) {
                    if (k >= VR) x4 = s_rowh[(k - VR) * THREADS + tid];
                    else x4 = v[k];
                } else
                    x4 = v[k];
                const uint32_t a = (uint32_t)(x4.x - mn), b = (uint32_t)(x4.y - mn), c = (uint32_t)(x4.z - mn),
                               d = (uint32_t)(x4.w - mn);
                if constexpr (sizeof(CT) == 1) {
                    const uint32_t word = (a & 0xFF) | ((b & 0xFF) << 8) | ((c & 0xFF) << 16) | (d << 24);
                    if constexpr (BID0) {
                        const bool in_slice = !zs_rpb || (unsigned)((q >> 2) - zc0) < (unsigned)zcpb;
                        if (!diag_only || in_slice) reinterpret_cast<uint32_t *>(dst)[q] = word;
                        // 0x80 in every byte of `word` that is zero (exact: no borrow between bytes)
                        // (branch-free: a divergent branch here makes the compiler wait for every store before the next)
                        const uint32_t z = ~(((word & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | word | 0x7F7F7F7Fu);
                        int t = (q >> 2) - zc0 - rot;   // position of the word's chunk in the row's rotated order
                        t += t < 0 ? (zs_rpb ? zcpb : nchunks) : 0;
                        const int cand = t * 16 + (q & 3) * 4 + (__builtin_ctz(z | 0x80000000u) >> 3);
                        fp = min(fp, (z && in_slice) ? cand : INT_MAX);
                        c0 += __builtin_popcount(z);
                    } else
                        reinterpret_cast<uint32_t *>(dst)[q] = word;
                } else if constexpr (sizeof(CT) == 2) {
                    reinterpret_cast<uint2 *>(dst)[q] = make_uint2((a & 0xFFFF) | (b << 16), (c & 0xFFFF) | (d << 16));
                } else {
                    reinterpret_cast<uint4 *>(dst)[q] = make_uint4(a, b, c, d);
                }
            }
        }
        // sentinel tail up to the 16-byte chunk boundary
        for (int j = n + tid; j < (int)pitch; j += THREADS) dst[j] = (CT)Tr<CT>::SENT;
        if constexpr (BID0) {
            if (deferred && tid == 0) r2c[row] = -2;
            if (!bids) fp = INT_MAX, c0 = 0;   // (row-uniform)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                fp = min(fp, __shfl_xor(fp, o));
                c0 += __shfl_xor(c0, o);
            }
            if (lane == 0) {
                s_fp[par][w] = fp;
                s_c0[par][w] = c0;
            }
            prev_row = row;
            prev_rot = rot;
        }
        par ^= 1;
    }
    if (tid == 0 && wmax > 0) atomicMax(reinterpret_cast<unsigned long long *>(&ctl[CTL_RSEEN]), (unsigned long long)wmax);
    if constexpr (BID0) {
        __syncthreads();   // (waits for the last row's stores too: the single-minimum path reads them back)
        flush_bid(par ^ 1);
        if (probe_tickets > 0) {   // the shape probe's column samples were taken by k_init_state, in front of this pass
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                if (atomicAdd(&ctl[CTL_SHAPE + 2], 1) == probe_tickets - 1) shape_decide(n, ctl);
            }
        }
    }
}

// =====================================================================================
// state init
// =====================================================================================
__device__ __forceinline__ void shape_probe(int n, const int32_t *__restrict__ cost, int *__restrict__ ctl, int tickets);

template <typename PT>
__global__ __launch_bounds__(256) void k_init_state(int n, int npad, int nrows, PT *pk, PT padkey, int *owner, int *r2c,
                                                    unsigned long long *bid, int *ctl, const int *rconst,
                                                    const int32_t *probe_cost /* non-null: run the shape probe */,
                                                    int probe_tickets = 0 /* > 0: the compress pass that FOLLOWS takes tickets too and decides */)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < npad) {
        pk[j] = (j < n) ? (PT)0 : padkey;
        bid[j] = 0ull;
        owner[j] = (j < n) ? -1 : -2;
    }
    // -1: free row; -2: constant row, deferred to k_place_const (never bids, never searched)
    if (j < nrows) r2c[j] = (rconst && rconst[j]) ? -2 : -1;
    if (j < CTL_WORDS && j != CTL_FLAG && j != CTL_NCONST && j != CTL_RANGE && j != CTL_RANGE + 1) ctl[j] = 0;
    if (probe_cost) shape_probe(n, probe_cost, ctl, probe_tickets > 0 ? probe_tickets : (int)gridDim.x);   // uniform: every workgroup takes part in the ticket
}

// =====================================================================================
// k_bid: one wavefront per unassigned row
// =====================================================================================
// EPSM = true: the literal eps-scaling auction of the north star (comparison mode, TD_SOLVER=eps):
// costs scaled by kscale = n + 1, every bid raises the price by (second - best) + eps, no tie rule.
template <typename CT, bool LDSP, bool EPSM = false>
__global__ __launch_bounds__(LDSP ? 1024 : 256) void k_bid(int n, int nrows, int row0, int nchunks,
                                                          const CT *__restrict__ cc,
                                                          const typename Tr<CT>::PT *__restrict__ pk,
                                                          const int *__restrict__ r2c,
                                                          unsigned long long *__restrict__ bid,
                                                          const int *__restrict__ ctl, int round, int tie_evict,
                                                          long long kscale = 1, long long eps = 0,
                                                          int *__restrict__ tied = nullptr /* round 0: count rows tied at their minimum */,
                                                          const uint8_t *__restrict__ only = nullptr /* non-null: only the rows flagged here bid (td_core_warm.h) */)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = Tr<CT>::E;
    constexpr int U = 4;  // 16-byte row chunks in flight per lane
    extern __shared__ __align__(16) unsigned char smem[];
    if (ctl[CTL_FLAG]) return;  // speculated storage width did not fit: the host redoes the solve
    if (!EPSM && round > 0 && ctl[CTL_PROG + round - 1] == 0) return;  // previous round placed no bid: converged
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t pitch = (size_t)nchunks * E;
    const PT *P = pk;
    if (LDSP) {
        PT *sp = reinterpret_cast<PT *>(smem);
        const int npad = nchunks * E;
        for (int j = threadIdx.x; j < npad; j += blockDim.x) sp[j] = pk[j];
        __syncthreads();
        P = sp;
    }
    for (int lrow = blockIdx.x * nw + w; lrow < nrows; lrow += gridDim.x * nw) {
        if (r2c[lrow] != -1) continue;   // assigned, or a deferred constant row
        if (only && !only[lrow]) continue;
        const int row = row0 + lrow;  // global row id (shards own rows [row0, row0+nrows))
        const CT *rp = cc + (size_t)lrow * pitch;
        // Start chunk of the rotated scan.  It spreads the tie-breaks of different rows over the
        // columns (with first-index tie-breaking every row of a perf.jl instance would bid for the
        // same few columns) and changes every round, so an evicted row does not walk back to the
        // column it was just thrown out of.
        const uint32_t hsh = ((uint32_t)row + 1u) * 0x9E3779B1u + (uint32_t)round * 0x85EBCA6Bu;
        const int rot = (int)(((uint64_t)(hsh ^ (hsh >> 15)) * (uint64_t)nchunks) >> 32);
        PT k1 = Tr<CT>::KMAX, k2 = Tr<CT>::KMAX;
        int pos1 = 0;
        for (int t0 = lane; t0 < nchunks; t0 += 64 * U) {
            uint4 cv[U];
            int chs[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int t = t0 + 64 * u;
                int ch = t + rot;
                if (ch >= nchunks) ch -= nchunks;
                chs[u] = t < nchunks ? ch : -1;
                if (t < nchunks) cv[u] = load16_stream(rp + (size_t)ch * E);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (chs[u] < 0) continue;
                const int t = t0 + 64 * u;
                uint32_t c[E];
                unpack<CT>(cv[u], c);
                PT pv[E];
                if constexpr (sizeof(PT) == 4) {
                    const int4 *pp = reinterpret_cast<const int4 *>(P + (size_t)chs[u] * E);
#pragma unroll
                    for (int q = 0; q < E / 4; q++) {
                        const int4 x = pp[q];
                        pv[4 * q + 0] = x.x;
                        pv[4 * q + 1] = x.y;
                        pv[4 * q + 2] = x.z;
                        pv[4 * q + 3] = x.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; e++) pv[e] = P[(size_t)chs[u] * E + e];
                }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const PT key = (EPSM ? (PT)(2 * (PT)c[e] * (PT)kscale) : (PT)(2 * (PT)c[e])) + pv[e];  // 2*(cost+price) + owned
                    const bool lt = key < k1;
                    const PT mx = key > k1 ? key : k1;
                    k2 = k2 < mx ? k2 : mx;
                    pos1 = lt ? (t * E + e) : pos1;
                    k1 = lt ? key : k1;
                }
            }
        }
        // wave64 butterfly: lexicographic min of (key, rotated position)
        PT bk = k1;
        int bp = (k1 == Tr<CT>::KMAX) ? INT_MAX : pos1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            PT ok = shfl_xor_t(bk, o);
            int op = __shfl_xor(bp, o);
            if (ok < bk || (ok == bk && op < bp)) {
                bk = ok;
                bp = op;
            }
        }
        const bool winner = (k1 == bk) && (bp == pos1) && (k1 != Tr<CT>::KMAX);
        PT x = winner ? k2 : k1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            PT ox = shfl_xor_t(x, o);
            x = ox < x ? ox : x;
        }
        if (lane == 0 && bp != INT_MAX) {
            const int t1 = bp / E;
            int ch = t1 + rot;
            if (ch >= nchunks) ch -= nchunks;
            const int j1 = ch * E + (bp - t1 * E);
            const PT inc = (x == Tr<CT>::KMAX) ? (PT)0 : (PT)((x >> 1) - (bk >> 1));
            const bool owned = (bk & 1) != 0;
            // sampled (every 16th row): 16 384 same-address atomics would cost more than the round
            if (tied && inc == 0 && (n < 1024 || (row & 15) == 0)) atomicAdd(tied, 1);   // (small models: every row, 4 samples of a 60-row model say nothing)
            // A tie on an owned column raises no price.  With tie_evict the row still takes the
            // column (complementary slackness stays exact, the previous owner re-bids next round
            // and usually finds a free tied column); otherwise it is left to the finisher.
            if (j1 < n && (EPSM || !(owned && inc == 0) || tie_evict)) {
                const PT newp = (P[j1] >> 1) + inc + (EPSM ? (PT)eps : (PT)0);
                atomicMax(&bid[j1], ((unsigned long long)newp << ROW_BITS) | (unsigned long long)(row + 1));
            }
        }
    }
}

// reset of the assignment between eps phases (prices are kept)
template <typename PT>
__global__ void k_eps_reset(int n, int nrows, PT *pk, int *owner, int *r2c, int *ctl)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        pk[j] = (PT)((pk[j] >> 1) << 1);
        owner[j] = -1;
    }
    if (j < nrows) r2c[j] = -1;
    if (j >= CTL_PROG && j < CTL_WORDS) ctl[j] = 0;
}

// k_bid_row: the same bidding round with one WORKGROUP (256 threads) per unassigned row — every
// lane issues all its 16-byte loads of the row at once, so a row costs one memory round trip
// instead of four.  Used for the later rounds, where a handful of rows are left and the round is
// pure latency.
template <typename CT>
__global__ __launch_bounds__(256) void k_bid_row(int n, int nrows, int row0, int nchunks, const CT *__restrict__ cc,
                                                 const typename Tr<CT>::PT *__restrict__ P,
                                                 const int *__restrict__ r2c, unsigned long long *__restrict__ bid,
                                                 const int *__restrict__ ctl, int round, int tie_evict)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = Tr<CT>::E;
    constexpr int U = 4;
    __shared__ PT s_k[4], s_x[4];
    __shared__ int s_p[4];
    if (ctl[CTL_FLAG]) return;
    if (round > 0 && ctl[CTL_PROG + round - 1] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t pitch = (size_t)nchunks * E;
    // grid-stride over the rows: a late round has a handful of bidders, and spawning one workgroup per ROW just to
    // see that it is assigned cost more than the round's work (16 384 exits = 6.5 us; 2 048 looping workgroups = 2.5)
    for (int lrow = blockIdx.x; lrow < nrows; lrow += gridDim.x) {
    if (r2c[lrow] != -1) continue;   // uniform
    const int row = row0 + lrow;
    const CT *rp = cc + (size_t)lrow * pitch;
    const uint32_t hsh = ((uint32_t)row + 1u) * 0x9E3779B1u + (uint32_t)round * 0x85EBCA6Bu;
    const int rot = (int)(((uint64_t)(hsh ^ (hsh >> 15)) * (uint64_t)nchunks) >> 32);
    PT k1 = Tr<CT>::KMAX, k2 = Tr<CT>::KMAX;
    int pos1 = 0;
    for (int t0 = tid; t0 < nchunks; t0 += 256 * U) {
        uint4 cv[U];
        int chs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + 256 * u;
            int ch = t + rot;
            if (ch >= nchunks) ch -= nchunks;
            chs[u] = t < nchunks ? ch : -1;
            if (t < nchunks) cv[u] = *reinterpret_cast<const uint4 *>(rp + (size_t)ch * E);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (chs[u] < 0) continue;
            const int t = t0 + 256 * u;
            uint32_t c[E];
            unpack<CT>(cv[u], c);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const PT key = (PT)(2 * (PT)c[e]) + P[(size_t)chs[u] * E + e];
                const bool lt = key < k1;
                const PT mx = key > k1 ? key : k1;
                k2 = k2 < mx ? k2 : mx;
                pos1 = lt ? (t * E + e) : pos1;
                k1 = lt ? key : k1;
            }
        }
    }
    PT bk = k1;
    int bp = (k1 == Tr<CT>::KMAX) ? INT_MAX : pos1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        PT ok = shfl_xor_t(bk, o);
        int op = __shfl_xor(bp, o);
        if (ok < bk || (ok == bk && op < bp)) {
            bk = ok;
            bp = op;
        }
    }
    const bool winner = (k1 == bk) && (bp == pos1) && (k1 != Tr<CT>::KMAX);
    PT x = winner ? k2 : k1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        PT ox = shfl_xor_t(x, o);
        x = ox < x ? ox : x;
    }
    if (lane == 0) {
        s_k[w] = bk;
        s_p[w] = bp;
        s_x[w] = x;
    }
    __syncthreads();
    if (tid == 0) {
        int wb = 0;
        for (int k = 1; k < 4; k++)
            if (s_k[k] < s_k[wb] || (s_k[k] == s_k[wb] && s_p[k] < s_p[wb])) wb = k;
        bk = s_k[wb];
        bp = s_p[wb];
        x = s_x[wb];
        for (int k = 0; k < 4; k++)
            if (k != wb && s_k[k] < x) x = s_k[k];
        if (bp != INT_MAX) {
            const int t1 = bp / E;
            int ch = t1 + rot;
            if (ch >= nchunks) ch -= nchunks;
            const int j1 = ch * E + (bp - t1 * E);
            const PT inc = (x == Tr<CT>::KMAX) ? (PT)0 : (PT)((x >> 1) - (bk >> 1));
            const bool owned = (bk & 1) != 0;
            if (j1 < n && (!(owned && inc == 0) || tie_evict)) {
                const PT newp = (P[j1] >> 1) + inc;
                atomicMax(&bid[j1], ((unsigned long long)newp << ROW_BITS) | (unsigned long long)(row + 1));
            }
        }
    }
    __syncthreads();   // the reduction scratch is reused by the next row
    }
}

// =====================================================================================
// k_assign: one thread per column
// =====================================================================================
template <typename PT>
__global__ __launch_bounds__(256) void k_assign(int n, int nrows, int row0, unsigned long long *__restrict__ bid,
                                                PT *__restrict__ pk, int *__restrict__ owner, int *__restrict__ r2c,
                                                int *__restrict__ ctl, int round, long long plimit = 0 /* narrow-price mode: flag prices at or above */)
{
    if (ctl[CTL_FLAG]) return;
    if (round > 0 && round < 60 && ctl[CTL_PROG + round - 1] == 0) return;   // round >= 60: eps mode, always apply
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = 0;
    if (j < n) {
        const unsigned long long k = bid[j];
        if (k) {
            const int row = (int)(k & ((1ull << ROW_BITS) - 1)) - 1;
            const PT newp = (PT)(k >> ROW_BITS);
            if (plimit && (long long)(k >> ROW_BITS) >= plimit) atomicOr(&ctl[CTL_FLAG], 8);
            const int old = owner[j];
            const PT oldp = pk[j] >> 1;
            // r2c holds this shard's rows only; owner/price are replicated on every shard
            if (old >= row0 && old < row0 + nrows) r2c[old - row0] = -1;
            owner[j] = row;
            if (row >= row0 && row < row0 + nrows) r2c[row - row0] = j;
            pk[j] = (PT)(newp << 1) | (PT)1;
            bid[j] = 0ull;
            // progress = a free column got an owner, or a price went up (dual ascent).  A
            // price-neutral eviction (tie) is not progress: when a round consists only of those,
            // the rows are cycling through a tie class and the finisher takes over.
            cnt = (old < 0 || newp > oldp) ? 1 : 0;
        }
    }
    const unsigned long long m = __ballot(cnt);
    if ((threadIdx.x & 63) == 0 && m) {
        if (round < 60) atomicAdd(&ctl[CTL_PROG + round], (int)__popcll(m));
    }
}

// Ordered list of the entries equal to `want` (free rows: r2c == -1), built by ONE workgroup: every thread counts the free
// rows of its own contiguous slice, one block-wide exclusive scan (wave DPP-free shuffle scan +
// 16-entry LDS exchange), then every thread writes its rows at its offset.  Returns the count.
__device__ __forceinline__ int build_free_list(int n, const int *__restrict__ r2c, int *__restrict__ list, int want = -1)
{
    __shared__ int s_fl_w[16];
    __shared__ int s_fl_tot;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    const int per = (n + T - 1) / T;
    const int lo = tid * per, hi = min(n, lo + per);
    int cnt = 0;
    // 16-byte loads where a thread's slice allows it (n = 16 384 with 1024 threads: 4 independent loads instead of 16)
    const bool vec = (per & 3) == 0 && (reinterpret_cast<uintptr_t>(r2c) & 15) == 0;
    if (vec) {
        int r = lo;
#pragma unroll 4
        for (; r + 4 <= hi; r += 4) {
            const int4 x = *reinterpret_cast<const int4 *>(r2c + r);
            cnt += (x.x == want) + (x.y == want) + (x.z == want) + (x.w == want);
        }
        for (; r < hi; r++) cnt += (r2c[r] == want) ? 1 : 0;
    } else
        for (int r = lo; r < hi; r++) cnt += (r2c[r] == want) ? 1 : 0;
    int incl = cnt;  // inclusive scan within the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_fl_w[w] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += s_fl_w[k];
    if (tid == T - 1) s_fl_tot = base + incl;
    int pos = base + incl - cnt;
    if (cnt) {   // (most slices hold nothing to list)
        if (vec) {
            int r = lo;
            for (; r + 4 <= hi; r += 4) {
                const int4 x = *reinterpret_cast<const int4 *>(r2c + r);
                if (x.x == want) list[pos++] = r;
                if (x.y == want) list[pos++] = r + 1;
                if (x.z == want) list[pos++] = r + 2;
                if (x.w == want) list[pos++] = r + 3;
            }
            for (; r < hi; r++)
                if (r2c[r] == want) list[pos++] = r;
        } else
            for (int r = lo; r < hi; r++)
                if (r2c[r] == want) list[pos++] = r;
    }
    __syncthreads();
    const int tot = s_fl_tot;
    __syncthreads();   // the scratch is reused by a second call
    return tot;
}

__global__ __launch_bounds__(1024) void k_freelist(int n, const int *__restrict__ r2c, int *__restrict__ list,
                                                   int *__restrict__ ctl)
{
    if (ctl[CTL_FLAG]) return;
    const int cnt = build_free_list(n, r2c, list);
    if (threadIdx.x == 0) ctl[CTL_NFREE] = cnt;
}

#include "td_blocks.h"
#include "td_core_warm.h"

// Constant rows (every cell equal: dummy rows of a padded rectangular model, cabs with no request
// in range) cost the same in any column, so they sit out the bidding and the searches and take
// the columns nobody owns at the end, k-th constant row <- k-th free column. Exactness: a column
// that was never owned still has price 0 and prices never go below 0, so the row's dual
// min_j(c + p_j) is attained there and the LP certificate (k_dual) closes as before.
__device__ __forceinline__ void place_const_tail(int n, int *__restrict__ r2c, int *__restrict__ owner,
                                                 int *__restrict__ lista, int *__restrict__ listb,
                                                 int *__restrict__ ctl)
{
    if (ctl[CTL_NCONST] == 0) return;
    const int nr = build_free_list(n, r2c, lista, -2);
    if (nr == 0) return;
    const int nc = build_free_list(n, owner, listb, -1);
    if (nc != nr) {   // free rows left by the finisher, or a broken owner table
        if (threadIdx.x == 0) atomicOr(&ctl[CTL_ERR], 8);
        return;
    }
    for (int k = threadIdx.x; k < nr; k += blockDim.x) {
        const int r = lista[k], j = listb[k];
        r2c[r] = j;
        owner[j] = r;
    }
}

__global__ __launch_bounds__(1024) void k_place_const(int n, int *__restrict__ r2c, int *__restrict__ owner,
                                                      int *__restrict__ lista, int *__restrict__ listb,
                                                      int *__restrict__ ctl)
{
    if (ctl[CTL_FLAG]) return;
    place_const_tail(n, r2c, owner, lista, listb, ctl);
}

// Shape probe (td_assign, speculative attempt only, right after the compress pass): 16 sampled
// rows estimate how many COLUMNS are constant; the compress pass has counted the constant rows
// exactly. Many constant columns (dummy requests of a padded rectangular model,
// Simulator.java:244-252) mean the row-wise solve would have to price hundreds of rows out of the
// real columns, a price war of ~big_cost / (cost step) rounds; the transposed problem has them as
// constant ROWS, which are deferred. Sets CTL_FLAG bit 2 -> every later kernel of the attempt
// exits, the host transposes and redoes. The probe only picks the cheaper of two exact
// formulations; its estimate needs no guarantee.
// the last ticket holder decides (the constant rows have been counted by then: by the compress pass before this
// kernel, or — k_compress_reg<BID0>, which runs AFTER the state init — by the workgroups that took the other tickets)
__device__ __forceinline__ void shape_decide(int n, int *__restrict__ ctl)
{
    int *sh = ctl + CTL_SHAPE;
    __threadfence();
    const int ncol = atomicAdd(&sh[0], 0);
    const int nrow = atomicAdd(&ctl[CTL_NCONST], 0);
    sh[1] = nrow;
    // more constant columns than constant rows = more real rows than real columns: every surplus real row
    // would need a search that scans all real columns before it reaches a dummy one (~30 us each, and too
    // long a record for the speculative batches); in the transposed problem every real row finds a real column
    const int margin = n / 256 > 32 ? n / 256 : 32;
    if (ncol >= 16 && ncol - nrow >= margin) {
        sh[3] = 1;
        atomicOr(&ctl[CTL_FLAG], 4);
    }
}

__device__ __forceinline__ void shape_probe(int n, const int32_t *__restrict__ cost, int *__restrict__ ctl, int tickets)
{
    int *sh = ctl + CTL_SHAPE;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int cc = 0;
    long long rng = 0;
    if (t < n) {   // column t over 16 sampled rows: coalesced
        const int v0 = cost[t];
        int mn = v0, mx = v0;
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const int v = cost[(int64_t)(((int64_t)k * n) >> 4) * n + t];
            mn = min(mn, v);
            mx = max(mx, v);
        }
        cc = (mn == mx) ? 1 : 0;
        rng = (long long)mx - (long long)mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cc += __shfl_xor(cc, o);
        const long long orng = __shfl_xor(rng, o);
        rng = orng > rng ? orng : rng;
    }
    if ((threadIdx.x & 63) == 0) {
        if (cc) atomicAdd(&sh[0], cc);
        // lower bound of the transposed problem's row range: lets the redo skip widths that cannot fit
        if (rng > 254) atomicMax(reinterpret_cast<unsigned long long *>(&sh[4]), (unsigned long long)rng);
    }
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = (atomicAdd(&sh[2], 1) == tickets - 1);
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) shape_decide(n, ctl);
}

// A model's cell made on the fly from the position arrays — td_cost_build's positional rule (greedy_opt.py:86-99,
// simulate.py:17-33, Simulator.java:493-520): cost[i][j] = dist[cab_to[i]][dem_from[j]] for a real pair whose distance is
// below the threshold, else `fill`.  td_build_assign hands this to the fused transposing compress pass and to k_final
// instead of an int32 matrix: the 4 N^2-byte matrix of a padded / thresholded model is then neither written nor read.
struct CellSrc {
    const int32_t *cab_to = nullptr, *dem_from = nullptr, *dist = nullptr;   // device arrays; dist == null: |a - b|
    int n_s = 0, n_d = 0, S = 0;
    int32_t fill = 0, thr = -1;
    __device__ __forceinline__ int cell(int i, int j) const
    {
        if (i >= n_s || j >= n_d) return fill;
        const int a = cab_to[i], b = dem_from[j];
        int x;
        if (dist) {
            if ((uint32_t)a >= (uint32_t)S || (uint32_t)b >= (uint32_t)S) return fill;   // never index outside the table (k_cost_build)
            x = dist[(int64_t)a * S + b];
        } else
            x = a > b ? a - b : b - a;
        return (thr < 0 || x < thr) ? x : fill;
    }
    __device__ __forceinline__ int4 load4(int i, int j0) const { return make_int4(cell(i, j0), cell(i, j0 + 1), cell(i, j0 + 2), cell(i, j0 + 3)); }
};

// out[j][i] = in[i][j], 64 x 64 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void k_transpose(int n, const int32_t *__restrict__ in, int32_t *__restrict__ out)
{
    __shared__ int32_t tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int i = by + r, j = bx + tx;
        if (i < n && j < n) tile[r][tx] = in[(int64_t)i * n + j];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int j = bx + r, i = by + tx;
        if (i < n && j < n) out[(int64_t)j * n + i] = tile[tx][r];
    }
}

// Fused transpose + compress for padded models (dummy requests = constant trailing columns, Simulator.java:493-520,
// simulate.py:17-33): ONE pass over the caller's int32 matrix writes the TRANSPOSED problem's 4-byte cells
// out[j][i] = in[i][j] - base (pitch npad, pad cells = sentinel) — instead of a void 1-byte compress pass, a
// transpose into a second int32 matrix and a 4-byte compress pass of that.  The transposed rows keep `base` as
// their row constant (any constant <= the row minimum is a valid row dual); their minimum / maximum (constant
// rows = dummy requests are deferred, the range decides the price width) are reduced per workgroup in LDS and
// merged with one atomic per column and workgroup.  A cell below base or a range beyond 32 bits raises CTL_FLAG:
// the host then takes the general path.
template <int RT, bool GEN = false>
__global__ __launch_bounds__(256) void k_compress_tr(int n, int npad, const int32_t *__restrict__ in, uint32_t *__restrict__ out, int base,
                                                     int *__restrict__ colmin, int *__restrict__ colmax, int *__restrict__ ctl, const CellSrc src = CellSrc())
{
    __shared__ int32_t tile[64][65];
    __shared__ int s_mn[64], s_mx[64];
    const int bx = blockIdx.x * 64, ry0 = blockIdx.y * (64 * RT);
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    if (tid < 64) {
        s_mn[tid] = INT_MAX;
        s_mx[tid] = INT_MIN;
    }
    bool bad = false;
    for (int rt = 0; rt < RT; rt++) {
        const int by = ry0 + rt * 64;
        if (by >= n) break;
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            const int i = by + r, j = bx + tx;
            tile[r][tx] = (i < n && j < n) ? (GEN ? src.cell(i, j) : in[(int64_t)i * n + j]) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = (tid >> 4) + 16 * k, cq = (tid & 15) * 4;
            const int j = bx + rr;
            if (j < n && by + cq < npad) {
                uint32_t o[4];
                int mn = INT_MAX, mx = INT_MIN;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int v = tile[cq + e][rr];
                    if (by + cq + e < n) {
                        mn = min(mn, v);
                        mx = max(mx, v);
                        o[e] = (uint32_t)(v - base);
                        bad = bad || v < base;
                    } else
                        o[e] = 0xFFFFFFFFu;
                }
                *reinterpret_cast<uint4 *>(out + (size_t)j * npad + by + cq) = make_uint4(o[0], o[1], o[2], o[3]);
                if (mn <= mx) {
                    atomicMin(&s_mn[rr], mn);
                    atomicMax(&s_mx[rr], mx);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 64 && bx + tid < n && s_mn[tid] <= s_mx[tid]) {
        atomicMin(&colmin[bx + tid], s_mn[tid]);
        atomicMax(&colmax[bx + tid], s_mx[tid]);
    }
    if (bad) atomicOr(&ctl[CTL_FLAG], 1);
}

// The same pass for 1-byte cells with an escape (u8e): every cell is `esc_raw` (the fill value, code 254) or lies in
// base .. base + 253, else CTL_FLAG is raised and the host redoes the pass with 4-byte cells.  A workgroup takes 64 of
// the caller's columns x TR8_ROWS rows (512): the codes are staged transposed in LDS (row pitch TR8_ROWS + 4 bytes, an odd
// number of dwords: the 64 lanes of a wave hit 64 different banks), then every transposed row leaves as one contiguous piece.
template <int TR8_ROWS>
__global__ __launch_bounds__(256) void k_compress_tr8(int n, int npad, const int32_t *__restrict__ in, uint8_t *__restrict__ out, int base,
                                                      int esc_raw, int *__restrict__ colmin, int *__restrict__ colmax, int *__restrict__ ctl)
{
    constexpr int TR8_LP = TR8_ROWS + 4;
    static_assert(TR8_ROWS % 256 == 0, "a transposed row leaves as TR8_ROWS / 256 dwords per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char tb[];   // [64][TR8_LP]
    __shared__ int s_mn[64], s_mx[64];
    const int bx = blockIdx.x * 64, ry0 = blockIdx.y * TR8_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 64) {
        s_mn[tid] = INT_MAX;
        s_mx[tid] = INT_MIN;
    }
    __syncthreads();
    // Load phase: lane (cg = lane % 16, rs = lane / 16) of wave w reads 16 bytes = columns 4cg .. 4cg+3 of row 16p + 4w + rs
    // (a wave reads 4 full 256-byte row pieces per instruction, 8 instructions in flight).  The four lanes cg, cg+16,
    // cg+32, cg+48 hold a 4 x 4 block (4 rows x 4 columns): each packs its row's codes into one dword, fetches the
    // other three by shuffles and writes ONE dword = 4 consecutive rows of column 4cg + rs — a conflict-free
    // ds_write_b32 (dword address (4cg + rs) * 257 + row / 4: 64 different banks over the wave).
    const int cg = lane & 15, rs = lane >> 4;
    int mn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, mx[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
    bool bad = false;
    const int jc0 = bx + 4 * cg;
    const bool vec = (n % 4 == 0);
    constexpr int U = 8;
    for (int p0 = 0; p0 < TR8_ROWS / 16; p0 += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = 16 * (p0 + u) + 4 * wv + rs, i = ry0 + r;
            if (i < n && vec && jc0 + 3 < n)
                v[u] = *reinterpret_cast<const int4 *>(in + (int64_t)i * n + jc0);
            else {
                int t4[4];
#pragma unroll
                for (int x = 0; x < 4; x++) t4[x] = (i < n && jc0 + x < n) ? in[(int64_t)i * n + jc0 + x] : esc_raw;
                v[u] = make_int4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = 16 * (p0 + u) + 4 * wv + rs, i = ry0 + r;
            const int vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t packed = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                uint32_t code = 0xFFu;
                if (i < n && jc0 + x < n) {
                    mn[x] = min(mn[x], vv[x]);
                    mx[x] = max(mx[x], vv[x]);
                    const int d = vv[x] - base;
                    if (vv[x] == esc_raw)
                        code = U8E_ESC;
                    else {
                        code = (uint32_t)d & 0xFFu;
                        bad = bad || d < 0 || d > 253;
                    }
                }
                packed |= code << (8 * x);
            }
            // 4 x 4 transpose over the lanes cg + 16 * q: my dword = byte `rs` of the packed rows q = 0 .. 3
            uint32_t mine = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t pq = (uint32_t)__shfl((int)packed, cg + 16 * q);
                mine |= ((pq >> (8 * rs)) & 0xFFu) << (8 * q);
            }
            const int rbase = 16 * (p0 + u) + 4 * wv;   // the four rows of this block
            *reinterpret_cast<uint32_t *>(tb + (size_t)(4 * cg + rs) * TR8_LP + rbase) = mine;
        }
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
        if (mn[x] <= mx[x]) {
            atomicMin(&s_mn[4 * cg + x], mn[x]);
            atomicMax(&s_mx[4 * cg + x], mx[x]);
        }
    __syncthreads();
    // wave wv writes the transposed rows wv * 16 .. + 15: lane l holds the dwords l, l + 64, .. of the row
    for (int k = 0; k < 16; k++) {
        const int rr = wv * 16 + k, jj = bx + rr;
        if (jj >= n) break;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tb + (size_t)rr * TR8_LP);
#pragma unroll
        for (int q = 0; q < TR8_ROWS / 256; q++) {
            const int dw = lane + 64 * q, b = ry0 + 4 * dw;
            if (b < npad) *reinterpret_cast<uint32_t *>(out + (size_t)jj * npad + b) = src[dw];
        }
    }
    if (tid < 64 && bx + tid < n && s_mn[tid] <= s_mx[tid]) {
        atomicMin(&colmin[bx + tid], s_mn[tid]);
        atomicMax(&colmax[bx + tid], s_mx[tid]);
    }
    if (bad) atomicOr(&ctl[CTL_FLAG], 1);
}

// The same pass with WIDE tiles: 256 of the caller's columns x TR8W_ROWS rows per workgroup.  Wave w takes the column group
// [64w, 64w + 64): one instruction of the four waves together reads 4 rows x 1 KiB CONTIGUOUS (the narrow tile read 256-byte
// pieces at a 64 KiB stride: 3.8 TB/s of traffic), every transposed row leaves as TR8W_ROWS contiguous bytes.
template <int TR8W_ROWS, bool GEN = false>
__global__ __launch_bounds__(256) void k_compress_tr8w(int n, int npad, const int32_t *__restrict__ in, uint8_t *__restrict__ out, int base,
                                                       int esc_raw, int *__restrict__ colmin, int *__restrict__ colmax, int *__restrict__ ctl,
                                                       const CellSrc src = CellSrc())
{
    constexpr int LP = TR8W_ROWS + 4;
    static_assert(TR8W_ROWS % 128 == 0 && TR8W_ROWS <= 256, "a transposed row leaves as TR8W_ROWS / 4 dwords: 32 or 64 lanes");
    extern __shared__ __attribute__((aligned(16))) unsigned char tb[];   // [256][LP]
    __shared__ int s_mn[256], s_mx[256];
    const int bx = blockIdx.x * 256, ry0 = blockIdx.y * TR8W_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_mn[tid] = INT_MAX;
    s_mx[tid] = INT_MIN;
    __syncthreads();
    const int cg = lane & 15, rs = lane >> 4;
    int mn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, mx[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
    bool bad = false;
    const int lc0 = 64 * wv + 4 * cg;   // the lane's first column inside the tile
    const int jc0 = bx + lc0;
    const bool vec = (n % 4 == 0);
    constexpr int U = 8;
    for (int p0 = 0; p0 < TR8W_ROWS / 4; p0 += U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = 4 * (p0 + u) + rs, i = ry0 + r;
            if constexpr (GEN) {
                int t4[4];
#pragma unroll
                for (int x = 0; x < 4; x++) t4[x] = (i < n && jc0 + x < n) ? src.cell(i, jc0 + x) : esc_raw;
                v[u] = make_int4(t4[0], t4[1], t4[2], t4[3]);
            } else if (i < n && vec && jc0 + 3 < n)
                v[u] = *reinterpret_cast<const int4 *>(in + (int64_t)i * n + jc0);
            else {
                int t4[4];
#pragma unroll
                for (int x = 0; x < 4; x++) t4[x] = (i < n && jc0 + x < n) ? in[(int64_t)i * n + jc0 + x] : esc_raw;
                v[u] = make_int4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = 4 * (p0 + u) + rs, i = ry0 + r;
            const int vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t packed = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                uint32_t code = 0xFFu;
                if (i < n && jc0 + x < n) {
                    mn[x] = min(mn[x], vv[x]);
                    mx[x] = max(mx[x], vv[x]);
                    const int d = vv[x] - base;
                    if (vv[x] == esc_raw)
                        code = U8E_ESC;
                    else {
                        code = (uint32_t)d & 0xFFu;
                        bad = bad || d < 0 || d > 253;
                    }
                }
                packed |= code << (8 * x);
            }
            // 4 x 4 transpose over the lanes cg + 16 * q (rows 4(p0+u) + q): my dword = the four rows of column lc0 + rs
            uint32_t mine = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t pq = (uint32_t)__shfl((int)packed, cg + 16 * q);
                mine |= ((pq >> (8 * rs)) & 0xFFu) << (8 * q);
            }
            *reinterpret_cast<uint32_t *>(tb + (size_t)(lc0 + rs) * LP + 4 * (p0 + u)) = mine;
        }
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
        if (mn[x] <= mx[x]) {
            atomicMin(&s_mn[lc0 + x], mn[x]);
            atomicMax(&s_mx[lc0 + x], mx[x]);
        }
    __syncthreads();
    // wave wv writes the transposed rows 64 wv .. + 63, 64 / (TR8W_ROWS / 4) of them per instruction
    constexpr int DW = TR8W_ROWS / 4, RPI = 64 / DW;
    for (int k = 0; k < 64; k += RPI) {
        const int rr = wv * 64 + k + lane / DW, jj = bx + rr;
        const int dw = lane % DW, b = ry0 + 4 * dw;
        if (jj < n && b < npad)
            *reinterpret_cast<uint32_t *>(out + (size_t)jj * npad + b) = *reinterpret_cast<const uint32_t *>(tb + (size_t)rr * LP + 4 * dw);
    }
    if (bx + tid < n && s_mn[tid] <= s_mx[tid]) {
        atomicMin(&colmin[bx + tid], s_mn[tid]);
        atomicMax(&colmax[bx + tid], s_mx[tid]);
    }
    if (bad) atomicOr(&ctl[CTL_FLAG], 1);
}

__global__ void k_tr_finish(int n, int base, const int *__restrict__ colmin, const int *__restrict__ colmax, int32_t *__restrict__ rowmin,
                            int *__restrict__ rconst, int *__restrict__ ctl, int esc_raw = 0, int with_esc = 0,
                            long long assumed_range = -1 /* >= 0: the speculative pass went on as if every cell were <= base + this (32-bit prices, BIG = 2^28): a larger cell voids the attempt */)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && with_esc) g_esc8 = (uint32_t)(esc_raw - base);
    if (j >= n) return;
    const int mn = colmin[j], mx = colmax[j];
    if (assumed_range >= 0 && (long long)mx - (long long)base > assumed_range) atomicOr(&ctl[CTL_FLAG], 1);
    rowmin[j] = base;
    const int cst = (mn == mx) ? 1 : 0;
    rconst[j] = cst;
    if (cst) atomicAdd(&ctl[CTL_NCONST], 1);
    atomicMax(reinterpret_cast<unsigned long long *>(&ctl[CTL_RANGE]), (unsigned long long)((int64_t)mx - (int64_t)base));
}

// =====================================================================================
// k_sap: shortest augmenting paths, one persistent workgroup
// =====================================================================================

// wave64 unsigned min through DPP (row_shr 1/2/4/8, row_bcast 15/31): six VALU ops instead of
// six LDS-routed ds_bpermute shuffles; the result is returned wave-uniform.
__device__ __forceinline__ uint32_t wave_umin32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x111, 0xF, 0xF, false);
    v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x112, 0xF, 0xF, false);
    v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x114, 0xF, 0xF, false);
    v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x118, 0xF, 0xF, false);
    v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x142, 0xA, 0xF, false);
    v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, 0x143, 0xC, 0xF, false);
    v = v < t ? v : t;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// wave-wide argmin of (key, j) with payloads (owner o, price p); result uniform in every lane.
template <typename PT>
__device__ __forceinline__ void wave_argmin(PT &key, int &j, int &o, PT &p)
{
    if constexpr (sizeof(PT) == 4) {
        const uint32_t m = wave_umin32((uint32_t)key);
        // among lanes at the minimum key the smallest j wins (deterministic, and lets callers
        // encode a tie-break in j)
        const uint32_t jm = wave_umin32(((uint32_t)key == m) ? (uint32_t)j : 0xFFFFFFFFu);
        const unsigned long long b = __ballot((uint32_t)key == m && (uint32_t)j == jm);
        const int L = __ffsll((long long)b) - 1;
        j = (int)jm;
        o = __builtin_amdgcn_readlane(o, L);
        p = (PT)__builtin_amdgcn_readlane((int)p, L);
        key = (PT)m;
    } else {
        // 64-bit keys (non-negative): three DPP min reductions (high word, low word among the
        // high-word winners, then j) instead of 36 LDS-routed shuffles
        const uint32_t hi = (uint32_t)((unsigned long long)key >> 32), lo = (uint32_t)key;
        const uint32_t mh = wave_umin32(hi);
        const uint32_t ml = wave_umin32(hi == mh ? lo : 0xFFFFFFFFu);
        const bool at = (hi == mh) && (lo == ml);
        const uint32_t jm = wave_umin32(at ? (uint32_t)j : 0xFFFFFFFFu);
        const unsigned long long b = __ballot(at && (uint32_t)j == jm);
        const int L = __ffsll((long long)b) - 1;
        j = (int)jm;
        o = __builtin_amdgcn_readlane(o, L);
        const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)p, L);
        const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)p >> 32), L);
        p = (PT)(((unsigned long long)phi << 32) | plo);
        key = (PT)(((unsigned long long)mh << 32) | ml);
    }
}

// Row shards of the compressed matrix as seen from the finisher's device: local memory, or
// peer memory of other GPUs mapped over xGMI (hipIpc) — row o lives in shard o / rps.
struct ShardTab {
    const void *p[16];
    int rps;
    int count;
};

template <typename CT>
__device__ __forceinline__ const CT *shard_row(const ShardTab &tab, int o, size_t pitch)
{
    const int sh = (tab.count == 1) ? 0 : o / tab.rps;
    return reinterpret_cast<const CT *>(tab.p[sh]) + (size_t)(o - sh * tab.rps) * pitch;
}

constexpr int PS_G = 192;     // searches per batch
constexpr int PS_CAP = 4096;  // finalised columns recorded per search; longer searches are left to the serial finisher

template <typename PT>
struct PsRec {
    int f, endcol, nS, plen, status;  // status 1 = usable
    int pad[3];
    PT mind;
    int S_col[PS_CAP];
    PT S_d[PS_CAP];
    int path[PS_CAP + 2];
};

#ifndef TD_G512
#define TD_G512 8
#endif
#ifndef TD_G_U32
#define TD_G_U32 4
#endif
#ifndef TD_RELAX_ANY
#define TD_RELAX_ANY 1
#endif
constexpr int SAPB_W = 16;  // generic finisher: columns scanned per step (frontier + window)

// TB: largest workgroup the instance is launched with. 512 threads halve the waves per SIMD and so
// double the register budget (256 VGPRs), which is what lets a step keep 8-16 rows in flight.
// SPEC = false: the serial finisher (applies every augmentation itself).
// SPEC = true : one speculative search per workgroup against a read-only snapshot (nothing global
//               is written but the record recs[blockIdx.x], applied or rejected by k_pcommit).
template <typename CT, int CH, bool LDSST, int TB = 1024, bool SPEC = false>
__global__ __launch_bounds__(TB) void k_sap(int n, int nchunks, const ShardTab tab,
                                              typename Tr<CT>::PT *__restrict__ pk, int *__restrict__ owner_g,
                                              int *__restrict__ r2c, int *__restrict__ pred_g, int *__restrict__ list,
                                              int *__restrict__ ctl, PsRec<typename Tr<CT>::PT> *__restrict__ recs = nullptr)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = Tr<CT>::E;
    constexpr int NV = CH * E;
    constexpr bool PREG = (NV * (int)sizeof(PT) / 4) <= 32;  // cache own prices / owners in registers
    constexpr PT KMAX = Tr<CT>::KMAX;
    static_assert(NV <= 64, "scanned mask is 64 bits");
    static_assert(!SPEC || (PREG && LDSST), "speculative searches keep prices in registers and owner/pred in LDS");
    // LDS: owner[] and pred[] (random access by the path walk and the argmin winner).  Prices
    // stay in registers during a search (global memory between searches), and NO global store
    // happens inside the step loop: a __syncthreads() drains vmcnt, so a store there would put
    // a full memory round trip on the critical path of every step.
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ PT s_rk[2][16];
    __shared__ PT s_rp[2][16];
    __shared__ int s_rj[2][16];
    __shared__ int s_ro[2][16];
    // batch of columns finalised together in one step (see the step loop)
    __shared__ PT s_bd[SAPB_W], s_bp[SAPB_W];
    __shared__ int s_bcol[SAPB_W], s_bown[SAPB_W];
    __shared__ int s_wcnt[16];

    if (ctl[CTL_FLAG]) return;
    if (SPEC && (ctl[CTL_PSTOP] || (int)blockIdx.x >= ctl[CTL_NFREE])) return;
    if (!SPEC && ctl[CTL_NFREE] <= 0) return;   // nothing was left free: no price unpack / re-pack round trip over all columns
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    const int npad = nchunks * E;
    const size_t pitch = (size_t)npad;
    PT *P = pk;  // unpacked in place below, re-packed at the end
    int *OWN = LDSST ? reinterpret_cast<int *>(smem) : owner_g;
    int *PRED = LDSST ? reinterpret_cast<int *>(smem + (size_t)npad * sizeof(int)) : pred_g;

    for (int j = tid; j < npad; j += T) {
        if (!SPEC) P[j] = pk[j] >> 1;
        if (LDSST) OWN[j] = owner_g[j];
    }
    // the ordered list of free rows was built by k_freelist / the last k_pcommit
    const int nfree = ctl[CTL_NFREE];
    __syncthreads();

    // columns owned by this thread: chunk q*T + tid, q = 0..CH-1
    unsigned long long padmask = 0ull;  // bits of columns that do not exist
#pragma unroll
    for (int q = 0; q < CH; q++) {
        const int ch = q * T + tid;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int j = ch * E + e;
            if (ch >= nchunks || j >= n) padmask |= 1ull << (q * E + e);
        }
    }

    long long steps = 0;
    int par = 0;
    bool bad = false;
    PT delta = 0;   // batch window above the frontier distance, adapted step by step (wave-uniform)
    // own prices stay in registers for the whole kernel when they fit (PREG): a thread is the
    // only writer of its columns' prices, so global memory sees them once, at the end
    PT preg[PREG ? NV : 1];
    if (PREG) {
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int ch = q * T + tid;
#pragma unroll
            for (int e = 0; e < E; e++)
                preg[q * E + e] = (ch < nchunks) ? (SPEC ? (PT)(pk[ch * E + e] >> 1) : P[ch * E + e]) : (PT)0;
        }
    }
    // software prefetch of the NEXT free row (its id and its 16-byte chunks) behind the current search
    // SPEC: the searches of a batch are spread evenly over the (row-ordered) free list: neighbouring
    // rows tend to want the same columns (cabs at one stand) and would only reject each other
    const int gact = nfree < (int)gridDim.x ? nfree : (int)gridDim.x;
    int fnext = nfree > 0 ? list[SPEC ? (int)(((long long)blockIdx.x * nfree) / gact) : 0] : 0;
    // SPEC: ties (equal distance, same owned bit) are broken by a column order rotated per search:
    // with the plain lowest-index rule every search of a batch would end in the same free column
    // of a tie class and all but one would be rejected
    const int rot = SPEC ? (int)(((uint64_t)(((uint32_t)fnext + 1u) * 0x9E3779B1u) * (uint64_t)npad) >> 32) : 0;
    uint4 cvn[CH];
    if (nfree > 0) {
        const CT *nrow = shard_row<CT>(tab, fnext, pitch);
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int ch = q * T + tid;
            if (ch < nchunks) cvn[q] = *reinterpret_cast<const uint4 *>(nrow + (size_t)ch * E);
        }
    }
    for (int fi = 0; fi < (SPEC ? 1 : nfree) && !bad; fi++) {
        const int f = fnext;
        uint4 cvf[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) cvf[q] = cvn[q];
        if (!SPEC && fi + 1 < nfree) {
            fnext = list[fi + 1];
            const CT *nrow = shard_row<CT>(tab, fnext, pitch);
#pragma unroll
            for (int q = 0; q < CH; q++) {
                const int ch = q * T + tid;
                if (ch < nchunks) cvn[q] = *reinterpret_cast<const uint4 *>(nrow + (size_t)ch * E);
            }
        }
        PT d[NV];
        PT dp[NV];   // d - price, the per-column side of the relax test; pad columns can never improve
        int ownr[PREG ? NV : 1];
        unsigned long long scanned = padmask;
        // bit set <=> the column has an owner (or does not exist).  Folded into the argmin key
        // so that among columns at the same distance a FREE one is taken first: with heavily
        // tied costs (perf.jl, simulator instances) a search ends as soon as any free column
        // reaches the frontier distance instead of scanning the whole tie class.
        unsigned long long ownedmask = padmask;
        // distances from the free row f (the row dual of f is a constant shift: left out)
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int ch = q * T + tid;
            if (ch < nchunks) {
                uint32_t c[E];
                unpack<CT>(cvf[q], c);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int j = ch * E + e;
                    const PT p = PREG ? preg[q * E + e] : P[j];
                    const int o = OWN[j];
                    if (PREG) ownr[q * E + e] = o;
                    if (o != -1) ownedmask |= 1ull << (q * E + e);
                    d[q * E + e] = (PT)c[e] + p;
                    dp[q * E + e] = (j < n) ? (PT)c[e] : -(KMAX >> 2);
                    PRED[j] = -1;   // predecessor COLUMN; -1 = reached from the root row
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    d[q * E + e] = KMAX >> 2;
                    dp[q * E + e] = -(KMAX >> 2);
                    if (PREG) ownr[q * E + e] = -2;
                }
            }
        }
        PT mind = 0;
        int endcol = -1;
        for (int guard = 0; guard <= npad; guard++) {
            // block-wide argmin over unscanned columns of key = 2*d + owned
            PT bk = KMAX, bp = 0;
            int bj = INT_MAX, bo = -2;
#pragma unroll
            for (int q = 0; q < CH; q++) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool ok = !((scanned >> (q * E + e)) & 1ull);
                    const PT v = (PT)(d[q * E + e] << 1) | (PT)((ownedmask >> (q * E + e)) & 1ull);
                    int jr = (q * T + tid) * E + e - rot;   // rotated index while reducing (SPEC)
                    jr += (jr < 0) ? npad : 0;
                    if (ok && (v < bk || (SPEC && v == bk && jr < bj))) {
                        bk = v;
                        bj = jr;
                        if (PREG) {
                            bo = ownr[q * E + e];
                            bp = preg[q * E + e];
                        }
                    }
                }
            }
            wave_argmin<PT>(bk, bj, bo, bp);
            if (nw > 1) {
                if (lane == 0) {
                    s_rk[par][w] = bk;
                    s_rj[par][w] = bj;
                    s_ro[par][w] = bo;
                    s_rp[par][w] = bp;
                }
                __syncthreads();
                const bool has = lane < nw;
                bk = has ? s_rk[par][lane] : KMAX;
                bj = has ? s_rj[par][lane] : INT_MAX;
                bo = has ? s_ro[par][lane] : -2;
                bp = has ? s_rp[par][lane] : (PT)0;
                wave_argmin<PT>(bk, bj, bo, bp);
                par ^= 1;
            }
            if (bj == INT_MAX) {
                bad = true;
                break;
            }
            bj += rot;  // back to the column index
            bj -= (bj >= npad) ? npad : 0;
            const bool col_owned = (bk & 1) != 0;
            const PT bd = bk >> 1;  // the distance
            if (!col_owned) {      // free column reached
                mind = bd;
                endcol = bj;
                if (SPEC) {
                    // Usually a whole class of free columns is tied at the end distance (dummy
                    // columns of a simulator model): take the k-th of them, k spread over the
                    // searches of the batch, so that concurrent searches end in DIFFERENT columns
                    // instead of rejecting each other at commit.
                    unsigned long long tiemask = 0ull;
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        const bool tie = !((scanned >> k) & 1ull) && !((ownedmask >> k) & 1ull) && d[k] == bd;
                        tiemask |= tie ? (1ull << k) : 0ull;
                    }
                    const int mycnt = __popcll(tiemask);
                    int incl = mycnt;
#pragma unroll
                    for (int sh = 1; sh < 64; sh <<= 1) {
                        const int v = __shfl_up(incl, sh);
                        if (lane >= sh) incl += v;
                    }
                    if (lane == 63) s_wcnt[w] = incl;
                    __syncthreads();
                    int base = 0, total = 0;
                    for (int q = 0; q < nw; q++) {
                        const int cw = s_wcnt[q];
                        base += (q < w) ? cw : 0;
                        total += cw;
                    }
                    const int target = (int)(((uint64_t)(((uint32_t)blockIdx.x + 1u) * 0x9E3779B1u) * (uint64_t)total) >> 32);
                    const int lo = base + incl - mycnt;
                    if (target >= lo && target < lo + mycnt) {
                        int skip = target - lo;
#pragma unroll
                        for (int k = 0; k < NV; k++) {
                            if ((tiemask >> k) & 1ull) {
                                if (skip == 0) s_bcol[0] = ((k / E) * T + tid) * E + (k % E);
                                skip--;
                            }
                        }
                    }
                    __syncthreads();
                    if (total > 0) endcol = s_bcol[0];
                    __syncthreads();
                }
                break;
            }
            const int o = PREG ? bo : OWN[bj];
            steps++;
            // ---- batch: besides the frontier column bj, up to SAPB_W-1 more owned, not yet
            // scanned columns within `delta` of the frontier distance are scanned in the same
            // step (their owners' rows are streamed together: one memory round trip and one
            // barrier pair instead of one per column). Their labels need not be final: a column
            // whose label drops after it was scanned becomes unscanned again (label-correcting),
            // so at the end every column below the end distance carries its exact distance
            // (argument in DESIGN.md, "batched finisher steps") and ties / paths stay exact.
            unsigned long long cand = 0ull;
            {
                const PT lim = bd + delta;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const bool ok = !((scanned >> k) & 1ull) && ((ownedmask >> k) & 1ull) && !((padmask >> k) & 1ull) &&
                                    d[k] <= lim;
                    cand |= ok ? (1ull << k) : 0ull;
                }
                const int chq = bj / E;   // the frontier column is entry 0, not a candidate
                if ((chq % T) == tid) {
                    const unsigned long long bit = 1ull << ((chq / T) * E + (bj - chq * E));
                    cand &= ~bit;
                    scanned |= bit;
                }
            }
            const int mycnt = __popcll(cand);
            int incl = mycnt;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int v = __shfl_up(incl, sh);
                if (lane >= sh) incl += v;
            }
            if (lane == 63) s_wcnt[w] = incl;
            if (tid == 0) {
                s_bcol[0] = bj;
                s_bd[0] = bd;
                s_bown[0] = o;
                s_bp[0] = PREG ? bp : P[bj];
            }
            __syncthreads();
            int base = 1, total = 0;   // slot 0 is the frontier column
            for (int q = 0; q < nw; q++) {
                const int cw = s_wcnt[q];
                base += (q < w) ? cw : 0;
                total += cw;
            }
            {
                int pos = base + incl - mycnt;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (((cand >> k) & 1ull) && pos < SAPB_W) {
                        const int jc = ((k / E) * T + tid) * E + (k % E);
                        s_bcol[pos] = jc;
                        s_bd[pos] = d[k];
                        s_bown[pos] = PREG ? ownr[k] : OWN[jc];
                        s_bp[pos] = PREG ? preg[k] : P[jc];
                        scanned |= 1ull << k;
                        pos++;
                    }
                }
            }
            const int nb = min(total + 1, SAPB_W);
            // window control: aim at a batch that is about full
            if (total + 1 > SAPB_W)
                delta >>= 1;
            else if (total + 1 <= SAPB_W / 2)
                delta = (delta < (KMAX >> 4)) ? delta * 2 + 1 : delta;
            __syncthreads();
            // stream the rows of the batch columns' owners and relax; SAPB_G rows in flight at a time
            constexpr int SAPB_G = (TB <= 512) ? ((NV <= 8) ? TD_G512 : 8) : ((CH == 1) ? ((E == 4) ? TD_G_U32 : 4) : ((CH == 2) ? 2 : 1));
            for (int e0 = 0; e0 < nb; e0 += SAPB_G) {
                uint4 cv[SAPB_G][CH];
                PT wst[SAPB_G], de[SAPB_G];
                int ce[SAPB_G];
                // branch-free: slots past the batch repeat its last entry (relaxing twice is harmless),
                // so all loads of the group are issued back to back
#pragma unroll
                for (int g = 0; g < SAPB_G; g++) {
                    const int e = min(e0 + g, nb - 1);
                    const int oc = s_bown[e];
                    ce[g] = s_bcol[e];
                    de[g] = s_bd[e];
                    const CT *rp = shard_row<CT>(tab, oc, pitch);
                    wst[g] = (PT)rp[ce[g]] + s_bp[e];   // (owner, column) is tight: the owner's row dual
#pragma unroll
                    for (int q = 0; q < CH; q++) {
                        const int ch = min(q * T + tid, nchunks - 1);
                        cv[g][q] = *reinterpret_cast<const uint4 *>(rp + (size_t)ch * E);
                    }
                }
#pragma unroll
                for (int g = 0; g < SAPB_G; g++) {
                    // label through this row: h = de + (c + p - wst). The test h < d is done as
                    // (de - wst) + c < d - p: the right side is per column and changes only on the
                    // rare improvement, so the common case costs one 64-bit add and one compare.
                    const PT bs = de[g] - wst[g];
#pragma unroll
                    for (int q = 0; q < CH; q++) {
                        uint32_t c[E];
                        unpack<CT>(cv[g][q], c);
#if TD_RELAX_ANY
                        bool any = false;
#pragma unroll
                        for (int x = 0; x < E; x++) any = any || (bs + (PT)c[x] < dp[q * E + x]);
                        if (any)
#endif
                        {
#pragma unroll
                            for (int x = 0; x < E; x++) {
                                const int k = q * E + x;
                                const PT t = bs + (PT)c[x];
                                if (t < dp[k]) {
                                    const PT p = PREG ? preg[k] : P[(q * T + tid) * E + x];
                                    dp[k] = t;
                                    d[k] = t + p;
                                    PRED[(q * T + tid) * E + x] = ce[g];
                                    scanned &= ~(1ull << k);   // (re)opened
                                }
                            }
                        }
                    }
                }
            }
        }
        if (SPEC) {
            // record: the columns whose label is final and below the end distance (exactly the set
            // the dual update would touch), the end column, the path
            PsRec<PT> *rec = recs + blockIdx.x;
            unsigned long long fin = 0ull;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const bool in = ((scanned >> k) & 1ull) && !((padmask >> k) & 1ull) && d[k] < mind;
                fin |= in ? (1ull << k) : 0ull;
            }
            const int mycnt = (endcol >= 0) ? __popcll(fin) : 0;
            int incl = mycnt;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int v = __shfl_up(incl, sh);
                if (lane >= sh) incl += v;
            }
            __syncthreads();
            if (lane == 63) s_wcnt[w] = incl;
            __syncthreads();
            int base = 0, total = 0;
            for (int q = 0; q < nw; q++) {
                const int cw = s_wcnt[q];
                base += (q < w) ? cw : 0;
                total += cw;
            }
            const bool fits = endcol >= 0 && total <= PS_CAP;
            if (fits) {
                int pos = base + incl - mycnt;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if ((fin >> k) & 1ull) {
                        rec->S_col[pos] = ((k / E) * T + tid) * E + (k % E);
                        rec->S_d[pos] = d[k];
                        pos++;
                    }
                }
            }
            if (tid == 0) {
                int status = fits ? 1 : 0, plen = 0;
                if (status) {
                    int jc = endcol;
                    while (jc >= 0 && plen <= PS_CAP) {
                        rec->path[plen++] = jc;
                        jc = PRED[jc];
                    }
                    if (jc >= 0) status = 0;
                }
                rec->f = f;
                rec->endcol = endcol;
                rec->nS = fits ? total : 0;
                rec->plen = plen;
                rec->mind = mind;
                rec->status = status;
            }
            return;
        }
        if (endcol < 0) {
            bad = true;
            break;
        }
        // dual update on scanned columns: price += mind - d
        const unsigned long long upd = scanned & ~padmask;
#pragma unroll
        for (int q = 0; q < CH; q++) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (((upd >> (q * E + e)) & 1ull) && d[q * E + e] < mind) {
                    const int j = (q * T + tid) * E + e;
                    if (PREG)
                        preg[q * E + e] += mind - d[q * E + e];
                    else
                        P[j] = P[j] + (mind - d[q * E + e]);
                    if constexpr (IsNP<CT>::value) {   // narrow-price mode: a price at the limit ends the attempt
                        if ((PREG ? preg[q * E + e] : P[j]) >= (PT)NP_PLIMIT) atomicOr(&ctl[CTL_FLAG], 8);
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {  // flip the path: column j goes to the (old) owner of its predecessor column
            int j = endcol;
            for (int hop = 0; hop <= n; hop++) {
                const int pc = PRED[j];
                const int i = (pc < 0) ? f : OWN[pc];
                OWN[j] = i;
                r2c[i] = j;   // store only: no global load on the walk
                if (pc < 0) break;
                j = pc;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (PREG) {
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int ch = q * T + tid;
            if (ch < nchunks) {
#pragma unroll
                for (int e = 0; e < E; e++) P[ch * E + e] = preg[q * E + e];
            }
        }
        __syncthreads();
    }
    for (int j = tid; j < npad; j += T) {
        pk[j] = (PT)(P[j] << 1) | (PT)1;
        if (LDSST && j < n) owner_g[j] = OWN[j];
    }
    if (tid == 0) {
        ctl[CTL_NFREE] = nfree;
        ctl[CTL_STEPS] = (int)(steps > INT_MAX ? INT_MAX : steps);
        if (bad) atomicOr(&ctl[CTL_ERR], 2);
    }
}

// =====================================================================================
// k_sapx: the serial finisher spread over K workgroups (one search at a time, columns split).
//
// One workgroup cannot stream rows faster than one CU's load path: at N = 16 384 a batched step of
// k_sap moves 1 MiB through a single CU (~38 us). Here workgroup g owns the column chunks
// [g*T, (g+1)*T); every step each workgroup publishes up to SX_WL of its open owned columns (its
// local minimum first, then the ones inside the window above the last global minimum), ONE grid
// barrier, then every workgroup relaxes ITS columns against all published rows. The search is the
// same label-correcting scheme as k_sap (any scan order is valid; a column whose label drops is
// re-opened), it ends when no workgroup has an open owned column below the smallest free-column
// label — plain Dijkstra termination, so labels below the end distance are exact.
// The barrier is a monotonic 64-bit counter in global memory, release/acquire at agent scope,
// with a spin limit: a workgroup that waits too long raises `abort`, every workgroup leaves and
// the host reports TD_EINTERNAL (no unbounded spin). K <= 64 workgroups of <= 256 threads are
// always co-resident on the 256 CUs (the stream is serial: nothing else runs).
// =====================================================================================
constexpr int SX_WL = 8;       // columns a workgroup may publish per step
constexpr int SX_KMAX = 64;    // workgroups
#ifndef TD_SX_G
#define TD_SX_G 8
#endif
#ifndef TD_SX_NG
#define TD_SX_NG 2   // row groups per workgroup of k_sapx for 4-byte cells (measured: 1 -> 1125 ms, 2 -> 1033 ms, 4 -> 1237 ms on the |a-b| N = 16384 instance)
#endif
constexpr int SX_G = TD_SX_G;  // rows in flight per relax group
struct SxSlot {
    long long flabel;   // smallest label among this workgroup's free columns
    int fcol;
    int cnt;            // published entries
};
struct SxEnt {
    long long d, p;
    int col, own;
};
struct SxShared {
    unsigned long long bar;
    int abort;
    int pad;
    long long dbg[6];   // workgroup 0: cycles in barrier wait, selection, publish read, relax; rows published; steps (TD_DEBUG)
    SxSlot slot[2][SX_KMAX];
    SxEnt ent[2][SX_KMAX][SX_WL];
};

// NG row groups: the workgroup has NG * TX threads; thread (grp, tid) owns column chunk wg * TX + tid like
// before, all groups hold the same per-column state, and in the relax phase group g takes every NG-th block
// of published rows — NG waves per SIMD instead of one hide the latency of the row loads and of the LDS
// reads (the relax phase was 61 % of a step with one wave per SIMD).  The groups' labels are merged
// through LDS at the end of the step (minimum; ties -> lowest group), so the state stays replicated.
template <typename CT, int TX, int NG>
__global__ __launch_bounds__(TX * NG) void k_sapx(int n, int nchunks, const ShardTab tab, typename Tr<CT>::PT *__restrict__ pk,
                                             int *owner_g, int *__restrict__ r2c, int *pred_g,
                                             const int *__restrict__ list, int *__restrict__ ctl, SxShared *sh)
{
    using PT = typename Tr<CT>::PT;
    // labels in the width of the prices: 32-bit for u8 rows and the narrow-price mode (prices < 2^28
    // there), which halves the ALU work of the relax loop
    typedef typename std::conditional<sizeof(PT) == 4, int, long long>::type LT;
    constexpr int E = Tr<CT>::E;
    constexpr int NW = TX / 64;
    constexpr int SXG = NG >= 4 ? SX_G / 2 : SX_G;   // rows in flight per thread: the register budget halves at 1024 threads
    constexpr LT LMAX = sizeof(LT) == 4 ? (LT)(1 << 30) : (LT)((long long)1 << 62);
    __shared__ LT s_k[NW], s_p[NW], s_f[NW];
    __shared__ int s_j[NW], s_o[NW], s_fj[NW], s_wcnt[NW];
    __shared__ SxSlot s_slot[SX_KMAX];
    // per published column, computed once per workgroup: label minus the owner's row dual, the
    // owner's row (this workgroup's segment), the column
    __shared__ LT s_bs[SX_KMAX * SX_WL];
    __shared__ const CT *s_rp[SX_KMAX * SX_WL];
    __shared__ int s_ce[SX_KMAX * SX_WL];
    __shared__ int s_off[64];
    __shared__ int s_ok;
    __shared__ LT s_cd[NG > 1 ? NG * TX * E : 1];    // merge of the row groups' labels
    __shared__ int s_cc[NG > 1 ? NG * TX * E : 1];   // ... and of the predecessor columns
    if (ctl[CTL_FLAG]) return;
    const int nfree = ctl[CTL_NFREE];
    if (nfree <= 0) return;
    const int K = gridDim.x, wg = blockIdx.x, grp = threadIdx.x / TX, tid = threadIdx.x % TX, lane = tid & 63, w = tid >> 6;
    const bool g0 = grp == 0;   // the group that writes shared / global state
    const int npad = nchunks * E;
    const size_t pitch = (size_t)npad;
    const int ch = wg * TX + tid;
    const bool has = ch < nchunks;
    const int jbase = ch * E;
    uint32_t valid = 0;
    LT preg[E];
    int ownr[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const bool ok = has && jbase + e < n;
        valid |= ok ? (1u << e) : 0u;
        preg[e] = has ? (LT)(pk[jbase + e] >> 1) : 0;   // pad columns keep their (huge) price
        ownr[e] = ok ? owner_g[jbase + e] : -2;
    }
    // cost of every owned column in its owner's row: with the price it is the owner's row dual, which a
    // published column carries along (the consumers then need no second, dependent load per entry)
    LT cown[E];
#pragma unroll
    for (int e = 0; e < E; e++) cown[e] = (ownr[e] >= 0) ? (LT)shard_row<CT>(tab, ownr[e], pitch)[jbase + e] : 0;
    unsigned long long epoch = 0;
    // grid barrier; false = some workgroup gave up (abort raised)
    auto grid_sync = [&]() -> bool {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(&sh->bar, 1ull);
            const unsigned long long target = (epoch + 1ull) * (unsigned long long)K;
            int ok = 1;
            for (long long spins = 0;; spins++) {
                if (__hip_atomic_load(&sh->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                if (spins > 2000000ll || __hip_atomic_load(&sh->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&sh->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_ok = ok;
        }
        epoch++;
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return s_ok != 0;
    };
    long long steps = 0;
    LT delta = 0;
    bool bad = false;
    for (int fi = 0; fi < nfree && !bad; fi++) {
        const int f = list[fi];
        LT d[E], dp[E];
        uint32_t scanned = ~valid, owned = ~valid;
        {
            uint32_t c[E];
            if (has) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(shard_row<CT>(tab, f, pitch) + (size_t)jbase);
                unpack<CT>(cv, c);
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                const bool ok = (valid >> e) & 1u;
                d[e] = ok ? (LT)c[e] + preg[e] : LMAX;
                dp[e] = ok ? (LT)c[e] : -LMAX;
                if (ok && g0) pred_g[jbase + e] = -1;
                if (ownr[e] != -1) owned |= 1u << e;
            }
        }
        LT F = LMAX, base = LMAX, mind = 0;   // stale global values from the last barrier
        int endcol = -1, par = 0;
        bool first = true, done = false;
        for (int guard = 0; guard <= npad + 8 && !done; guard++) {
            const long long tc0 = clock64();
            // ---- local selection: smallest open owned column below F, smallest free label
            LT bk = LMAX, bp = 0, fk = LMAX;
            int bj = INT_MAX, bo = -2, fj = INT_MAX;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const bool open = !((scanned >> e) & 1u);
                const bool ow = (owned >> e) & 1u;
                if (open && ow && d[e] < F && d[e] < bk) {
                    bk = d[e];
                    bj = jbase + e;
                    bo = ownr[e];
                    bp = cown[e] + preg[e];
                }
                if (((valid >> e) & 1u) && !ow && d[e] < fk) {
                    fk = d[e];
                    fj = jbase + e;
                }
            }
            wave_argmin<LT>(bk, bj, bo, bp);
            {
                int o2 = 0;
                LT p2 = 0;
                wave_argmin<LT>(fk, fj, o2, p2);
            }
            if (g0 && lane == 0) {
                s_k[w] = bk;
                s_j[w] = bj;
                s_o[w] = bo;
                s_p[w] = bp;
                s_f[w] = fk;
                s_fj[w] = fj;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NW; q++) {
                if (s_k[q] < bk || (s_k[q] == bk && s_j[q] < bj)) {
                    bk = s_k[q];
                    bj = s_j[q];
                    bo = s_o[q];
                    bp = s_p[q];
                }
                if (s_f[q] < fk || (s_f[q] == fk && s_fj[q] < fj)) {
                    fk = s_f[q];
                    fj = s_fj[q];
                }
            }
            // candidates besides the local minimum: open owned columns below F inside the window
            uint32_t cand = 0;
            const LT lim = (first || base >= LMAX - delta) ? -1 : base + delta;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const bool ok = !((scanned >> e) & 1u) && ((owned >> e) & 1u) && d[e] < F && d[e] <= lim && (jbase + e) != bj;
                cand |= ok ? (1u << e) : 0u;
            }
            const int mycnt = __popc(cand);
            int incl = mycnt;
#pragma unroll
            for (int shf = 1; shf < 64; shf <<= 1) {
                const int v = __shfl_up(incl, shf);
                if (lane >= shf) incl += v;
            }
            __syncthreads();   // s_k.. consumed
            if (g0 && lane == 63) s_wcnt[w] = incl;
            __syncthreads();
            const int head = (bj != INT_MAX) ? 1 : 0;
            int pos = head, tot = head;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                pos += (q < w) ? s_wcnt[q] : 0;
                tot += s_wcnt[q];
            }
            pos += incl - mycnt;
            SxEnt *ge = sh->ent[par][wg];
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (((cand >> e) & 1u) && pos < SX_WL) {
                    if (g0) {
                        ge[pos].d = d[e];
                        ge[pos].p = cown[e] + preg[e];   // the owner's row dual
                        ge[pos].col = jbase + e;
                        ge[pos].own = ownr[e];
                    }
                    scanned |= 1u << e;
                    pos++;
                }
            }
            if (head && bj >= jbase && bj < jbase + E) scanned |= 1u << (bj - jbase);
            if (g0 && tid == 0) {
                if (head) {
                    ge[0].d = bk;
                    ge[0].p = bp;
                    ge[0].col = bj;
                    ge[0].own = bo;
                }
                SxSlot sl;
                sl.flabel = fk;
                sl.fcol = fj;
                sl.cnt = min(tot, SX_WL);
                sh->slot[par][wg] = sl;
            }
            const long long tc1 = clock64();
            if (!grid_sync()) {
                bad = true;
                break;
            }
            const long long tc2 = clock64();
            // ---- everybody reads the published step
            if (g0 && tid < K) s_slot[tid] = sh->slot[par][tid];
            __syncthreads();
            int B = 0;
            LT nF = LMAX;
            int nFj = INT_MAX;
            for (int q = 0; q < K; q++) {
                const SxSlot sl = s_slot[q];
                if (sl.flabel < nF || (sl.flabel == nF && sl.fcol < nFj)) {
                    nF = sl.flabel;
                    nFj = sl.fcol;
                }
                B += sl.cnt;
            }
            F = nF;
            if (B == 0) {   // no open owned column below the smallest free label: Dijkstra is done
                if (nFj == INT_MAX) bad = true;
                mind = nF;
                endcol = nFj;
                done = true;
                break;
            }
            steps++;
            LT nbase = LMAX;
            {
                // offsets of the workgroups' entries in the compact list: one wave scan over the counts
                if (g0 && tid < 64) {
                    const int cq = (tid < K) ? s_slot[tid].cnt : 0;
                    int inc = cq;
#pragma unroll
                    for (int shf = 1; shf < 64; shf <<= 1) {
                        const int v = __shfl_up(inc, shf);
                        if (lane >= shf) inc += v;
                    }
                    s_off[tid] = inc - cq;
                }
                __syncthreads();
                // all entry loads of a thread are issued together: one memory round trip per step
                // however many entries a thread has to fetch (the entries carry the owner's row dual)
                constexpr int NIT = (SX_KMAX * SX_WL + TX - 1) / TX;
                SxEnt en[NIT];
                bool ok[NIT];
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int t = it * TX + tid;
                    const int q = t / SX_WL, i = t - q * SX_WL;
                    ok[it] = g0 && q < K && i < s_slot[q < K ? q : 0].cnt;
                    if (ok[it]) en[it] = sh->ent[par][q][i];
                }
                const CT *rps[NIT];
#pragma unroll
                for (int it = 0; it < NIT; it++)
                    if (ok[it]) rps[it] = shard_row<CT>(tab, en[it].own, pitch);
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    if (ok[it]) {
                        const int t = it * TX + tid;
                        const int q = t / SX_WL, i = t - q * SX_WL;
                        const int at = s_off[q] + i;
                        s_bs[at] = en[it].d - en[it].p;   // label minus the owner's row dual
                        s_rp[at] = rps[it] + (size_t)wg * TX * E;
                        s_ce[at] = en[it].col;
                        nbase = en[it].d < nbase ? en[it].d : nbase;
                    }
                }
            }
#pragma unroll
            for (int shf = 32; shf > 0; shf >>= 1) {
                const LT o = __shfl_xor(nbase, shf);
                nbase = o < nbase ? o : nbase;
            }
            if (g0 && lane == 0) s_k[w] = nbase;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NW; q++) nbase = s_k[q] < nbase ? s_k[q] : nbase;
            const long long tc3 = clock64();
            const int lofs = has ? tid * E : 0;
            // The row pointers come out of LDS, which would make these generic (flat) loads: flat
            // loads also count on lgkmcnt, so every later LDS read would wait for all of them.
            // Casting to the global address space keeps them on vmcnt alone, and the loads of the
            // next group are issued before the current group is relaxed (double buffer).
            typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
            typedef const v4u_t __attribute__((address_space(1))) *gvec_t;
            // row group `grp` relaxes the blocks grp, grp + NG, ... of SXG published rows
            LT dp0[E];
            int pc[E];
#pragma unroll
            for (int x = 0; x < E; x++) {
                dp0[x] = dp[x];
                pc[x] = -1;
            }
            v4u_t cvn[SXG];
#pragma unroll
            for (int g = 0; g < SXG; g++)
                cvn[g] = *(gvec_t)(uintptr_t)(s_rp[min(grp * SXG + g, B - 1)] + lofs);
            for (int e0 = grp * SXG; e0 < B; e0 += NG * SXG) {
                uint4 cv[SXG];
#pragma unroll
                for (int g = 0; g < SXG; g++) cv[g] = make_uint4(cvn[g].x, cvn[g].y, cvn[g].z, cvn[g].w);
                if (e0 + NG * SXG < B) {
#pragma unroll
                    for (int g = 0; g < SXG; g++)
                        cvn[g] = *(gvec_t)(uintptr_t)(s_rp[min(e0 + NG * SXG + g, B - 1)] + lofs);
                }
#pragma unroll
                for (int g = 0; g < SXG; g++) {
                    const LT bs = s_bs[min(e0 + g, B - 1)];
                    uint32_t c[E];
                    unpack<CT>(cv[g], c);
                    bool any = false;
#pragma unroll
                    for (int x = 0; x < E; x++) any = any || (bs + (LT)c[x] < dp[x]);
                    if (any) {
                        const int cg = s_ce[min(e0 + g, B - 1)];
#pragma unroll
                        for (int x = 0; x < E; x++) {
                            const LT t = bs + (LT)c[x];
                            if (t < dp[x]) {
                                dp[x] = t;
                                pc[x] = cg;
                            }
                        }
                    }
                }
            }
            if constexpr (NG > 1) {
                // merge the groups' labels: minimum, ties -> lowest group; every group ends with the same state
#pragma unroll
                for (int x = 0; x < E; x++) {
                    s_cd[(grp * TX + tid) * E + x] = dp[x];
                    s_cc[(grp * TX + tid) * E + x] = pc[x];
                }
                __syncthreads();
#pragma unroll
                for (int x = 0; x < E; x++) {
                    LT best = dp0[x];
                    int bc = -1;
#pragma unroll
                    for (int q = 0; q < NG; q++) {
                        const LT v = s_cd[(q * TX + tid) * E + x];
                        if (v < best) {
                            best = v;
                            bc = s_cc[(q * TX + tid) * E + x];
                        }
                    }
                    dp[x] = best;
                    pc[x] = bc;
                }
            }
#pragma unroll
            for (int x = 0; x < E; x++) {
                if (pc[x] >= 0) {   // improved in this step: (re)opened
                    d[x] = dp[x] + preg[x];
                    scanned &= ~(1u << x);
                    if (g0) pred_g[jbase + x] = pc[x];
                }
            }
            if (wg == 0 && g0 && tid == 0) {
                const long long tc4 = clock64();
                sh->dbg[0] += tc2 - tc1;
                sh->dbg[1] += tc1 - tc0;
                sh->dbg[2] += tc3 - tc2;
                sh->dbg[3] += tc4 - tc3;
                sh->dbg[4] += B;
                sh->dbg[5] += 1;
            }
            base = nbase;
            first = false;
            // window control: aim at about 4 published columns per workgroup
            if (B > 4 * K)
                delta >>= 1;
            else if (B <= 2 * K)
                delta = (delta < (LMAX >> 8)) ? delta * 2 + 1 : delta;
            par ^= 1;
        }
        if (bad || !done || endcol < 0) {
            bad = true;
            break;
        }
        // dual update on columns whose (exact) label is below the end distance
#pragma unroll
        for (int e = 0; e < E; e++)
            if (((scanned >> e) & 1u) && ((valid >> e) & 1u) && d[e] < mind) {
                preg[e] += mind - d[e];
                if constexpr (IsNP<CT>::value) {
                    if (preg[e] >= (LT)NP_PLIMIT) atomicOr(&ctl[CTL_FLAG], 8);
                }
            }
        if (wg == 0 && g0 && tid == 0) {   // flip the path (pred / owner are global; the barrier made them visible)
            int j = endcol;
            for (int hop = 0; hop <= n; hop++) {
                const int pc = pred_g[j];
                const int i = (pc < 0) ? f : owner_g[pc];
                owner_g[j] = i;
                r2c[i] = j;
                if (pc < 0) break;
                j = pc;
            }
        }
        if (!grid_sync()) {
            bad = true;
            break;
        }
#pragma unroll
        for (int e = 0; e < E; e++)
            if ((valid >> e) & 1u) {
                const int no = owner_g[jbase + e];
                if (no != ownr[e]) {   // a column of the augmenting path
                    ownr[e] = no;
                    cown[e] = (no >= 0) ? (LT)shard_row<CT>(tab, no, pitch)[jbase + e] : 0;
                }
            }
    }
#pragma unroll
    for (int e = 0; e < E; e++)
        if (has && g0) pk[jbase + e] = (PT)((PT)preg[e] << 1) | (PT)1;
    if (wg == 0 && g0 && tid == 0) {
        ctl[CTL_NFREE] = nfree;
        ctl[CTL_STEPS] = (int)(steps > INT_MAX ? INT_MAX : steps);
        if (bad) atomicOr(&ctl[CTL_ERR], 16);
    }
}

#include "td_forest.h"

// =====================================================================================
// k_sap8: the finisher specialised for u8 rows / int32 prices with ONE 16-column chunk per
// thread (n <= 16 384).  Same algorithm as k_sap; the step loop is cut to ~1/3 of the VALU
// instructions (it is issue-bound: 16 waves share 4 SIMDs):
//   * per column one packed argmin key am = dist<<5 | owned<<4 | slot, so the local argmin is 8
//     v_min3_u32 and the winner's slot falls out of the key;
//   * a scanned column gets am = MAX, dist = 0 and its pending dual change folded into the
//     cached price at scan time, so relax needs no "scanned" test (Dijkstra never improves a
//     finalised column) and the dual update is price + mind;
//   * pred[] lives in registers and is flushed to LDS once per search (4 ds_write_b128).
// =====================================================================================
constexpr int SAP_W = 16;  // columns of one tie class finalised per step
constexpr int SAP_G = 4;   // rows streamed per load group

// SPEC = false: the serial finisher (applies every augmentation itself).
// SPEC = true : one speculative search per workgroup against a read-only snapshot; the search is
//               recorded in recs[blockIdx.x] and applied (or rejected) by k_pcommit.
template <bool LDSST, bool SPEC>
__global__ __launch_bounds__(1024) void k_sap8(int n, int nchunks, const ShardTab tab, int32_t *__restrict__ pk,
                                               int *__restrict__ owner_g, int *__restrict__ r2c,
                                               int *__restrict__ pred_g, int *__restrict__ list,
                                               int *__restrict__ ctl, PsRec<int32_t> *__restrict__ recs,
                                               int place_const = 0 /* serial form: also place the deferred constant rows */)
{
    constexpr int E = 16;
    constexpr uint32_t AMAX = 0x7FFFFFFFu;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_rk[2][16];
    __shared__ int s_rc[2][16];
    __shared__ int s_bj[2][SAP_W];
    __shared__ int s_bp[2][SAP_W];
    __shared__ int s_rp[2][16];
    __shared__ int s_rj[2][16];
    __shared__ int s_wcnt[16];
    __shared__ int s_nfree;
    __shared__ int s_end[2];

    if (ctl[CTL_FLAG]) return;
    if (SPEC && (int)blockIdx.x >= ctl[CTL_NFREE]) return;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    const int npad = nchunks * E;
    const size_t pitch = (size_t)npad;
    int *OWN = LDSST ? reinterpret_cast<int *>(smem) : owner_g;
    int *PRED = LDSST ? reinterpret_cast<int *>(smem + (size_t)npad * sizeof(int)) : pred_g;

    // free rows: ordered list built by k_freelist / the last k_pcommit
    const int nfree = SPEC ? 1 : ctl[CTL_NFREE];
    if (!SPEC && nfree == 0) {
        if (tid == 0) ctl[CTL_STEPS] = 0;
        if (place_const) place_const_tail(n, r2c, owner_g, list, pred_g, ctl);
        return;
    }
    if (LDSST)
        for (int j = tid; j < npad; j += T) OWN[j] = owner_g[j];
    if (tid == 0) s_end[0] = s_end[1] = INT_MAX;
    __syncthreads();

    const bool has = tid < nchunks;
    const int jbase = tid * E;
    long long steps = 0;
    int par = 0;
    bool bad = false;
    PsRec<int32_t> *rec = SPEC ? recs + blockIdx.x : nullptr;
    for (int fi = 0; fi < nfree && !bad; fi++) {
        int f, rott = 0;
        if (SPEC) {
            // searches of a batch are spread evenly over the (row-ordered) free list, and ties
            // between free end columns are broken in a thread order rotated per search
            const int nf = ctl[CTL_NFREE];
            const int gact = nf < (int)gridDim.x ? nf : (int)gridDim.x;
            f = list[(int)(((long long)blockIdx.x * nf) / gact)];
            rott = (int)(((uint64_t)(((uint32_t)f + 1u) * 0x9E3779B1u) * (uint64_t)T) >> 32);
        } else {
            f = list[fi];
        }
        // key[e] = (dist+1) << 5 | owned << 4 | e ; a finalised or non-existent column keeps only
        // its low 5 bits: it can never be improved (any candidate key is >= it) and, seen through
        // "key - 32" (unsigned wrap), never wins the argmin.
        uint32_t key[E];
        int32_t preg[E], predr[E];
        uint32_t sc = 0;
        {
            uint32_t c[E];
            if (has) {
                const uint4 cv = *reinterpret_cast<const uint4 *>(shard_row<uint8_t>(tab, f, pitch) + (size_t)jbase);
                unpack<uint8_t>(cv, c);
            }
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int j = jbase + e;
                const bool valid = has && j < n;
                const int32_t p = valid ? (pk[j] >> 1) : 0;
                const int o = valid ? OWN[j] : -2;
                preg[e] = p;
                predr[e] = SPEC ? -1 : f;
                const uint32_t low = ((o != -1) ? 16u : 0u) | (uint32_t)e;
                key[e] = valid ? (((c[e] + (uint32_t)p + 1u) << 5) | low) : low;
            }
        }
        int32_t mind = 0;
        int endcol = -1;
        for (int guard = 0; guard <= npad; guard++) {
            uint32_t m0 = min(min(key[0] - 32u, key[1] - 32u), min(key[2] - 32u, key[3] - 32u));
            uint32_t m1 = min(min(key[4] - 32u, key[5] - 32u), min(key[6] - 32u, key[7] - 32u));
            uint32_t m2 = min(min(key[8] - 32u, key[9] - 32u), min(key[10] - 32u, key[11] - 32u));
            uint32_t m3 = min(min(key[12] - 32u, key[13] - 32u), min(key[14] - 32u, key[15] - 32u));
            const uint32_t m = min(min(m0, m1), min(m2, m3));
            // wave stage: min key, and how many lanes sit at the same DISTANCE as the wave minimum
            const uint32_t mw = wave_umin32(m);
            const unsigned long long tieb = __ballot((m >> 5) == (mw >> 5));
            uint32_t mg = mw;
            int total = __popcll(tieb), base = 0;
            if (nw > 1) {
                if (lane == 0) {
                    s_rk[par][w] = mw;
                    s_rc[par][w] = total;
                }
                __syncthreads();
                const bool hv = lane < nw;
                const uint32_t k2 = hv ? s_rk[par][lane] : 0xFFFFFFFFu;
                const int c2 = hv ? s_rc[par][lane] : 0;
                mg = wave_umin32(k2);
                const unsigned long long wb = __ballot(hv && (k2 >> 5) == (mg >> 5));  // waves at the global distance
                // ordered prefix of the tie counts of those waves
                total = 0;
                base = 0;
                unsigned long long rest = wb;
                while (rest) {
                    const int ww = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int cw = __builtin_amdgcn_readlane(c2, ww);
                    base += (ww < w) ? cw : 0;
                    total += cw;
                }
            }
            if (mg >= 0xF0000000u) {  // nothing left to scan: cannot happen with a free column around
                bad = true;
                break;
            }
            const int32_t bd = (int32_t)(mg >> 5);
            const bool mine_d = (m >> 5) == (uint32_t)bd;  // this lane holds a column at the frontier distance
            if (!(mg & 16u)) {  // a FREE column is at the frontier distance: the search ends there
                // several threads can hold a column with this key: the first in (rotated) thread
                // order wins — deterministic, and different for different speculative searches
                if (m == mg) {
                    int rt = tid - rott;
                    rt += (rt < 0) ? T : 0;
                    atomicMin(&s_end[par], rt);
                }
                __syncthreads();
                int wt = s_end[par] + rott;
                wt -= (wt >= T) ? T : 0;
                endcol = wt * E + (int)(mg & 15u);
                mind = bd;
                if (tid == 0) s_end[par ^ 1] = INT_MAX;
                par ^= 1;
                break;
            }
            // every column at the frontier distance is owned: finalise up to SAP_W of them in this
            // step (ordered by thread) — one memory round trip serves the whole tie class
            const int nb = total < SAP_W ? total : SAP_W;
            if (SPEC && steps + nb > PS_CAP) {  // too long for a speculative record
                bad = true;
                break;
            }
            const unsigned long long myb = __ballot(mine_d);
            const int rank = base + __popcll(myb & ((1ull << lane) - 1ull));
            const bool sel = mine_d && rank < SAP_W;
            {
                int pe = preg[0];
#pragma unroll
                for (int e = 1; e < E; e++) pe = ((m & 15u) == (uint32_t)e) ? preg[e] : pe;
                if (sel) {
                    s_bj[par][rank] = jbase + (int)(m & 15u);
                    s_bp[par][rank] = pe;
                    if (SPEC) {
                        rec->S_col[steps + rank] = jbase + (int)(m & 15u);
                        rec->S_d[steps + rank] = bd;
                    }
                }
                // finalise my column: pending dual change goes into preg, key keeps only its low bits
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const bool hit = sel && ((m & 15u) == (uint32_t)e);
                    preg[e] = hit ? preg[e] - bd : preg[e];
                    key[e] = hit ? (key[e] & 31u) : key[e];
                }
                sc |= sel ? (1u << (m & 15u)) : 0u;
            }
            if (tid == 0) s_end[par ^ 1] = INT_MAX;
            __syncthreads();
            steps += nb;
            // stream the owners' rows of the batch in groups of SAP_G (that many 16-byte loads in
            // flight per lane), then relax
            const int lpar = par;
            par ^= 1;
#pragma unroll 1
            for (int g = 0; g < nb; g += SAP_G) {
                uint4 cv[SAP_G];
                int32_t t1[SAP_G];
                int ob[SAP_G], cs[SAP_G];
#pragma unroll
                for (int q = 0; q < SAP_G; q++) {
                    const int qq = (g + q < nb) ? g + q : g;   // clamp: a duplicate relax is harmless
                    // wave-uniform values: keep them in SGPRs (scalar addressing, no VGPR cost)
                    const int bj = __builtin_amdgcn_readfirstlane(s_bj[lpar][qq]);
                    const int bp = __builtin_amdgcn_readfirstlane(s_bp[lpar][qq]);
                    const int o = __builtin_amdgcn_readfirstlane(OWN[bj]);
                    ob[q] = SPEC ? bj : o;   // predecessor: column (speculative record) or row
                    const uint8_t *rp = shard_row<uint8_t>(tab, o, pitch);
                    cs[q] = (int)rp[bj];       // consumed after ALL loads of the group are issued
                    t1[q] = bd - bp + 1;       // dist - row dual of o (+1: key bias), minus cs[q] below
                    if (has) cv[q] = *reinterpret_cast<const uint4 *>(rp + (size_t)jbase);
                }
                if (has) {
#pragma unroll
                    for (int q = 0; q < SAP_G; q++) {
                        uint32_t c[E];
                        unpack<uint8_t>(cv[q], c);
                        const int32_t tq = t1[q] - __builtin_amdgcn_readfirstlane(cs[q]);
#pragma unroll
                        for (int e = 0; e < E; e++) {
                            const uint32_t h1 = (uint32_t)((int32_t)c[e] + preg[e] + tq);
                            const uint32_t hk = (h1 << 5) | (key[e] & 31u);
                            const bool better = hk < key[e];
                            key[e] = better ? hk : key[e];
                            predr[e] = better ? ob[q] : predr[e];
                        }
                    }
                }
            }
        }
        if (endcol < 0) {
            bad = true;
            break;
        }
        if (has) {
            // dual update of finalised columns (their -dist is already folded into preg) + pred flush
            if (!SPEC) {
#pragma unroll
                for (int e = 0; e < E; e++)
                    if ((sc >> e) & 1u) pk[jbase + e] = ((preg[e] + mind) << 1) | 1;
            }
#pragma unroll
            for (int e = 0; e < E; e += 4)
                *reinterpret_cast<int4 *>(&PRED[jbase + e]) = make_int4(predr[e], predr[e + 1], predr[e + 2], predr[e + 3]);
        }
        __syncthreads();
        if (tid == 0) {
            if (SPEC) {  // record the path as columns: end column first, predecessors after
                int plen = 0, j = endcol;
                while (j >= 0 && plen <= PS_CAP) {
                    rec->path[plen++] = j;
                    j = PRED[j];
                }
                rec->plen = plen;
                if (j >= 0) bad = true;
            } else {  // flip the path
                int j = endcol;
                for (int hop = 0; hop <= n; hop++) {
                    const int i = PRED[j];
                    OWN[j] = i;
                    if (LDSST) owner_g[j] = i;   // write-through: no epilogue copy
                    if (j == endcol) pk[j] = pk[j] | 1;
                    const int jn = r2c[i];
                    r2c[i] = j;
                    j = jn;
                    if (i == f) break;
                }
            }
        }
        __syncthreads();
        if (SPEC) {
            if (tid == 0) {
                rec->f = f;
                rec->endcol = endcol;
                rec->nS = (int)steps;
                rec->mind = mind;
                rec->status = bad ? 0 : 1;
            }
            return;
        }
    }
    if (SPEC) {  // search aborted (record too long / no candidate): the row stays free
        if (tid == 0) recs[blockIdx.x].status = 0;
        return;
    }
    __syncthreads();
    if (tid == 0) {
        ctl[CTL_NFREE] = nfree;
        ctl[CTL_STEPS] = (int)(steps > INT_MAX ? INT_MAX : steps);
        if (bad) atomicOr(&ctl[CTL_ERR], 2);
    }
    if (place_const && !bad) {   // owner_g / r2c were written through by thread 0
        __threadfence();
        __syncthreads();
        place_const_tail(n, r2c, owner_g, list, pred_g, ctl);
    }
}

// =====================================================================================
// Speculative PARALLEL shortest augmenting paths.
//
// The serial finisher leaves 255 CUs idle.  Here up to PS_G free rows are searched at the same
// time, one workgroup each, all against the SAME read-only snapshot (prices, owners).  A search
// records the columns it finalised (with their distances), its end column and its path.  The
// commit kernel then accepts a set of searches whose {finalised columns + end column} sets are
// pairwise disjoint (each column is claimed by the lowest search id with a min; a search is
// accepted iff it holds every column it touched) and applies their dual updates and path flips.
// Why disjointness is enough: an accepted search only raises prices of ITS finalised columns and
// only re-matches rows/columns on ITS path, so (a) the union of the flips is a matching and
// (b) every matched pair stays tight — a pair in search A's region is tight after A's own update
// (plain Dijkstra argument on the snapshot) and another search only raises OTHER columns' prices,
// which cannot lower that row's minimum.  Exact complementary slackness is preserved; rejected
// rows simply stay free for the next batch (search 0 of a batch is always accepted).
// Results do not depend on workgroup timing: searches of a batch share one snapshot and the
// claim is a min over ids.
// =====================================================================================
// commit of one batch + rebuild of the ordered free-row list.  One workgroup.
//
// Acceptance (deterministic, order = search id r):
//   touch[col] = min id over searches that finalised col or have it on their path
//   pthm[col]  = min id over searches that have col on their path
//   r is accepted  <=>  every path column of r has touch == r   (nobody else touches r's path)
//                  and  every finalised column of r has pthm >= r (r touches nobody's path...
//                       a higher id with that column on its path fails its own first test)
// Two accepted searches may share finalised-only columns; such a column is raised by the MAX of
// the two raises (mind - d): each search's Dijkstra guarantees reduced costs >= its own raise,
// so the max keeps every reduced cost >= 0, and columns on an accepted path are touched by that
// search alone, so its path stays exactly tight.
template <typename PT>
__global__ __launch_bounds__(1024) void k_pcommit(int n, PT *__restrict__ pk, int *__restrict__ owner, int *__restrict__ r2c,
                                                  int *__restrict__ list, int *__restrict__ ctl,
                                                  unsigned long long *__restrict__ raise,  // n words, all zero on entry/exit
                                                  const PsRec<PT> *__restrict__ recs, int first, int ngrid = PS_G,
                                                  long long plimit = 0 /* narrow-price mode: flag prices at or above */)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int *touch = reinterpret_cast<int *>(smem);  // n ints
    int *pthm = touch + n;                       // n ints
    __shared__ int s_wcnt[16];
    __shared__ int s_nfree;
    __shared__ int s_acc;
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
    const int nres = first ? 0 : min(ctl[CTL_NFREE], ngrid);
    if (!first && nres == 0) return;
    for (int j = tid; j < n; j += T) {
        touch[j] = INT_MAX;
        pthm[j] = INT_MAX;
    }
    if (tid == 0) s_acc = 0;
    __syncthreads();
    // phase 1: claims
    for (int r = w; r < nres; r += nw) {
        const PsRec<PT> *rc = recs + r;
        if (rc->status != 1) continue;
        for (int k = lane; k < rc->nS; k += 64) atomicMin(&touch[rc->S_col[k]], r);
        for (int k = lane; k < rc->plen; k += 64) {
            atomicMin(&touch[rc->path[k]], r);
            atomicMin(&pthm[rc->path[k]], r);
        }
    }
    __syncthreads();
    // phase 2: accepted searches post their raises and flip their path
    for (int r = w; r < nres; r += nw) {
        const PsRec<PT> *rc = recs + r;
        if (rc->status != 1) continue;
        bool ok = true;
        for (int k = lane; k < rc->plen; k += 64) ok = ok && (touch[rc->path[k]] == r);
        for (int k = lane; k < rc->nS; k += 64) ok = ok && (pthm[rc->S_col[k]] >= r);
        if (__ballot(!ok)) continue;
        const PT mind = rc->mind;
        for (int k = lane; k < rc->nS; k += 64) {
            const PT dv = mind - rc->S_d[k];
            if (dv > 0) atomicMax(&raise[rc->S_col[k]], (unsigned long long)dv);
        }
        if (lane == 0) {
            // path[0] = end column, path[i+1] = predecessor column of path[i]; the row that takes
            // path[i] is the (old) owner of path[i+1], and the root row takes the last one
            const int pl = rc->plen;
            for (int i = 0; i < pl; i++) {
                const int col = rc->path[i];
                const int row = (i + 1 < pl) ? owner[rc->path[i + 1]] : rc->f;
                owner[col] = row;
                r2c[row] = col;
            }
            atomicAdd(&s_acc, 1);
        }
    }
    __threadfence();
    __syncthreads();
    // phase 2b: apply the raises — max over the accepted searches, each column exactly once (the
    // first search to swap the slot to zero applies it); only touched columns are visited
    for (int r = w; r < nres; r += nw) {
        const PsRec<PT> *rc = recs + r;
        if (rc->status != 1) continue;
        bool ok = true;
        for (int k = lane; k < rc->plen; k += 64) ok = ok && (touch[rc->path[k]] == r);
        for (int k = lane; k < rc->nS; k += 64) ok = ok && (pthm[rc->S_col[k]] >= r);
        if (__ballot(!ok)) continue;
        for (int k = lane; k < rc->nS; k += 64) {
            const int col = rc->S_col[k];
            const unsigned long long rv = atomicExch(&raise[col], 0ull);
            if (rv) {
                if (plimit && (long long)(pk[col] >> 1) + (long long)rv >= plimit) atomicOr(&ctl[CTL_FLAG], 8);
                pk[col] = pk[col] + (PT)((PT)rv << 1);
            }
        }
        if (lane == 0) pk[rc->endcol] = pk[rc->endcol] | (PT)1;  // the end column has an owner now
    }
    __syncthreads();
    // phase 3: ordered list of the rows that are still free
    const int nleft = build_free_list(n, r2c, list);
    if (tid == 0) s_nfree = nleft;
    __syncthreads();
    if (tid == 0) {
        // nothing accepted (every search overflowed its record): later batches of this group exit
        // at once; the rows stay in the list for the serial finisher
        ctl[CTL_NFREE] = s_nfree;
        if (!first && s_acc == 0) ctl[CTL_PSTOP] = 1;
        ctl[CTL_PACC] += s_acc;
    }
}

// =====================================================================================
// k_final: total from the original costs + permutation check; k_dual: LP bound
// =====================================================================================
template <bool GEN = false>
__global__ __launch_bounds__(256) void k_final(int n, int nrows, int row0, const int32_t *__restrict__ cost,
                                               const int *__restrict__ r2c, const int *__restrict__ owner,
                                               unsigned long long *__restrict__ out, int *__restrict__ ctl, int cost_is_transposed = 0,
                                               const CellSrc src = CellSrc())
{
    if (ctl[CTL_FLAG]) return;
    long long s = 0;
    int bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
        const int j = r2c[i];
        if (j < 0 || j >= n || owner[j] != row0 + i)
            bad = 1;
        else   // the fused transposed solve never materialises the transposed int32 matrix: its cell (i, j) is the caller's (j, i)
            s += GEN ? (long long)(cost_is_transposed ? src.cell(j, i) : src.cell(i, j))
                     : (long long)(cost_is_transposed ? cost[(int64_t)j * n + i] : cost[(int64_t)i * n + j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        bad |= __shfl_xor(bad, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)s);
        if (bad) atomicOr(&ctl[CTL_ERR], 4);
    }
}

__global__ void k_pack_totals(const long long *__restrict__ tot2, const int *__restrict__ ctl, long long *__restrict__ out3)
{
    if (threadIdx.x == 0) {
        out3[0] = tot2[0];
        out3[1] = tot2[1];
        out3[2] = (long long)(ctl[CTL_ERR] != 0) + (long long)(ctl[CTL_FLAG] != 0);
    }
}

// local row_to_col from the replicated owner[] (after the finisher ran on another shard)
__global__ void k_r2c_from_owner(int n, int nrows, int row0, const int *__restrict__ owner, int *__restrict__ r2c)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const int o = owner[j];
        if (o >= row0 && o < row0 + nrows) r2c[o - row0] = j;
    }
}

// constant-row deferral over row shards: each rank marks its constant rows in a replicated mask (summed by the
// caller); the finisher's rank hides them from the searches (-2) and counts them for place_const_tail
__global__ void k_const_mask(int row0, int nrows, const int *__restrict__ rconst, int *__restrict__ mask_full)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows && rconst[i]) mask_full[row0 + i] = 1;
}

__global__ __launch_bounds__(1024) void k_mark_const(int n, const int *__restrict__ mask, int *__restrict__ r2c_full, int *__restrict__ ctl)
{
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (mask[i]) {
            r2c_full[i] = -2;
            cnt++;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) ctl[CTL_NCONST] = s_cnt;
}

// replicated prices <-> plain int64 (for the broadcast after the finisher changed them on one rank)
template <typename PT>
__global__ void k_price_io(int n, PT *pk, long long *plain, int set)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        if (set)
            pk[j] = (PT)((PT)plain[j] << 1) | (PT)1;
        else
            plain[j] = (long long)(pk[j] >> 1);
    }
}

__global__ void k_fill_i32(int *p, int count, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// dual bound D = sum_i (rowmin_i + min_j (c'_ij + p_j)) - sum_j p_j ; one wave per (local) row
template <typename CT>
__global__ __launch_bounds__(256) void k_dual(int n, int nrows, int row0, int nchunks, const CT *__restrict__ cc,
                                              const typename Tr<CT>::PT *__restrict__ pk,
                                              const int32_t *__restrict__ rowmin, unsigned long long *__restrict__ out)
{
    using PT = typename Tr<CT>::PT;
    constexpr int E = Tr<CT>::E;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t pitch = (size_t)nchunks * E;
    long long acc = 0;
    for (int row = blockIdx.x * nw + w; row < nrows; row += gridDim.x * nw) {
        const CT *rp = cc + (size_t)row * pitch;
        PT m = Tr<CT>::KMAX;
        for (int ch = lane; ch < nchunks; ch += 64) {
            const uint4 cv = *reinterpret_cast<const uint4 *>(rp + (size_t)ch * E);
            uint32_t c[E];
            unpack<CT>(cv, c);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const PT v = (PT)c[e] + (pk[(size_t)ch * E + e] >> 1);
                m = v < m ? v : m;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            PT om = shfl_xor_t(m, o);
            m = om < m ? om : m;
        }
        if (lane == 0) acc += (long long)m + (long long)rowmin[row];
    }
    // minus the sum of prices: once (block 0 of the shard that owns row 0)
    if (blockIdx.x == 0 && row0 == 0) {
        long long ps = 0;
        for (int j = threadIdx.x; j < n; j += blockDim.x) ps += (long long)(pk[j] >> 1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o);
        if (lane == 0) acc -= ps;
    }
    if (lane == 0 && acc != 0) atomicAdd(&out[1], (unsigned long long)acc);
}

// -------------------------------------------------------------------------------------
// host side: one Solver per cost matrix (td_assign) or per row shard (td_shard_*)
// -------------------------------------------------------------------------------------
}  // namespace

struct td_shard {
    int n = 0, row0 = 0, nrows = 0;
    int bpc = 0;  // bytes per stored cell: 1, 2, 4 (0 = not compressed yet)
    int64_t range = 0;      // largest row range of this shard, exact once a compress pass at a narrower width has failed
    int64_t fit_bound = -1;  // else: the limit of the narrowest width the rows have fitted so far
    int nchunks = 0, npad = 0;
    const int32_t *d_cost = nullptr;  // nrows x n, device
    Buf stage, cc, price, owner, r2c, r2c_full, bid, pred, list, rowmin, rconst, misc, psrec, tbuf, xbuf, fbuf;
    bool defer_const = false;  // constant rows sit out the solve and take the left-over columns (td_assign only)
    int nconst = -1;                 // constant rows counted by the last compress pass (-1: not read back)
    const int32_t *probe = nullptr;  // non-null for the one k_init_state launch that carries the shape probe
    bool placed = false;       // ... and the finisher kernel has already placed them (no k_place_const launch)
    int64_t range_seen = -1;   // largest row range measured by the last synchronous compress pass
    bool want_bid0 = false;    // td_assign: let the compress pass write round 0's bids (k_compress_reg<BID0>)
    bool bid0_done = false;    // ... it did: state init has run before the pass, round 0 launches no k_bid
    Buf cmask;                 // sharded solve: the replicated mask of constant rows (td_shard_const_rows)
    bool have_cmask = false;
    bool fused_t = false;      // cc holds the TRANSPOSED problem built straight from the caller's matrix (k_compress_tr): d_cost is not transposed
    bool fused8 = false;       // ... as 1-byte cells with the escape code (u8e, bpc code 6)
    const long long *skip = nullptr;  // device flag of a pending line-metric probe: non-zero makes the compress pass a no-op
    // block-local start (td_blocks.h)
    bool gen = false;          // td_build_assign: the model's cells are made from the position arrays (gsrc), no int32 matrix exists (yet)
    CellSrc gsrc;
    Buf genbuf;                // ... the matrix, should the solve have to leave the fused path
    Buf gpos;                  // td_build_assign's position arrays / distance table on the device
    bool tick_sized = false;   // td_tick's remainder (hinted, n < 2048): the rounds leave a dozen rows, the serial workgroup is through before a speculative batch + its commit are — no read-back of the free-row count to decide that
    int zs_V = 0;              // diagonal blocks of the whole matrix the 1-byte attempt may start in (0: off)
    bool zs_done = false;      // the compress pass prepared the block-local start (wrote the zero-slice bids of phase A's round 0, or left them to k_zs_bid: zs_round0): sv_phase_a is due
    bool lazy_cc = false;      // td_assign: the block-local start may leave the narrow copy with the diagonal slices only (k_compress_reg diag_only)
    bool cc_partial = false;   // ... and this attempt's copy IS partial: whoever needs whole rows calls sv_complete_cc first
    int cc_rpb = 0;
    bool zs_round0 = false;    // ... round 0 of phase A is still to be bid (rows of <= 16 384 columns: the plain compress pass + one k_zs_bid round over 1/8 of every row beat the pass that also scans for the bids, 216 + ~10 against 243 us)
    bool began = false;        // the state (prices, owners, row_to_col, bid keys) has been initialised for the current compressed copy
    bool state_ready = false;  // sharded solve: the state was initialised in front of the compress pass and phase A has run on it (td_shard_begin must not redo it)
    Buf ob, esc, hop, hoptab;  // owned bytes per column, escape masks per row, HopCtl, the two-hop tables
    Buf core, core_n, core_t, core_need;   // sparse core of the warm start (td_core_warm.h): lists, their lengths, the smallest value outside, the rows that need their dense row
    void free_all()
    {
        Buf *bs[] = {&stage, &cc, &price, &owner, &r2c, &r2c_full, &bid, &pred, &list, &rowmin, &rconst, &misc, &psrec, &tbuf, &xbuf, &fbuf, &cmask, &ob, &esc, &hop, &hoptab, &core, &core_n, &core_t, &core_need, &genbuf, &gpos};
        for (Buf *b : bs) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
    }
};

namespace {

using Solver = td_shard;
Solver g_default;  // td_assign's workspace, kept across calls
Solver *g_active = &g_default;   // the workspace the next td_assign / td_build_assign call solves in (a td_solver_* call switches it for its duration)

int sv_prepare(Solver &sv, int n, int row0, int nrows, const int32_t *cost)
{
    Ctx &c = ctx();
    sv.n = n;
    sv.row0 = row0;
    sv.nrows = nrows;
    sv.bpc = 0;
    int rc;
    const void *d;
    if ((rc = to_device(cost, sizeof(int32_t) * (size_t)nrows * n, sv.stage, &d))) return rc;
    sv.d_cost = (const int32_t *)d;
    const size_t np = (size_t)((n + 3) / 4) * 4 + 16;
    if ((rc = ensure(sv.misc, 4096))) return rc;
    if ((rc = ensure(sv.rowmin, sizeof(int32_t) * (size_t)std::max(nrows, 1)))) return rc;
    if ((rc = ensure(sv.rconst, sizeof(int) * (size_t)std::max(nrows, 1)))) return rc;
    if ((rc = ensure(sv.price, sizeof(int64_t) * np))) return rc;
    if ((rc = ensure(sv.owner, sizeof(int) * np))) return rc;
    if ((rc = ensure(sv.r2c, sizeof(int) * np))) return rc;
    if ((rc = ensure(sv.pred, sizeof(int) * np))) return rc;
    if ((rc = ensure(sv.list, sizeof(int) * np))) return rc;
    if ((rc = ensure(sv.bid, sizeof(unsigned long long) * np))) return rc;
    (void)c;
    return TD_OK;
}

template <typename CT>
int sv_compress_t(Solver &sv, bool *fits, bool speculate = false)
{
    Ctx &c = ctx();
    constexpr int E = Tr<CT>::E;
    const int n = sv.n, nrows = sv.nrows;
    const int nchunks = (n + E - 1) / E;
    int rc;
    if ((rc = ensure(sv.cc, std::max<size_t>((size_t)nrows * nchunks * 16, 256)))) return rc;
    int *ctl = (int *)sv.misc.p;
    TD_HIP(hipMemsetAsync(ctl, 0, CTL_ALL * sizeof(int), c.stream));  // flag, error, stats, range, shape
    const bool vec = (n % 4 == 0) && (((uintptr_t)sv.d_cost & 15) == 0);
    const int grid = std::max(1, std::min(nrows, c.n_cu * 8));
    sv.bid0_done = false;
    sv.state_ready = false;
    sv.zs_done = false;
    sv.zs_round0 = false;
    sv.began = false;
    sv.cc_partial = false;
    if (nrows > 0) {
        const int nq = n / 4;
        CT *cc = (CT *)sv.cc.p;
        int32_t *rm = (int32_t *)sv.rowmin.p;
        int *rcs = (int *)sv.rconst.p;
        // round 0's bids out of the compress pass: the state is initialised BEFORE it (bid keys zeroed, rows free; the
        // pass itself marks the deferred constant rows), td_assign then skips sv_begin_t and round 0's k_bid
        const bool bid0 = sv.want_bid0 && g_bid0 && sizeof(CT) == 1 && g_creg && vec && nq >= 3072 && nq <= 1024 * 16;   // n >= 12 288 (n = 9000: 0.52 -> 0.56 ms, the pass has too few rows per CU to hide the scan)
        int tickets = 0;
        // block-local start: whole diagonal blocks only, chunk-aligned column slices
        int zs_rpb = 0;
        if (bid0 && sv.zs_V > 0 && n % (sv.zs_V * E) == 0) {
            const int rpb = n / sv.zs_V;
            if (sv.row0 % rpb == 0 && nrows % rpb == 0 && nrows / rpb <= HOP_BMAX) zs_rpb = rpb;
        }
        // block-local start of rows that fit the 256-thread pass (n <= 16 384): round 0 of phase A as its own k_zs_bid launch
        // (it reads 1/V of every row: 32 MB at n = 16 384) behind the PLAIN pass, which is 27 us faster than the one that
        // scans the narrow words for the bids; wider rows keep the fused pass (their plain pass is no faster: spills)
        const bool zs_sep = zs_rpb > 0 && g_zs_sep && nq <= 256 * 16;
        if (zs_sep) {
            sv.zs_done = true;
            sv.zs_round0 = true;
        }
        const bool bid0_kernel = bid0 && !zs_sep;
        // the narrow cells outside the diagonal slices are only read if phase A leaves rows: do not write them until then
        // (7/8 of the 0.25 GiB copy at n = 16 384)
        const int diag = (sizeof(CT) == 1 && sv.lazy_cc && g_lazy_cc && bid0_kernel && zs_rpb > 0 && sv.d_cost && !sv.gen) ? 1 : 0;
        // (two restructurings of the diag_only pass were built and measured at n = 16 384, both SLOWER than this pass's
        // 215 us: one WAVE per row with only the slice waiting in registers — no LDS, no barrier — 284 - 390 us, slower the
        // more waves stream at once; the next row's loads issued before the barrier, so that every workgroup has a row
        // in flight all the time, 267 - 288 us.  More bytes in flight do not help this pass)
        static const int gbm = getenv("TD_BID0_GRID") ? atoi(getenv("TD_BID0_GRID")) : 0;
        const int gb = std::max(1, std::min(nrows, c.n_cu * (gbm > 0 ? gbm : std::max(1, g_cgrid / 2))));   // 512-thread workgroups of the BID0 pass
        if (bid0_kernel) {
            using PT = typename Tr<CT>::PT;
            const int npad = nchunks * E;
            const PT padkey = (PT)(Tr<CT>::BIG << 1) | (PT)1;
            const int gi = (std::max(npad, (int)CTL_WORDS) + 255) / 256;
            const int gc = nq <= 256 * 16 ? gb : std::max(1, std::min(nrows, c.n_cu * 2));
            tickets = sv.probe ? gi + gc : 0;
            k_init_state<PT><<<gi, 256, 0, c.stream>>>(n, npad, nrows, (PT *)sv.price.p, padkey, (int *)sv.owner.p, (int *)sv.r2c.p,
                                                       (unsigned long long *)sv.bid.p, ctl, nullptr, sv.probe, tickets);
            TD_HIP(hipMemsetAsync((char *)sv.misc.p + 1024, 0, 16, c.stream));
            sv.bid0_done = true;
        }
        ProfScope ps(TD_K_COMPRESS);
        unsigned long long *bidp = (unsigned long long *)sv.bid.p;
        int *defer_r2c = sv.defer_const ? (int *)sv.r2c.p : nullptr;
        if (g_creg && vec && nq <= 256 * 16) {
            const int g2 = std::max(1, std::min(nrows, c.n_cu * g_cgrid));
#define TD_CR(VPT) k_compress_reg<CT, VPT, 256><<<g2, 256, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip)
            if (nq <= 256) { TD_CR(1); }
            else if (nq <= 512) { TD_CR(2); }
            else if (nq <= 1024) { TD_CR(4); }
            else if (nq <= 2048) { TD_CR(8); }
            else if (bid0_kernel) {
                // 512 threads x 8 pieces: half the row registers per thread, so that the zero-byte scan fits without giving up waves
                if constexpr (sizeof(CT) == 1) {
                    static const int shape = getenv("TD_BID0_SHAPE") ? atoi(getenv("TD_BID0_SHAPE")) : 0;
                    if (shape == 1)
                        k_compress_reg<CT, 4, 1024, true><<<gb, 1024, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip, bidp,
                                                                                    sv.row0, defer_r2c, tickets, zs_rpb, diag);
                    else
                        k_compress_reg<CT, 8, 512, true><<<gb, 512, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip, bidp,
                                                                                   sv.row0, defer_r2c, tickets, zs_rpb, diag);
                    sv.zs_done = zs_rpb > 0;
                    sv.cc_partial = diag != 0;
                    sv.cc_rpb = zs_rpb;
                }
            } else {
                static const int cshape = getenv("TD_CREG_SHAPE") ? atoi(getenv("TD_CREG_SHAPE")) : 0;
                if (cshape == 1)
                    k_compress_reg<CT, 8, 512><<<gb, 512, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip);
                else if (cshape == 2)
                    k_compress_reg<CT, 4, 1024><<<gb, 1024, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip);
                else { TD_CR(16); }
            }
#undef TD_CR
        } else if (g_creg && vec && nq <= 1024 * 16) {
            const int g4 = std::max(1, std::min(nrows, c.n_cu * 2));
            static const int clds = getenv("TD_CLDS") ? atoi(getenv("TD_CLDS")) : 0;   // pieces per thread parked in LDS (0: all in registers).  Measured at n = 65 536: 0 / 8 / 4 -> 4.046 / 4.032 / 4.637 ms: the 52 bytes per lane the register shape spills are not what holds the pass at 5.3 TB/s of traffic
            if (bid0_kernel) {
                if constexpr (sizeof(CT) == 1)
                {
                    if (clds == 8) {
                        const int shm = 8 * 1024 * 16;
                        (void)hipFuncSetAttribute((const void *)k_compress_reg<CT, 16, 1024, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
                        k_compress_reg<CT, 16, 1024, true, 8><<<g4, 1024, shm, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip, bidp,
                                                                                          sv.row0, defer_r2c, tickets, zs_rpb, diag);
                    } else if (clds == 4) {
                        const int shm = 4 * 1024 * 16;
                        (void)hipFuncSetAttribute((const void *)k_compress_reg<CT, 16, 1024, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
                        k_compress_reg<CT, 16, 1024, true, 4><<<g4, 1024, shm, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip, bidp,
                                                                                          sv.row0, defer_r2c, tickets, zs_rpb, diag);
                    } else
                        k_compress_reg<CT, 16, 1024, true><<<g4, 1024, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip, bidp,
                                                                                      sv.row0, defer_r2c, tickets, zs_rpb, diag);
                    sv.zs_done = zs_rpb > 0;
                    sv.cc_partial = diag != 0;
                    sv.cc_rpb = zs_rpb;
                }
            } else
                k_compress_reg<CT, 16, 1024><<<g4, 1024, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip);
        } else if (vec)
            k_compress<CT, true><<<grid, 256, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip);
        else
            k_compress<CT, false><<<grid, 256, 0, c.stream>>>(n, nrows, nchunks, sv.d_cost, cc, rm, ctl, rcs, sv.skip);
    }
    TD_HIP(hipGetLastError());
    if (speculate) {
        // do not stall the stream on the "row range fits" flag: carry on as if it fits; the flag
        // comes back with the final read-back and a wrong guess is simply redone one width up
        *fits = true;
        sv.nconst = -1;   // not known on the host
    } else {
        TD_HIP(hipMemcpyAsync(c.pinned, ctl, CTL_ALL * sizeof(int), hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        *fits = (((int *)c.pinned)[CTL_FLAG] == 0);
        sv.nconst = ((int *)c.pinned)[CTL_NCONST];
        sv.range_seen = (int64_t)(((const unsigned long long *)((int *)c.pinned + CTL_RSEEN))[0]);   // exact: the largest row range of the pass
        if (!*fits) c.stats[6] = (int64_t)(((const unsigned long long *)((int *)c.pinned + CTL_RANGE))[0]);
    }
    if (*fits) {
        sv.bpc = IsNP<CT>::value ? 5 : (int)sizeof(CT);   // 5: 4-byte cells, 32-bit prices
        sv.nchunks = nchunks;
        sv.npad = nchunks * E;
    }
    return TD_OK;
}

int sv_compress(Solver &sv, int bpc, bool *fits, bool speculate = false)
{
    switch (bpc) {
        case 1: return sv_compress_t<uint8_t>(sv, fits, speculate);
        case 2: return sv_compress_t<uint16_t>(sv, fits, speculate);
        case 4: return sv_compress_t<uint32_t>(sv, fits, speculate);
        case 5: return sv_compress_t<u32n>(sv, fits, speculate);
    }
    return fail(TD_EINVAL, "bytes per cell must be 1, 2 or 4");
}

// padded model: the transposed problem's 4-byte cells straight from the caller's matrix (k_compress_tr)
int sv_compress_fused(Solver &sv, bool *fits, int64_t *range, bool cells8 = false, int esc_raw = 0, bool nosync = false,
                      int64_t assumed_range = -1 /* nosync: the range the caller goes on with (a padded matrix whose real cells exceed the pad value) */)
{
    Ctx &c = ctx();
    const int n = sv.n;
    const int nchunks = cells8 ? (n + 15) / 16 : (n + 3) / 4, npad = cells8 ? nchunks * 16 : nchunks * 4;
    int rc;
    if ((rc = ensure(sv.cc, std::max<size_t>((size_t)n * nchunks * 16, 256)))) return rc;
    int *ctl = (int *)sv.misc.p;
    TD_HIP(hipMemsetAsync(ctl, 0, CTL_ALL * sizeof(int), c.stream));
    int *colmin = (int *)sv.pred.p, *colmax = (int *)sv.list.p;   // free until the finisher
    {
        ProfScope ps(TD_K_COMPRESS);
        k_fill_i32<<<(n + 255) / 256, 256, 0, c.stream>>>(colmin, n, INT_MAX);
        k_fill_i32<<<(n + 255) / 256, 256, 0, c.stream>>>(colmax, n, INT_MIN);
        if (cells8) {
            static const int tr8_rows = getenv("TD_TR8_ROWS") ? atoi(getenv("TD_TR8_ROWS")) : 512;   // rows per tile: 512 = 33 KB of LDS, 4 workgroups per CU (g3 N = 16 384: compress 0.33 ms; 1024 rows, 2 per CU: 0.35 - 0.38; 256: 0.36)
#define TD_TR8(R)                                                                                                                          \
    do {                                                                                                                                   \
        const size_t shm = (size_t)64 * (R + 4);                                                                                           \
        (void)hipFuncSetAttribute((const void *)k_compress_tr8<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                  \
        k_compress_tr8<R><<<dim3((n + 63) / 64, (n + R - 1) / R), 256, shm, c.stream>>>(n, npad, sv.d_cost, (uint8_t *)sv.cc.p, 0, esc_raw, \
                                                                                       colmin, colmax, ctl);                              \
    } while (0)
            static const int tr8_wide = getenv("TD_TR8_WIDE") ? atoi(getenv("TD_TR8_WIDE")) : 128;   // > 0: 256-column tiles of this many rows (128 / 256; g3 N = 16 384 step: 64-column tiles 0.838, 256 x 128 0.821, 256 x 256 0.841 ms — the tile shape is not what holds this pass at 3.8 TB/s of traffic)
#define TD_TR8W(R)                                                                                                                           \
    do {                                                                                                                                     \
        const size_t shm = (size_t)256 * (R + 4);                                                                                            \
        (void)hipFuncSetAttribute((const void *)k_compress_tr8w<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                   \
        k_compress_tr8w<R><<<dim3((n + 255) / 256, (n + R - 1) / R), 256, shm, c.stream>>>(n, npad, sv.d_cost, (uint8_t *)sv.cc.p, 0, esc_raw, \
                                                                                          colmin, colmax, ctl);                             \
    } while (0)
            if (sv.gen) {   // cells made from the position arrays (td_build_assign)
                const size_t shm = (size_t)256 * (128 + 4);
                (void)hipFuncSetAttribute((const void *)k_compress_tr8w<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                k_compress_tr8w<128, true><<<dim3((n + 255) / 256, (n + 127) / 128), 256, shm, c.stream>>>(n, npad, nullptr, (uint8_t *)sv.cc.p, 0, esc_raw,
                                                                                                          colmin, colmax, ctl, sv.gsrc);
            } else if (tr8_wide == 128) TD_TR8W(128);
            else if (tr8_wide == 256) TD_TR8W(256);
            else if (tr8_rows == 256) TD_TR8(256);
            else if (tr8_rows == 512) TD_TR8(512);
            else TD_TR8(1024);
#undef TD_TR8W
#undef TD_TR8
        } else if (sv.gen && n > 4096)
            k_compress_tr<8, true><<<dim3((n + 63) / 64, (n + 511) / 512), 256, 0, c.stream>>>(n, npad, nullptr, (uint32_t *)sv.cc.p, 0, colmin, colmax, ctl, sv.gsrc);
        else if (sv.gen)
            k_compress_tr<1, true><<<dim3((n + 63) / 64, (n + 63) / 64), 256, 0, c.stream>>>(n, npad, nullptr, (uint32_t *)sv.cc.p, 0, colmin, colmax, ctl, sv.gsrc);
        else if (n > 4096)   // tall workgroups: fewer atomics per column; small models need the workgroups instead
            k_compress_tr<8><<<dim3((n + 63) / 64, (n + 511) / 512), 256, 0, c.stream>>>(n, npad, sv.d_cost, (uint32_t *)sv.cc.p, 0, colmin, colmax, ctl);
        else
            k_compress_tr<1><<<dim3((n + 63) / 64, (n + 63) / 64), 256, 0, c.stream>>>(n, npad, sv.d_cost, (uint32_t *)sv.cc.p, 0, colmin, colmax, ctl);
        k_tr_finish<<<(n + 255) / 256, 256, 0, c.stream>>>(n, 0, colmin, colmax, (int32_t *)sv.rowmin.p, (int *)sv.rconst.p, ctl, esc_raw,
                                                           cells8 ? 1 : 0, nosync ? (long long)assumed_range : -1ll);
    }
    TD_HIP(hipGetLastError());
    if (nosync) {   // speculative like the 1-byte attempt of a square model: the flags come home with the final read-back
        *fits = true;
        sv.nconst = -1;
        sv.nchunks = nchunks;
        sv.npad = npad;
        return TD_OK;
    }
    TD_HIP(hipMemcpyAsync(c.pinned, ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    const int *h = (const int *)c.pinned;
    *range = (int64_t)(((const unsigned long long *)(h + CTL_RANGE))[0]);
    *fits = h[CTL_FLAG] == 0 && *range <= (int64_t)Tr<uint32_t>::LIMIT && *range >= 0;
    sv.nconst = h[CTL_NCONST];
    if (*fits) {
        sv.nchunks = nchunks;
        sv.npad = npad;
    }
    return TD_OK;
}

// One two-hop pass (td_blocks.h) over `nb` blocks of rpb rows x ncols_blk columns starting at column col_lo.
// The rest of a narrow copy whose compress pass stored the diagonal slices only (k_compress_reg diag_only).
int sv_complete_cc(Solver &sv)
{
    if (!sv.cc_partial) return TD_OK;
    Ctx &c = ctx();
    ProfScope ps(TD_K_COMPRESS);
    k_compress_rest<<<std::max(1, std::min(sv.nrows, c.n_cu * 8)), 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, sv.nchunks, sv.cc_rpb, sv.d_cost,
                                                                                      (uint8_t *)sv.cc.p, (const int32_t *)sv.rowmin.p,
                                                                                      (const int *)sv.misc.p);
    TD_HIP(hipGetLastError());
    sv.cc_partial = false;
    return TD_OK;
}

// raw: the cells come from the caller's int32 matrix and the row minima (a solve whose narrow copy holds the diagonal
// slices only, sv.cc_partial)
template <typename CT>
int sv_hop_t(Solver &sv, int rpb, int ncols_blk, int col_lo, int nb, bool window_zero, uint8_t *ob, bool raw = false)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    int rc;
    if ((rc = ensure(sv.esc, sizeof(unsigned long long) * 2 * (size_t)std::max(sv.nrows, 1)))) return rc;
    if ((rc = ensure(sv.hop, sizeof(HopCtl)))) return rc;
    if ((rc = ensure(sv.hoptab, sizeof(int) * (size_t)nb * HOP_FMAX * HOP_FMAX))) return rc;
    HopCtl *hc = (HopCtl *)sv.hop.p;
    int *frl = (int *)sv.list.p, *fcl = (int *)sv.pred.p;   // free until the finisher
    const int *ctl = (const int *)sv.misc.p;
    k_hop_lists<<<nb, 1024, 0, c.stream>>>(rpb, ncols_blk, col_lo, (const int *)sv.r2c.p, (const int *)sv.owner.p, frl, fcl, hc, ctl);
    if (raw) {
        if (!sv.d_cost || sv.n % 4) return fail(TD_EINVAL, "two-hop pass on the int32 matrix: no matrix, or n %% 4 != 0");
        k_hop_esc<CT, true><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(sv.nrows, sv.n / 4, rpb, ncols_blk, col_lo, g_hop_max_rows,
                                                                      (const CT *)sv.d_cost, (const PT *)sv.price.p, (const int *)sv.r2c.p, fcl,
                                                                      hc, (unsigned long long *)sv.esc.p, ctl, (const int32_t *)sv.rowmin.p);
        k_hop_table<CT, true><<<nb * HOP_FMAX, 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, sv.n / 4, rpb, ncols_blk, col_lo, g_hop_max_rows,
                                                                   window_zero ? 1 : 0, (const CT *)sv.d_cost, (const PT *)sv.price.p,
                                                                   (const int *)sv.owner.p, frl, hc, (const unsigned long long *)sv.esc.p,
                                                                   (int *)sv.hoptab.p, ctl, (const int32_t *)sv.rowmin.p);
    } else {
    k_hop_esc<CT><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(sv.nrows, sv.nchunks, rpb, ncols_blk, col_lo, g_hop_max_rows, (const CT *)sv.cc.p,
                                                            (const PT *)sv.price.p, (const int *)sv.r2c.p, fcl, hc,
                                                            (unsigned long long *)sv.esc.p, ctl);
    k_hop_table<CT><<<nb * HOP_FMAX, 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, sv.nchunks, rpb, ncols_blk, col_lo, g_hop_max_rows,
                                                         window_zero ? 1 : 0, (const CT *)sv.cc.p, (const PT *)sv.price.p,
                                                         (const int *)sv.owner.p, frl, hc, (const unsigned long long *)sv.esc.p,
                                                         (int *)sv.hoptab.p, ctl);
    }
    k_hop_match<PT><<<nb, HOP_FMAX, sizeof(uint32_t) * (size_t)((rpb + 31) / 32), c.stream>>>(
        sv.nrows, sv.row0, rpb, ncols_blk, col_lo, g_hop_max_rows, (PT *)sv.price.p, (int *)sv.owner.p, (int *)sv.r2c.p, ob, frl, fcl, hc,
        (const int *)sv.hoptab.p, (int *)sv.misc.p);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

// Phase A of a 1-byte attempt whose compress pass wrote the zero-slice bids (sv.zs_done): everything local to this
// shard's diagonal blocks, no price moves.  Leaves the ordinary state (owner / r2c / packed prices with the owned bit)
// for the global rounds; the free rows it left are counted in HopCtl::left.
template <typename CT>
int sv_begin_t(Solver &sv);

int sv_phase_a(Solver &sv, int hop_passes)
{
    Ctx &c = ctx();
    const int n = sv.n, nrows = sv.nrows;
    const int rpb = n / sv.zs_V, nb = nrows / rpb;
    const int col_lo = (sv.row0 / rpb) * rpb, col_hi = col_lo + nb * rpb;
    int rc;
    if ((rc = ensure(sv.ob, (size_t)sv.npad + 64))) return rc;
    TD_HIP(hipMemsetAsync(sv.ob.p, 0, (size_t)sv.npad, c.stream));
    ProfScope ps(TD_K_BID);
    int *ctl = (int *)sv.misc.p;
    const int ga = (col_hi - col_lo + 255) / 256;
    unsigned long long *bid = (unsigned long long *)sv.bid.p;
    auto assign = [&]() {
        k_zs_assign<<<ga, 256, 0, c.stream>>>(col_lo, col_hi, nrows, sv.row0, bid, (int32_t *)sv.price.p, (int *)sv.owner.p, (int *)sv.r2c.p,
                                              (uint8_t *)sv.ob.p, ctl);
    };
    if (sv.zs_round0) {   // round 0 as its own launch behind the plain compress pass (the same bids as the fused pass writes, bit for bit)
        if (!sv.began && (rc = sv_begin_t<uint8_t>(sv))) return rc;   // (a shard: td_shard_begin comes after the exchange)
        k_zs_bid<<<std::min(nrows, c.n_cu * 16), 256, 0, c.stream>>>(nrows, sv.row0, sv.nchunks, rpb, (const uint8_t *)sv.cc.p,
                                                                     (const uint8_t *)sv.ob.p, (const int *)sv.r2c.p, bid, ctl, 0, 0);
        sv.zs_round0 = false;
    }
    assign();   // round 0
    for (int r = 1; r <= g_zs_rounds; r++) {
        k_zs_bid<<<std::min(nrows, c.n_cu * 16), 256, 0, c.stream>>>(nrows, sv.row0, sv.nchunks, rpb, (const uint8_t *)sv.cc.p,
                                                                     (const uint8_t *)sv.ob.p, (const int *)sv.r2c.p, bid, ctl, r, g_tie_evict);
        assign();
    }
    TD_HIP(hipGetLastError());
    for (int p = 0; p < hop_passes; p++)
        if ((rc = sv_hop_t<uint8_t>(sv, rpb, rpb, col_lo, nb, true, (uint8_t *)sv.ob.p))) return rc;
    if (getenv("TD_DEBUG") && hop_passes > 0) {
        HopCtl h;
        TD_HIP(hipMemcpyAsync(&h, sv.hop.p, sizeof(h), hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        fprintf(stderr, "[td] phase A: %d blocks of %d rows, %d local rounds; free rows / columns per block before the last two-hop pass:", nb, rpb, g_zs_rounds);
        for (int b = 0; b < nb; b++) fprintf(stderr, " %d/%d", h.nfr[b], h.nfc[b]);
        fprintf(stderr, " | matched %d, left %d\n", h.matched, h.left);
    }
    sv.zs_done = false;
    sv.bid0_done = false;   // the global rounds start with a round 0 of their own
    sv.state_ready = true;
    return TD_OK;
}

template <typename CT>
int sv_begin_t(Solver &sv)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    const PT padkey = (PT)(Tr<CT>::BIG << 1) | (PT)1;
    k_init_state<PT><<<(std::max(sv.npad, (int)CTL_WORDS) + 255) / 256, 256, 0, c.stream>>>(
        sv.n, sv.npad, sv.nrows, (PT *)sv.price.p, padkey, (int *)sv.owner.p, (int *)sv.r2c.p,
        (unsigned long long *)sv.bid.p, (int *)sv.misc.p, sv.defer_const ? (const int *)sv.rconst.p : nullptr, sv.probe);
    TD_HIP(hipMemsetAsync((char *)sv.misc.p + 1024, 0, 16, c.stream));
    TD_HIP(hipGetLastError());
    sv.began = true;
    return TD_OK;
}

template <typename CT>
int sv_bid_t(Solver &sv, int r, unsigned long long *keys)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    const int n = sv.n, nrows = sv.nrows;
    if (nrows == 0) return TD_OK;
    const size_t lds_prices = (size_t)sv.npad * sizeof(PT);
    const bool can_lds = lds_prices <= 128 * 1024 && n >= 2048 && nrows >= 1024;
    const int tie_evict = (r >= 1) ? g_tie_evict : 0;
    ProfScope ps(TD_K_BID);
    int *tied = (r == 0) ? (int *)sv.misc.p + CTL_TIED : nullptr;
    const bool sparse = sv.state_ready;   // after the block-local start few rows are free: every round is a handful of row workgroups
    if (can_lds && r < g_lds_rounds && !sparse) {
        if (lds_prices > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_bid<CT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prices);
        const int grid = std::min((nrows + 15) / 16, c.n_cu * g_lds_grid);
        k_bid<CT, true><<<grid, 1024, lds_prices, c.stream>>>(n, nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p, (const PT *)sv.price.p,
                                                              (const int *)sv.r2c.p, keys, (const int *)sv.misc.p, r, tie_evict, 1, 0, tied);
    } else if (r >= g_row_rounds || sparse) {
        // one workgroup per row while a round still has hundreds of bidders (perf.jl: 9.5 us against 10.4 with the loop),
        // a grid-stride loop over the rows from round TD_ROW_LOOP on, where a handful is left and spawning 16 384
        // workgroups to see that their rows are assigned costs more than the round's work
        k_bid_row<CT><<<(r < g_row_loop && !sparse) ? nrows : std::min(nrows, c.n_cu * 8), 256, 0, c.stream>>>(n, nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p, (const PT *)sv.price.p,
                                                   (const int *)sv.r2c.p, keys, (const int *)sv.misc.p, r, tie_evict);
    } else {
        k_bid<CT, false><<<(nrows + 3) / 4, 256, 0, c.stream>>>(n, nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p, (const PT *)sv.price.p,
                                                                (const int *)sv.r2c.p, keys, (const int *)sv.misc.p, r, tie_evict, 1, 0, tied);
    }
    TD_HIP(hipGetLastError());
    return TD_OK;
}

template <typename CT>
int sv_apply_t(Solver &sv, int r, unsigned long long *keys)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    ProfScope ps(TD_K_ASSIGN);
    k_assign<PT><<<(sv.n + 255) / 256, 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, keys, (PT *)sv.price.p, (int *)sv.owner.p,
                                                           (int *)sv.r2c.p, (int *)sv.misc.p, r, IsNP<CT>::value ? g_np_plimit : 0ll);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

template <typename CT, int CH, bool LDSST, int TB = 1024>
void launch_sap(Solver &sv, const ShardTab &tab, int *r2c_full, int T, size_t shm)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    if (shm > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)k_sap<CT, CH, LDSST, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    k_sap<CT, CH, LDSST, TB><<<1, T, shm, c.stream>>>(sv.n, sv.nchunks, tab, (PT *)sv.price.p, (int *)sv.owner.p, r2c_full,
                                                      (int *)sv.pred.p, (int *)sv.list.p, (int *)sv.misc.p);
}

// finisher; r2c_full is indexed by GLOBAL row (== sv.r2c for an unsharded solve)
template <typename CT>
int sv_finish_t(Solver &sv, const ShardTab &tab, int *r2c_full)
{
    constexpr int E = Tr<CT>::E;
    const int n = sv.n, nchunks = sv.nchunks, npad = sv.npad;
    int CH = 1;
    while (CH * 1024 < nchunks) CH *= 2;
    int T = (nchunks + CH - 1) / CH;
    T = std::min(1024, std::max(64, ((T + 63) / 64) * 64));
    const size_t st = (size_t)npad * 2 * sizeof(int);  // owner[] + pred[]
    const bool lds = st <= 150 * 1024;
    const size_t shm = lds ? st : 0;
    if (CH * E > 64) return fail(TD_ERANGE, "n=%d too large for the single-workgroup finisher", n);
    ProfScope ps(TD_K_SAP);
    k_freelist<<<1, 1024, 0, ctx().stream>>>(n, r2c_full, (int *)sv.list.p, (int *)sv.misc.p);
    // ---- 4-byte rows of n >= TD_FOREST_MIN_N: the incremental shortest-path forest over all CUs (td_forest.h)
    if constexpr (sizeof(CT) == 4) {
        if (g_forest && n >= g_forest_min_n && n <= 32768 && tab.count == 1) {
            Ctx &c = ctx();
            using PT = typename Tr<CT>::PT;
            {   // nothing left for a finisher (thresholded models, easy instances): no cooperative launch for it
                TD_HIP(hipMemcpyAsync(c.pinned, (int *)sv.misc.p + CTL_NFREE, sizeof(int), hipMemcpyDeviceToHost, c.stream));
                TD_HIP(hipStreamSynchronize(c.stream));
                if (((int *)c.pinned)[0] == 0) return TD_OK;
            }
            typedef typename std::conditional<sizeof(PT) == 4, int, long long>::type LT;
            const size_t off_base = (sizeof(FoShared) + 255) / 256 * 256;
            const size_t n4 = (size_t)(n + 3) / 4 * 4;   // the int arrays start 16-byte aligned and are padded (16-byte loads)
            const size_t off_root = off_base + sizeof(long long) * n4, off_col = off_root + sizeof(int) * n4;
            const size_t off_pc = off_col + sizeof(int) * n4;
            int rc = ensure(sv.fbuf, off_pc + sizeof(int) * n4);
            if (rc) return rc;
            FoShared *fs = (FoShared *)sv.fbuf.p;
            long long *g_base = (long long *)((char *)sv.fbuf.p + off_base);
            int *g_root = (int *)((char *)sv.fbuf.p + off_root), *g_col = (int *)((char *)sv.fbuf.p + off_col);
            int *g_pcp = (int *)((char *)sv.fbuf.p + off_pc);
            TD_HIP(hipMemsetAsync(fs, 0, offsetof(FoShared, board), c.stream));
            k_forest_fill<<<(n + 255) / 256, 256, 0, c.stream>>>(n, g_base, g_root, g_col, g_pcp, (long long)FoLim<LT>::INF);
            k_forest_init<CT><<<(n + 3) / 4, 256, 0, c.stream>>>(n, nchunks, tab, (const PT *)sv.price.p, (const int *)sv.list.p,
                                                                (const int *)sv.misc.p, g_base, g_root, g_col);
            int a_n = n, a_nch = nchunks, a_pcl = (n <= 16384) ? 1 : 0;
            ShardTab a_tab = tab;
            PT *a_pk = (PT *)sv.price.p;
            int *a_owner = (int *)sv.owner.p, *a_r2c = r2c_full, *a_ctl = (int *)sv.misc.p;
            long long a_w0 = g_forest_w0, a_wx = g_forest_wx;
            void *kargs[] = {&a_n, &a_nch, &a_tab, &a_pk, &a_owner, &a_r2c, &g_pcp, &g_base, &g_root, &g_col, &a_ctl, &fs, &a_w0, &a_wx, &a_pcl};
            const size_t dyn = std::max<size_t>(2 * n4, a_pcl ? 6 * n4 : 0);   // forest row list (u16) / predecessor columns + owners (u16) of workgroup 0 at an END
            hipError_t le;
            static const int force_cw = getenv("TD_FOREST_CW") ? atoi(getenv("TD_FOREST_CW")) : 0;
            static const int force_tb = getenv("TD_FOREST_TB") ? atoi(getenv("TD_FOREST_TB")) : 1024;
#define TD_FO_LAUNCH(CWV, TBV)                                                                                                        \
    do {                                                                                                                              \
        (void)hipFuncSetAttribute((const void *)k_forest<CT, CWV, TBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);        \
        le = hipLaunchCooperativeKernel((const void *)k_forest<CT, CWV, TBV>, dim3((n + CWV - 1) / CWV), dim3(TBV), kargs, dyn, c.stream); \
    } while (0)
            if (n <= 16384 && force_cw != 128) {
                if (force_tb >= 1024) TD_FO_LAUNCH(64, 1024);
                else if (force_tb >= 512) TD_FO_LAUNCH(64, 512);
                else TD_FO_LAUNCH(64, 256);
            } else {
                if (force_tb >= 512) TD_FO_LAUNCH(128, 512);
                else TD_FO_LAUNCH(128, 256);
            }
#undef TD_FO_LAUNCH
            if (le == hipSuccess) {
                TD_HIP(hipGetLastError());
                if (getenv("TD_DEBUG")) {
                    long long st[16];
                    TD_HIP(hipMemcpyAsync(st, (char *)fs + offsetof(FoShared, stat), sizeof(st), hipMemcpyDeviceToHost, c.stream));
                    TD_HIP(hipStreamSynchronize(c.stream));
                    fprintf(stderr, "[td] k_forest n=%d: levels %lld entries %lld ENDs %lld empty %lld repair rows %lld (need cols wg0 %lld) trees %lld | Mcycles wg0: select %lld barrier %lld board %lld relax %lld (main %lld comb %lld) END %lld (lpc %lld) repair %lld\n",
                            n, st[0], st[1], st[2], st[3], st[4], st[15], st[5], st[8] >> 20, st[9] >> 20, st[10] >> 20, st[11] >> 20, st[6] >> 20, st[7] >> 20, st[12] >> 20, st[14] >> 20, st[13] >> 20);
                }
                return TD_OK;
            }
            (void)hipGetLastError();   // refused: the finishers below do the job
        }
    }
    // ---- speculative parallel searches first (a few batches), the serial workgroup mops up
    // (u8 instances go straight to the lean tie-batching serial workgroup, which is faster there)
    int nfree_left = -1;   // free rows the speculative batches left (-1: not read back)
    if (g_psap_batches > 0 && CH <= 4 && CH * E <= 16 && lds && n >= 64 && !IsLean8<CT>::value && !sv.tick_sized) {
        Ctx &c = ctx();
        using PT = typename Tr<CT>::PT;
        int rc = ensure(sv.psrec, sizeof(PsRec<PT>) * (size_t)PS_G);
        if (rc) return rc;
        PsRec<PT> *recs = (PsRec<PT> *)sv.psrec.p;
        const size_t cshm = 2 * sizeof(int) * (size_t)n;
        if (cshm > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_pcommit<PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cshm);
        unsigned long long *raise = (unsigned long long *)sv.bid.p;  // all zero after the bidding rounds
        // the speculative search is the finisher kernel itself in SPEC mode (same batched steps)
        const bool tb512 = g_sap512 && T <= 512;
        auto search = [&]() {
#define TD_PS(CHV, TBV)                                                                                                    \
    if constexpr (CHV * E <= 16) {                                                                                         \
        if (shm > 48 * 1024)                                                                                               \
            (void)hipFuncSetAttribute((const void *)k_sap<CT, CHV, true, TBV, true>,                                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                               \
        k_sap<CT, CHV, true, TBV, true><<<PS_G, T, shm, c.stream>>>(n, nchunks, tab, (PT *)sv.price.p, (int *)sv.owner.p,  \
                                                                   r2c_full, (int *)sv.pred.p, (int *)sv.list.p,           \
                                                                   (int *)sv.misc.p, recs);                                \
    }
            if (CH == 1) {
                if (tb512) { TD_PS(1, 512) } else { TD_PS(1, 1024) }
            } else if (CH == 2) {
                if (tb512) { TD_PS(2, 512) } else { TD_PS(2, 1024) }
            } else {
                if (tb512) { TD_PS(4, 512) } else { TD_PS(4, 1024) }
            }
#undef TD_PS
        };
        // Batches go out in groups; one small read-back of the free-row count per group decides
        // whether another group is worth it. With a handful of free rows the serial workgroup
        // is faster than any batch, and a group that commits fewer than two rows per batch means
        // the searches have started to collide (or to run over the record length).
        auto read_nfree = [&](int *out) -> int {
            TD_HIP(hipMemcpyAsync(c.pinned, (int *)sv.misc.p + CTL_NFREE, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            TD_HIP(hipStreamSynchronize(c.stream));
            *out = ((int *)c.pinned)[0];
            return TD_OK;
        };
        int nfree = 0, launched = 0;
        if ((rc = read_nfree(&nfree))) return rc;
        // a tick-sized model with a dozen free rows: the serial workgroup is through before a batch + its commit are (tick: 0.346 -> 0.332 ms)
        const int psap_min = n < 2048 ? std::max(g_psap_min, 32) : g_psap_min;
        while (nfree >= psap_min && launched < g_psap_cap) {
            const int nb = std::max(1, std::min(g_psap_batches, (nfree + 31) / 32));
            for (int b = 0; b < nb; b++) {
                search();
                k_pcommit<PT><<<1, 1024, cshm, c.stream>>>(n, (PT *)sv.price.p, (int *)sv.owner.p, r2c_full, (int *)sv.list.p,
                                                           (int *)sv.misc.p, raise, recs, 0, PS_G, IsNP<CT>::value ? g_np_plimit : 0ll);
            }
            launched += nb;
            TD_HIP(hipGetLastError());
            int left = 0;
            if ((rc = read_nfree(&left))) return rc;
            const bool worth = (nfree - left) >= g_psap_worth * nb;
            nfree = left;
            if (!worth) break;
        }
        nfree_left = nfree;
    }
    {   // cooperative finisher: the columns of one search split over several CUs, 256-thread
        // workgroups.  Launched with hipLaunchCooperativeKernel: its grid barrier needs every
        // workgroup resident at the same time, which an ordinary launch does not promise when other
        // streams or processes share the GPU.  If the cooperative launch is refused the
        // single-workgroup finisher below does the job.
#ifndef TD_SX_TX
#define TD_SX_TX 128
#endif
        constexpr int TXsel = (sizeof(CT) == 4) ? TD_SX_TX : 256;
        // narrow cells of a very wide matrix (1-byte rows at n = 65 536: 4096 chunks): 64-thread column slices, so that 64
        // CUs instead of 16 pull the published rows (a step of ~70 rows x 64 KiB took 27 us through 16 CUs)
        const bool slim = sizeof(CT) < 4 && nchunks >= g_sapx_slim_chunks;
        const int KX = slim ? (nchunks + 63) / 64 : (nchunks + TXsel - 1) / TXsel;
        const bool lean8 = IsLean8<CT>::value && CH == 1 && g_sap8;
        bool launched_x = false;
        // from 8 x 256 chunks on always; from 4 x 256 on when many rows are left (|a-b| n = 6000: 190 -> 109 ms with 62 rows;
        // uniform 0..10^6 n = 4096 with 10 rows: 9.8 -> 12.1 ms, so not for a handful)
        const bool big = slim || KX * TXsel >= g_sapx_min * 256 || (KX * TXsel >= g_sapx_min * 128 && nfree_left >= g_sapx_rows);
        if (g_sapx && !lean8 && big && KX <= SX_KMAX) {
            Ctx &c = ctx();
            using PT = typename Tr<CT>::PT;
            int rc = ensure(sv.xbuf, sizeof(SxShared));
            if (rc) return rc;
            TD_HIP(hipMemsetAsync(sv.xbuf.p, 0, 64, c.stream));   // barrier counter, abort flag, debug counters
            int a_n = n, a_nch = nchunks;
            ShardTab a_tab = tab;
            PT *a_pk = (PT *)sv.price.p;
            int *a_owner = (int *)sv.owner.p, *a_r2c = r2c_full, *a_pred = (int *)sv.pred.p, *a_ctl = (int *)sv.misc.p;
            const int *a_list = (const int *)sv.list.p;
            SxShared *a_sh = (SxShared *)sv.xbuf.p;
            void *kargs[] = {&a_n, &a_nch, &a_tab, &a_pk, &a_owner, &a_r2c, &a_pred, &a_list, &a_ctl, &a_sh};
            // 4-byte cells: 4 row groups per workgroup (1024 threads, 4 waves per SIMD in the relax phase)
            constexpr int NGsel = (sizeof(CT) == 4) ? TD_SX_NG : 1;
            hipError_t le;
            if constexpr (sizeof(CT) < 4) {
                if (slim)
                    le = hipLaunchCooperativeKernel((const void *)k_sapx<CT, 64, 1>, dim3(KX), dim3(64), kargs, 0, c.stream);
                else
                    le = hipLaunchCooperativeKernel((const void *)k_sapx<CT, TXsel, NGsel>, dim3(KX), dim3(TXsel * NGsel), kargs, 0, c.stream);
            } else
                le = hipLaunchCooperativeKernel((const void *)k_sapx<CT, TXsel, NGsel>, dim3(KX), dim3(TXsel * NGsel), kargs, 0, c.stream);
            if (le == hipSuccess) launched_x = true;
            else (void)hipGetLastError();
        }
        if (launched_x) {
            Ctx &c = ctx();
            TD_HIP(hipGetLastError());
            if (getenv("TD_DEBUG")) {
                long long dbg[8];
                TD_HIP(hipMemcpyAsync(dbg, sv.xbuf.p, sizeof(dbg), hipMemcpyDeviceToHost, c.stream));
                TD_HIP(hipStreamSynchronize(c.stream));
                fprintf(stderr, "[td] k_sapx K=%d cycles: barrier %lld select %lld read %lld relax %lld | rows published %lld in %lld steps\n",
                        KX, dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]);
            }
            return TD_OK;
        }
    }
    if constexpr (IsLean8<CT>::value) {
        if (CH == 1 && g_sap8) {
            Ctx &c = ctx();
            if (lds && g_psap8_batches > 0 && n >= 64) {
                // u8: the free rows (a handful on perf.jl instances) are searched concurrently by
                // the lean tie-batching search, one workgroup each; no host read-back — workgroups
                // beyond the number of free rows exit at once
                int rc = ensure(sv.psrec, sizeof(PsRec<int32_t>) * (size_t)PS_G);
                if (rc) return rc;
                PsRec<int32_t> *recs = (PsRec<int32_t> *)sv.psrec.p;
                const size_t cshm = 2 * sizeof(int) * (size_t)n;
                if (shm > 48 * 1024)
                    (void)hipFuncSetAttribute((const void *)k_sap8<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                if (cshm > 48 * 1024)
                    (void)hipFuncSetAttribute((const void *)k_pcommit<int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cshm);
                unsigned long long *raise = (unsigned long long *)sv.bid.p;
                for (int b = 0; b < g_psap8_batches; b++) {
                    k_sap8<true, true><<<g_psap8_grid, T, shm, c.stream>>>(n, nchunks, tab, (int32_t *)sv.price.p, (int *)sv.owner.p,
                                                                          r2c_full, (int *)sv.pred.p, (int *)sv.list.p,
                                                                          (int *)sv.misc.p, recs);
                    k_pcommit<int32_t><<<1, 1024, cshm, c.stream>>>(n, (int32_t *)sv.price.p, (int *)sv.owner.p, r2c_full,
                                                                    (int *)sv.list.p, (int *)sv.misc.p, raise, recs, 0, g_psap8_grid);
                }
            }
            if (lds) {
                if (shm > 48 * 1024)
                    (void)hipFuncSetAttribute((const void *)k_sap8<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                k_sap8<true, false><<<1, T, shm, c.stream>>>(sv.n, sv.nchunks, tab, (int32_t *)sv.price.p, (int *)sv.owner.p, r2c_full,
                                                             (int *)sv.pred.p, (int *)sv.list.p, (int *)sv.misc.p, nullptr,
                                                             sv.defer_const ? 1 : 0);
            } else {
                k_sap8<false, false><<<1, T, 0, c.stream>>>(sv.n, sv.nchunks, tab, (int32_t *)sv.price.p, (int *)sv.owner.p, r2c_full,
                                                            (int *)sv.pred.p, (int *)sv.list.p, (int *)sv.misc.p, nullptr,
                                                            sv.defer_const ? 1 : 0);
            }
            sv.placed = sv.defer_const;   // the constant rows were placed by the finisher itself
            TD_HIP(hipGetLastError());
            return TD_OK;
        }
    }
    {   // 512-thread instance: same code compiled for a 256-VGPR budget (more rows in flight per step)
        int ch5 = 1;
        while (ch5 * 512 < nchunks) ch5 *= 2;
        // (measured: only a win when the workgroup has <= 512 threads anyway, i.e. no wave is given up)
        if (g_sap512 && ch5 == CH && ch5 * E <= 16) {
            int T5 = (nchunks + ch5 - 1) / ch5;
            T5 = std::min(512, std::max(64, ((T5 + 63) / 64) * 64));
#define TD_SAP5(CHV)                                                               \
    if constexpr (CHV * E <= 16) {                                                 \
        if (lds) launch_sap<CT, CHV, true, 512>(sv, tab, r2c_full, T5, shm);       \
        else launch_sap<CT, CHV, false, 512>(sv, tab, r2c_full, T5, shm);          \
    }
            if (ch5 == 1) { TD_SAP5(1) }
            else if (ch5 == 2) { TD_SAP5(2) }
            else { TD_SAP5(4) }
#undef TD_SAP5
            TD_HIP(hipGetLastError());
            return TD_OK;
        }
    }
#define TD_SAP(CHV)                                                    \
    if (lds) launch_sap<CT, CHV, true>(sv, tab, r2c_full, T, shm);     \
    else launch_sap<CT, CHV, false>(sv, tab, r2c_full, T, shm)
    switch (CH) {
        case 1: TD_SAP(1); break;
        case 2: TD_SAP(2); break;
        case 4: TD_SAP(4); break;
        case 8:
            if constexpr (E <= 8) { TD_SAP(8); }
            break;
        case 16:
            if constexpr (E <= 4) { TD_SAP(16); }
            break;
        default: return fail(TD_ERANGE, "n=%d too large for the single-workgroup finisher", n);
    }
#undef TD_SAP
    TD_HIP(hipGetLastError());
    return TD_OK;
}

template <typename CT>
int sv_totals_t(Solver &sv, bool want_dual)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    unsigned long long *out = (unsigned long long *)((char *)sv.misc.p + 1024);
    ProfScope ps(TD_K_FINAL);
    if (sv.nrows > 0 && sv.gen)
        k_final<true><<<std::min((sv.nrows + 255) / 256, 256), 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, nullptr, (const int *)sv.r2c.p,
                                                                                  (const int *)sv.owner.p, out, (int *)sv.misc.p, sv.fused_t ? 1 : 0, sv.gsrc);
    else if (sv.nrows > 0)
        k_final<false><<<std::min((sv.nrows + 255) / 256, 256), 256, 0, c.stream>>>(sv.n, sv.nrows, sv.row0, sv.d_cost, (const int *)sv.r2c.p,
                                                                                   (const int *)sv.owner.p, out, (int *)sv.misc.p, sv.fused_t ? 1 : 0);
    if (want_dual && (sv.nrows > 0 || sv.row0 == 0))
        k_dual<CT><<<std::max(1, std::min((sv.nrows + 3) / 4, c.n_cu * 8)), 256, 0, c.stream>>>(
            sv.n, sv.nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p, (const PT *)sv.price.p, (const int32_t *)sv.rowmin.p, out);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

// The north star's literal algorithm, kept as a measured comparison (TD_SOLVER=eps): Bertsekas
// eps-scaling Jacobi auction, costs scaled by n+1, eps from eps0 down to 1 by a factor theta;
// the eps = 1 phase ends with the exact optimum.  Each phase resets the assignment, keeps the
// prices and bids until every row is assigned; the host polls the free-row count every 8 rounds.
template <typename CT>
int sv_solve_eps_t(Solver &sv, long long eps0_mult, int theta, int64_t *rounds_out, int64_t *phases_out)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    const int n = sv.n;
    const long long K = (long long)n + 1;
    long long eps = eps0_mult > 0 ? K * eps0_mult : 1;
    int64_t rounds = 0, phases = 0;
    unsigned long long *keys = (unsigned long long *)sv.bid.p;
    for (;;) {
        phases++;
        k_eps_reset<PT><<<(std::max(n, (int)CTL_WORDS) + 255) / 256, 256, 0, c.stream>>>(n, sv.nrows, (PT *)sv.price.p, (int *)sv.owner.p,
                                                                                       (int *)sv.r2c.p, (int *)sv.misc.p);
        for (;;) {
            for (int r = 0; r < 8; r++) {
                k_bid<CT, false, true><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p,
                                                                               (const PT *)sv.price.p, (const int *)sv.r2c.p, keys,
                                                                               (const int *)sv.misc.p, 60, 1, K, eps);
                k_assign<PT><<<(n + 255) / 256, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, keys, (PT *)sv.price.p, (int *)sv.owner.p,
                                                                   (int *)sv.r2c.p, (int *)sv.misc.p, 60, IsNP<CT>::value ? g_np_plimit : 0ll);
            }
            rounds += 8;
            k_freelist<<<1, 1024, 0, c.stream>>>(n, (const int *)sv.r2c.p, (int *)sv.list.p, (int *)sv.misc.p);
            TD_HIP(hipMemcpyAsync(c.pinned, (int *)sv.misc.p + CTL_NFREE, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            TD_HIP(hipStreamSynchronize(c.stream));
            if (((int *)c.pinned)[0] == 0) break;
            if (rounds > 4000000) return fail(TD_EINTERNAL, "eps auction did not converge in 4M rounds");
        }
        if (eps == 1) break;
        eps = std::max<long long>(1, eps / theta);
    }
    *rounds_out = rounds;
    *phases_out = phases;
    return TD_OK;
}

// Price warm start for wide, tie-free rows (|a-b| geometry, uniform 0..10^6): the eps = 0 rounds
// stall there with a quarter of the rows free (every bid evicts somebody, increments are the tiny
// gaps between best and second best) and leave thousands of long augmenting paths to the finisher.
// A few Bertsekas auction phases with eps = range/4, /16, ... (integer eps on the unscaled costs,
// each phase cut off once 98 % of the rows are placed) bring the prices close to equilibrium in a
// few dozen streaming rounds. Only the PRICES are kept: any non-negative price vector is dual
// feasible, so exactness still rests on the eps = 0 rounds + shortest augmenting paths + the LP
// certificate that follow; the assignment of the eps phases is thrown away.
template <typename CT>
int sv_warm_t(Solver &sv, int64_t range, int64_t *rounds_out, bool *random_like = nullptr)
{
    Ctx &c = ctx();
    using PT = typename Tr<CT>::PT;
    const int n = sv.n;
    unsigned long long *keys = (unsigned long long *)sv.bid.p;
    const long long eps_last = std::max<long long>(1, range >> g_warm_bits);
    int64_t rounds = 0;
    bool first = true;
    ProfScope ps(TD_K_BID);
    // the phases bid on a sparse core of every row (td_core_warm.h): 8 MB instead of 1 GiB per full round at n = 16 384
    bool use_core = g_core && sizeof(CT) == 4 && n >= g_core_min_n && sv.nrows == n;
    bool dense_too = true;   // rows whose list does not prove their best column bid on the dense row (skipped while a group of rounds flags nobody)
    int64_t core_rounds = 0, miss_total = 0, bids_total = 0;
    bool phase_core = true;
    double t_avg = 0.0;
    int warm_div = g_warm_div;
    if (random_like) *random_like = false;
    if (g_rand_test && sizeof(CT) == 4 && n >= g_core_min_n && sv.nrows == n) {
        // random-like or metric structure (k_row_corr)?  64 row pairs; the verdict rides on the read-back the lists need anyway
        double *d_corr = (double *)((char *)sv.misc.p + 2048 + 128);   // 64 doubles, clear of the control words, the totals and t_sum
        k_row_corr<CT><<<64, 256, 0, c.stream>>>(n, sv.nchunks, (const CT *)sv.cc.p, d_corr);
        TD_HIP(hipMemcpyAsync((char *)c.pinned + 2048, d_corr, 64 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        double m = 0;
        for (int k = 0; k < 64; k++) m += ((const double *)((const char *)c.pinned + 2048))[k];
        m /= 64.0;
        if (getenv("TD_DEBUG")) fprintf(stderr, "[td] row-correlation test: mean |r| = %.3f over 64 row pairs -> %s\n", m, m < g_rand_corr ? "random-like" : "structured");
        if (m < g_rand_corr) {
            warm_div = std::max(g_warm_div, g_rand_div);
            if (random_like) *random_like = true;
        }
    }
    if (use_core) {
        int rc;
        if ((rc = ensure(sv.core, sizeof(uint2) * (size_t)n * CORE_CAP))) return rc;
        if ((rc = ensure(sv.core_n, sizeof(int) * (size_t)n))) return rc;
        if ((rc = ensure(sv.core_t, sizeof(uint32_t) * (size_t)n))) return rc;
        if ((rc = ensure(sv.core_need, (size_t)n + 64))) return rc;
        TD_HIP(hipMemsetAsync(sv.core_need.p, 0, (size_t)n, c.stream));
        unsigned long long *t_sum = (unsigned long long *)((char *)sv.misc.p + 2048 + 64);
        TD_HIP(hipMemsetAsync(t_sum, 0, sizeof(unsigned long long), c.stream));
        k_core_build<CT><<<std::min(n, c.n_cu * 8), 256, 0, c.stream>>>(n, n, sv.nchunks, g_core_k, (const CT *)sv.cc.p, (uint2 *)sv.core.p,
                                                                       (int *)sv.core_n.p, (uint32_t *)sv.core_t.p, t_sum);
        TD_HIP(hipGetLastError());
        if (g_core_mode == 1) {
            TD_HIP(hipMemcpyAsync(c.pinned, t_sum, sizeof(unsigned long long), hipMemcpyDeviceToHost, c.stream));
            TD_HIP(hipStreamSynchronize(c.stream));
            t_avg = (double)(((unsigned long long *)c.pinned)[0]) / (double)n;
        }
    }
    for (long long eps = std::max<long long>(1, range / warm_div);; eps = std::max<long long>(eps_last, eps / g_warm_theta)) {
        (void)first;
        k_eps_reset<PT><<<(std::max(n, (int)CTL_WORDS) + 255) / 256, 256, 0, c.stream>>>(n, sv.nrows, (PT *)sv.price.p, (int *)sv.owner.p,
                                                                                           (int *)sv.r2c.p, (int *)sv.misc.p);
        first = false;
        const bool trust = use_core && g_core_mode == 1;
        if (trust) phase_core = (double)eps <= g_core_eps * t_avg;
        for (int grp = 0; grp < g_warm_groups; grp++) {
            for (int r = 0; r < 8; r++) {
                if (use_core && phase_core) {
                    k_bid_core<PT><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, (const uint2 *)sv.core.p, (const int *)sv.core_n.p,
                                                                            (const uint32_t *)sv.core_t.p, (const PT *)sv.price.p,
                                                                            (const int *)sv.r2c.p, keys, (int *)sv.misc.p, eps,
                                                                            trust ? nullptr : (uint8_t *)sv.core_need.p);
                    if (dense_too && !trust)
                        k_bid<CT, false, true><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p,
                                                                                       (const PT *)sv.price.p, (const int *)sv.r2c.p, keys,
                                                                                       (const int *)sv.misc.p, 60, 1, 1, eps, nullptr,
                                                                                       (const uint8_t *)sv.core_need.p);
                    core_rounds++;
                } else
                    k_bid<CT, false, true><<<(sv.nrows + 3) / 4, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, sv.nchunks, (const CT *)sv.cc.p,
                                                                                   (const PT *)sv.price.p, (const int *)sv.r2c.p, keys,
                                                                                   (const int *)sv.misc.p, 60, 1, 1, eps);
                k_assign<PT><<<(n + 255) / 256, 256, 0, c.stream>>>(n, sv.nrows, sv.row0, keys, (PT *)sv.price.p, (int *)sv.owner.p,
                                                                   (int *)sv.r2c.p, (int *)sv.misc.p, 60, IsNP<CT>::value ? g_np_plimit : 0ll);
            }
            rounds += 8;
            k_freelist<<<1, 1024, 0, c.stream>>>(n, (const int *)sv.r2c.p, (int *)sv.list.p, (int *)sv.misc.p);
            TD_HIP(hipMemcpyAsync(c.pinned, (int *)sv.misc.p + CTL_NFREE, sizeof(int), hipMemcpyDeviceToHost, c.stream));
            if (use_core && phase_core && !trust) {
                TD_HIP(hipMemcpyAsync((int *)c.pinned + 1, (int *)sv.misc.p + CTL_COREMISS, 2 * sizeof(int), hipMemcpyDeviceToHost, c.stream));
                TD_HIP(hipMemsetAsync((int *)sv.misc.p + CTL_COREMISS, 0, 2 * sizeof(int), c.stream));
                k_price_min<PT><<<1, 1024, 0, c.stream>>>(n, (const PT *)sv.price.p, (int *)sv.misc.p);
            }
            TD_HIP(hipStreamSynchronize(c.stream));
            if (use_core && phase_core && !trust) {
                const int miss = ((int *)c.pinned)[1], bids = ((int *)c.pinned)[2];   // flagged bidders / all bidders of this group's 8 rounds
                miss_total += miss;
                bids_total += bids;
                dense_too = miss > 0;
                // most of the bidders outgrew their lists — every early phase (eps far above the lists' thresholds: price
                // differences, not costs, decide the bids) and geometric rows in crowded regions: the lists only add a launch
                // per round; the next phase tries them again
                if (phase_core && (long long)miss * 2 > (long long)bids) phase_core = false;
            }
            if (((int *)c.pinned)[0] <= (g_warm_cut > 0 ? n / g_warm_cut : 0)) break;
            // a phase on trusted lists that is not through after TD_CORE_PATIENCE groups: the lists do not hold the columns the
            // rows need (|a - b| with the recogniser off: rows of a crowded stretch need partners hundreds of ranks away and
            // their lists inflate the prices of the stretch instead) — the dense rows from here on, for good
            if (trust && phase_core && grp + 1 >= g_core_patience) {
                phase_core = false;
                use_core = false;
            }
        }
        if (use_core && !trust) dense_too = true, phase_core = true;   // a new phase: every row is free again
        if (eps <= eps_last) break;
    }
    // only the prices are kept
    k_eps_reset<PT><<<(std::max(n, (int)CTL_WORDS) + 255) / 256, 256, 0, c.stream>>>(n, sv.nrows, (PT *)sv.price.p, (int *)sv.owner.p,
                                                                                   (int *)sv.r2c.p, (int *)sv.misc.p);
    TD_HIP(hipGetLastError());
    if (getenv("TD_DEBUG")) fprintf(stderr, "[td] warm start: %lld rounds, %lld of them on the sparse core (mode %d, average reach of the lists %.0f), %lld of %lld bidders took their dense row\n",
                                    (long long)rounds, (long long)core_rounds, g_core_mode, t_avg, (long long)miss_total, (long long)bids_total);
    *rounds_out = rounds;
    return TD_OK;
}


#define TD_DISPATCH(sv, CALL, ...)                                  \
    do {                                                            \
        switch ((sv).bpc) {                                         \
            case 1: rc = CALL<uint8_t>(__VA_ARGS__); break;         \
            case 2: rc = CALL<uint16_t>(__VA_ARGS__); break;        \
            case 4: rc = CALL<uint32_t>(__VA_ARGS__); break;        \
            case 5: rc = CALL<u32n>(__VA_ARGS__); break;            \
            case 6: rc = CALL<u8e>(__VA_ARGS__); break;             \
            default: rc = fail(TD_EINVAL, "shard is not compressed yet"); \
        }                                                           \
    } while (0)

// read back ctl + totals (one sync)
int sv_readback(Solver &sv, int64_t *total, int64_t *dual, int max_rounds, int *range_flag = nullptr)
{
    Ctx &c = ctx();
    char *pin = (char *)c.pinned;
    static_assert(CTL_ALL * sizeof(int) <= 1024, "the totals sit at byte 1024 of the control block");
    TD_HIP(hipMemcpyAsync(pin, sv.misc.p, 1024 + 16, hipMemcpyDeviceToHost, c.stream));   // control words + totals: one copy
    TD_HIP(hipStreamSynchronize(c.stream));
    const int *hctl = (const int *)pin;
    if (range_flag) {
        *range_flag = hctl[CTL_FLAG];
        if (*range_flag & 4) {  // transposed formulation wanted: hand back the sampled column range
            c.stats[6] = (int64_t)(((const unsigned long long *)(hctl + CTL_SHAPE + 4))[0]);
            return TD_OK;
        }
        if (*range_flag) {  // the speculated storage width was too narrow: caller retries
            c.stats[6] = (int64_t)(((const unsigned long long *)(hctl + CTL_RANGE))[0]);
            return TD_OK;
        }
    }
    if (hctl[CTL_ERR]) return fail(TD_EINTERNAL, "device-side consistency check failed (code %d)", hctl[CTL_ERR]);
    if (total) *total = ((const int64_t *)(pin + 1024))[0];
    if (dual) *dual = ((const int64_t *)(pin + 1024))[1];
    int rounds = 0;
    for (int r = 0; r < max_rounds && r < 48; r++)
        if (hctl[CTL_PROG + r] > 0) rounds++;
    if (getenv("TD_DEBUG")) {
        fprintf(stderr, "[td] n=%d bpc=%d progress per round:", sv.n, sv.bpc);
        for (int r = 0; r < max_rounds && r < 48; r++) fprintf(stderr, " %d", hctl[CTL_PROG + r]);
        fprintf(stderr, " | parallel-sap %d | serial-sap rows %d steps %d\n", hctl[CTL_PACC], hctl[CTL_NFREE], hctl[CTL_STEPS]);
    }
    c.stats[0] = rounds;
    c.stats[1] = 0;
    c.stats[2] = hctl[CTL_NFREE];
    c.stats[3] = hctl[CTL_STEPS];
    c.stats[4] = sv.bpc == 5 ? 4 : (sv.bpc == 6 ? 1 : sv.bpc);   // bytes per stored cell (mode 5 = 4-byte cells with 32-bit prices, 6 = 1-byte cells + escape)
    c.stats[5] = hctl[CTL_PACC];
    c.stats[10] = hctl[CTL_FOREST];
    return TD_OK;
}

}  // namespace

// td_tick knows the model it hands over (how many dummy requests / cabs pad it, the fill value): the next td_assign call
// skips the line probe and the speculative 1-byte attempt and goes straight to the fused pass when the shape rule says so
namespace {
struct AssignHint {
    bool valid = false;
    int const_cols = 0, const_rows = 0, fill = 0;
    bool tick = false;   // td_tick's remainder: its own round count, no speculative batches (tick-sized tuning)
    bool gen = false;    // the cells come from position arrays (src), `cost` is a placeholder
    CellSrc src;
} g_hint;
}  // namespace
void td::assign_hint_padded(int const_cols, int const_rows, int32_t fill)
{
    g_hint = AssignHint();
    g_hint.valid = true;
    g_hint.const_cols = const_cols;
    g_hint.const_rows = const_rows;
    g_hint.fill = fill;
    g_hint.tick = true;
}

// Cost build + optimal assignment from DEVICE position arrays without the int32 matrix where the model allows it: a model
// padded with dummy requests (n_s - n_d beyond the shape rule's margin: every Simulator.java / simulate.py tick) goes
// through the fused transposing compress pass with its cells made on the fly (CellSrc), k_final sums them the same way;
// any other shape is built into a library buffer and handed to td_assign as it is.
int td::build_assign_device(const int32_t *d_cab, int n_s, const int32_t *d_dem, int n_d, const int32_t *d_dist, int S, int32_t fill,
                            int32_t threshold, bool tick, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    read_tunables();
    const int n = std::max(n_s, n_d);
    Solver &sv = *g_active;
    int rc;
    if ((rc = ensure(sv.misc, 4096))) return rc;
    const int const_cols = n - n_d, const_rows = n - n_s;
    const int margin = n / 256 > 32 ? n / 256 : 32;
    const bool fuse = g_fuse_gen && g_fuse_t && g_shape && !g_solver_eps && n >= 64 && const_cols >= 16 && const_cols - const_rows >= margin &&
                      fill >= 255 && (int64_t)fill <= NP_RANGE;
    if (!fuse) {
        if ((rc = ensure(sv.genbuf, sizeof(int32_t) * (size_t)n * n))) return rc;
        if ((rc = td::cost_build_async(d_cab, n_s, d_dem, n_d, d_dist, S, fill, threshold, (int32_t *)sv.genbuf.p))) return rc;
        if (tick) td::assign_hint_padded(const_cols, const_rows, fill);
        return td_assign(n, (const int32_t *)sv.genbuf.p, row_to_col, total, dual_bound);
    }
    g_hint = AssignHint();
    g_hint.valid = true;
    g_hint.const_cols = const_cols;
    g_hint.const_rows = const_rows;
    g_hint.fill = fill;
    g_hint.tick = tick;
    g_hint.gen = true;
    g_hint.src.cab_to = d_cab;
    g_hint.src.dem_from = d_dem;
    g_hint.src.dist = d_dist;
    g_hint.src.n_s = n_s;
    g_hint.src.n_d = n_d;
    g_hint.src.S = S;
    g_hint.src.fill = fill;
    g_hint.src.thr = threshold;
    return td_assign(n, (const int32_t *)sv.misc.p /* placeholder: no matrix exists */, row_to_col, total, dual_bound);
}

// =====================================================================================
// td_assign
// =====================================================================================
namespace {
int assign_impl(Solver &sv, int n, const int32_t *cost, int32_t *row_to_col, int64_t *total, int64_t *dual_bound);
}  // namespace

extern "C" int td_assign(int n, const int32_t *cost, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    return assign_impl(*g_active, n, cost, row_to_col, total, dual_bound);
}

namespace {
int assign_impl(Solver &sv, int n, const int32_t *cost, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    // the hint of td_tick is for THIS call only, whatever way the call ends (an early return left it for the
    // next, unrelated call)
    const AssignHint hint = g_hint;
    g_hint.valid = false;
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    read_tunables();
    if (n < 0) return fail(TD_EINVAL, "n < 0");
    if (n == 0) {  // solver.py:12  "if n==0: return 0, []"
        if (total) *total = 0;
        if (dual_bound) *dual_bound = 0;
        return TD_OK;
    }
    if (!cost || !row_to_col) return fail(TD_EINVAL, "null array");
    if (n >= (1 << ROW_BITS) - 1) return fail(TD_ERANGE, "n=%d exceeds the packed bid key", n);
    sv.skip = nullptr;
    sv.fused_t = false;
    sv.fused8 = false;
    int rc;
    if ((rc = sv_prepare(sv, n, 0, n, cost))) return rc;
    int64_t tot = 0, dual = 0;
    if (n == 1) {
        TD_HIP(hipMemcpyAsync(c.pinned, sv.d_cost, sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
        TD_HIP(hipStreamSynchronize(c.stream));
        tot = dual = ((int32_t *)c.pinned)[0];
        int32_t zero = 0;
        if (is_device_ptr(row_to_col)) {
            TD_HIP(hipMemcpyAsync(row_to_col, &zero, sizeof(int32_t), hipMemcpyHostToDevice, c.stream));
            TD_HIP(hipStreamSynchronize(c.stream));
        } else
            row_to_col[0] = 0;
        if (total) *total = tot;
        if (dual_bound) *dual_bound = dual;
        return TD_OK;
    }
    // The reference's own distance table is a line (greedy_opt.py:122-127): td_line.hip tries the sorted matching
    // and keeps it only if its certificate pass proves it optimal on this matrix.  The probe is queued in
    // front of the first compress pass and its verdict is awaited while that pass runs, so a refusal costs
    // the probe kernel alone; when the probe says "plausible" the compress pass returns at once (sv.skip).
    bool line_pending = false, early_check = false, r2c_in_pinned = false;
    constexpr size_t R2C_PIN_OFF = 8192;   // clear of the control words (0..), the totals (1024) and the probe verdict (4096)
    c.stats[8] = c.stats[9] = 0;
    const int hint_margin = n / 256 > 32 ? n / 256 : 32;
    const bool hinted_fuse = hint.valid && g_fuse_t && g_shape && !g_solver_eps && n >= 64 && hint.const_cols >= 16 &&
                             hint.const_cols - hint.const_rows >= hint_margin && hint.fill >= 255 && (int64_t)hint.fill <= NP_RANGE;
    sv.gen = hint.valid && hint.gen;
    sv.gsrc = hint.src;
    // the cells of a td_build_assign call exist only as position arrays; the moment the solve leaves the fused path they are
    // written out as the int32 matrix every other pass reads
    auto materialise = [&]() -> int {
        if (!sv.gen) return TD_OK;
        int mrc;
        if ((mrc = ensure(sv.genbuf, sizeof(int32_t) * (size_t)n * n))) return mrc;
        if ((mrc = td::cost_build_async(sv.gsrc.cab_to, sv.gsrc.n_s, sv.gsrc.dem_from, sv.gsrc.n_d, sv.gsrc.dist, sv.gsrc.S, sv.gsrc.fill, sv.gsrc.thr,
                                        (int32_t *)sv.genbuf.p)))
            return mrc;
        sv.d_cost = (const int32_t *)sv.genbuf.p;
        sv.gen = false;
        return TD_OK;
    };
    if (sv.gen && !hinted_fuse && (rc = materialise())) return rc;
    if (!hinted_fuse && g_line && n >= g_line_min_n && !g_solver_eps) {
        if ((rc = line_probe_launch(n, sv.d_cost, &sv.skip))) return rc;
        line_pending = true;
    }
    c.stats[8] = 0;
    int max_rounds = g_max_rounds;
    bool solved = false, transposed = false, np_failed = false, no_fuse = false, fused_spec = false;
    int64_t range_hint = -1;
    sv.defer_const = g_defer_const && !g_solver_eps;
    sv.bid0_done = false;
    sv.tick_sized = false;
    int64_t hinted_range = -1;
    if (hinted_fuse) {   // the caller (td_tick) told the shape: the fused pass at once, speculatively (flags with the final read-back)
        bool ff = false;
        int64_t fr = 0;
        fused_spec = true;
        if (g_fuse_t >= 2 && n > 2048 && n <= 65536) {
            if ((rc = sv_compress_fused(sv, &ff, &fr, true, hint.fill, true, hint.fill))) return rc;
            sv.fused8 = ff;
        }
        if (!ff && (rc = sv_compress_fused(sv, &ff, &fr, false, 0, true, hint.fill))) return rc;
        sv.fused_t = true;
        transposed = true;
        hinted_range = hint.fill;
        max_rounds = std::min(max_rounds, (hint.tick && n < 2048) ? std::min(g_fused_rounds, g_tick_rounds) : g_fused_rounds);
        sv.tick_sized = hint.tick && n < 2048;
    }
restart:
    for (int orient = 0; orient < 2 && !solved; orient++) {
    bool want_transpose = false;
    int64_t known_range = sv.fused_t ? hinted_range : (transposed ? range_hint : -1);
    for (int bpc : {1, 2, 6, 5, 4}) {   // 5 = 4-byte cells with 32-bit prices (narrow-price mode, see u32n); 6 = 1-byte cells + escape (u8e, fused pass only)
        bool fits = false;
        if (bpc == 6 && !sv.fused8) continue;
        if (sv.fused_t && (sv.fused8 ? bpc != 6 : (bpc != 5 && bpc != 4))) continue;   // the widths the fused pass has written
        if (known_range > 254 && bpc == 1) continue;    // the probe's sampled column range: u8 cannot hold it
        if (known_range > 65534 && bpc == 2) continue;  // u16 cannot hold it either
        if (bpc == 5 && (!g_narrow_price || g_solver_eps || np_failed || known_range < 0 || known_range > NP_RANGE)) continue;
        // the packed bid key keeps (price << 20 | row): prices stay below n * range
        if (known_range >= 0 && (double)(known_range + 1) * (double)(n + 1) >= 4.0e12)
            return fail(TD_ERANGE, "row cost range %lld with n=%d overflows the packed bid key (price < 2^43)",
                        (long long)known_range, n);
        // u8 is tried speculatively (no host round trip in the common case)
        const bool spec = (bpc == 1) && g_speculate;
    compress_pass:
        if (sv.fused_t) {   // cc already holds the transposed problem (k_compress_tr); the same bytes serve both price widths
            sv.bid0_done = false;   // (a 1-byte attempt queued before the fused pass may have set it)
            sv.cc_partial = false;
            sv.zs_done = false;
            fits = true;
            sv.bpc = bpc;
            if (np_failed) k_fill_i32<<<1, 64, 0, c.stream>>>((int *)sv.misc.p + CTL_FLAG, 2, 0);   // the price-limit flag of the 32-bit attempt
        } else {
            // 1-byte attempt: state init + the shape probe ride in front of the compress pass, which writes round 0's bids
            sv.want_bid0 = spec && !g_solver_eps;
            sv.lazy_cc = true;
            sv.zs_V = (sv.want_bid0 && orient == 0) ? (g_blocks >= 0 ? g_blocks : (n >= g_blocks_min_n ? 8 : 0)) : 0;
            sv.probe = (spec && orient == 0 && g_shape && n >= 64 && n <= g_shape_max_n && !g_solver_eps) ? sv.d_cost : nullptr;
            sv.range_seen = -1;
            rc = sv_compress(sv, bpc, &fits, spec);
            sv.want_bid0 = false;
            sv.lazy_cc = false;
            sv.probe = nullptr;
            if (rc) return rc;
            // a width that fits after a mere lower bound of the range (the line probe's "row 0 is too wide for one byte"):
            // the warm start's schedule and the "is it wide" test need the measured range
            if (!spec && fits && sv.range_seen > known_range) known_range = sv.range_seen;
        }
        if (line_pending) {
            line_pending = false;
            sv.skip = nullptr;
            int mode = 0, kd = 0, accepted = 0, susp = 0, shape3[4] = {0, 0, 0, 0};
            if ((rc = line_probe_wait(&mode, &kd, &susp, shape3))) return rc;
            if (mode == 0 && susp) {
                // The probe made the speculative 1-byte pass queued behind it a no-op (its device flag): that attempt
                // was going to be void — row 0 is too wide for one byte, or the model is padded with dummy requests.
                const int margin = n / 256 > 32 ? n / 256 : 32;   // the rule of the shape probe (k_init_state), on the probe's estimate
                if (g_fuse_t && g_shape && !g_solver_eps && !no_fuse && shape3[0] >= 16 && shape3[0] - shape3[1] >= margin) {
                    bool ff = false;
                    int64_t fr = 0;
                    // Speculative (no host round trip) when the fill value looks like one: cells in 0 .. fill, 32-bit prices.
                    // A cell that does not fit raises CTL_FLAG, every later kernel exits, and the final read-back sends the
                    // call back to the general path (restart).
                    fused_spec = g_fuse_spec && shape3[3] >= 255 && (int64_t)shape3[3] <= NP_RANGE;
                    // 1-byte cells + escape first ({small range} u {fill}: every reference model of this kind), else 4-byte cells
                    if (g_fuse_t >= 2 && n > 2048 && n <= 65536) {   // a small model is launch-bound: 64 x 1024 tiles leave most CUs idle (tick: 0.465 vs 0.432 ms)
                        if ((rc = sv_compress_fused(sv, &ff, &fr, true, shape3[3], fused_spec, shape3[3]))) return rc;
                        sv.fused8 = ff;
                    }
                    if (!ff && (rc = sv_compress_fused(sv, &ff, &fr, false, 0, fused_spec, shape3[3]))) return rc;
                    if (fused_spec) fr = shape3[3];
                    if (getenv("TD_DEBUG")) fprintf(stderr, "[td] fused transpose + compress: n=%d fits %d range %lld, %d constant rows\n", n, (int)ff, (long long)fr, sv.nconst);
                    if (ff) {
                        sv.fused_t = true;
                        transposed = true;
                        known_range = fr;
                        // thresholded models are a few huge tie classes: the rounds stop making progress after 5 - 6
                        // (profiles/r3), every further launch is ~10 us of early exits; what is left goes to the finisher
                        max_rounds = std::min(max_rounds, g_fused_rounds);
                        continue;   // on to the 4-byte widths
                    }
                    early_check = true;   // the 1-byte attempt is likely void: one look at its flags before the rounds are queued
                    goto compress_pass;   // negative cells or a range beyond 32 bits: the general path, from the 1-byte attempt
                }
                if (shape3[2]) {   // row 0 does not fit one byte per cell: skip the 1-byte attempt, the next width measures the range
                    known_range = std::max<int64_t>(known_range, 255);
                    continue;
                }
                early_check = true;
                goto compress_pass;   // only a hint that did not lead anywhere: run the skipped pass
            }
            if (getenv("TD_DEBUG")) fprintf(stderr, "[td] line probe: n=%d verdict %d, %d constant rows, suspicious %d\n", n, mode, kd, susp);
            if (mode) {
                const int32_t *res = nullptr;
                bool line_t = false;
                if (mode == 2) {
                    // constant trailing COLUMNS (more cabs than requests): the same model on the transpose
                    if ((rc = ensure(sv.tbuf, sizeof(int32_t) * (size_t)n * n))) return rc;
                    k_transpose<<<dim3((n + 63) / 64, (n + 63) / 64), 256, 0, c.stream>>>(n, sv.d_cost, (int32_t *)sv.tbuf.p);
                    TD_HIP(hipGetLastError());
                    const long long *unused = nullptr;
                    if ((rc = line_probe_launch(n, (const int32_t *)sv.tbuf.p, &unused))) return rc;
                    if ((rc = line_probe_wait(&mode, &kd))) return rc;
                    if (getenv("TD_DEBUG")) fprintf(stderr, "[td] line probe on the transpose: verdict %d, %d constant rows\n", mode, kd);
                    line_t = true;
                }
                if (mode == 1 || mode == 3) {
                    if ((rc = line_finish(n, mode == 3 ? kd : 0, line_t ? (const int32_t *)sv.tbuf.p : sv.d_cost, &res, &tot, &accepted)))
                        return rc;
                }
                if (accepted) {
                    for (int k = 0; k < 16; k++) c.stats[k] = 0;
                    c.stats[4] = 4;
                    c.stats[7] = line_t ? 1 : 0;
                    c.stats[8] = 1;
                    c.stats[9] = mode == 3 ? kd : 0;
                    if (line_t) {   // res[column] = row of the caller's matrix: invert
                        k_r2c_from_owner<<<(n + 255) / 256, 256, 0, c.stream>>>(n, n, 0, (const int *)res, (int *)sv.r2c.p);
                        TD_HIP(hipGetLastError());
                        res = (const int32_t *)sv.r2c.p;
                    }
                    TD_HIP(hipMemcpyAsync(row_to_col, res, sizeof(int32_t) * (size_t)n,
                                          is_device_ptr(row_to_col) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c.stream));
                    TD_HIP(hipStreamSynchronize(c.stream));
                    if (total) *total = tot;
                    if (dual_bound) *dual_bound = tot;   // the certificate IS the dual bound (sum of row minima + prices)
                    return TD_OK;
                }
                goto compress_pass;   // plausible but not proven: the pass above was skipped, run it now
            }
        }
        if (!fits) {
            known_range = c.stats[6];
            continue;
        }
        // shape probe (inside k_init_state): may ask (CTL_FLAG bit 2) for the transposed formulation;
        // like a failed width speculation this costs one empty pass through the early-exiting kernels
        sv.probe = (spec && orient == 0 && g_shape && n >= 64 && n <= g_shape_max_n && !g_solver_eps) ? sv.d_cost : nullptr;
        rc = TD_OK;
        if (!sv.bid0_done) TD_DISPATCH(sv, sv_begin_t, sv);
        sv.probe = nullptr;
        if (rc) return rc;
        if (spec && early_check) {
            // the line probe saw a row too wide for one byte or a constant last column: this speculative attempt
            // will most likely be void (width flag, or the shape probe asking for the transpose).  One round trip
            // now instead of the ~30 launches of rounds and finishers that would exit at once.
            early_check = false;
            int ef = 0;
            if ((rc = sv_readback(sv, nullptr, nullptr, 0, &ef))) return rc;
            if (ef & 4) {
                want_transpose = true;
                range_hint = c.stats[6] > 254 ? c.stats[6] : -1;
                break;
            }
            if (ef) {
                known_range = c.stats[6];
                continue;
            }
        }
        int64_t warm_rounds = 0;
        if (g_solver_eps) {
            int64_t er = 0, ep = 0;
            TD_DISPATCH(sv, sv_solve_eps_t, sv, g_eps0_mult, g_eps_theta, &er, &ep);
            if (rc) return rc;
            TD_DISPATCH(sv, sv_totals_t, sv, false);
            if (rc) return rc;
            int flag = 0;
            if ((rc = sv_readback(sv, &tot, &dual, 0, spec ? &flag : nullptr))) return rc;
            if (spec && flag) {
                known_range = c.stats[6];
                continue;
            }
            dual = tot;   // exact by the eps < 1/n argument; no certificate pass in this mode
            c.stats[0] = er;
            c.stats[1] = ep;
            solved = true;
            break;
        }
        int round_cap = max_rounds;
        auto rounds = [&](bool first) -> int {
            for (int r = 0; r < round_cap; r++) {
                rc = TD_OK;
                if (!(r == 0 && first && sv.bid0_done)) TD_DISPATCH(sv, sv_bid_t, sv, r, (unsigned long long *)sv.bid.p);
                if (rc) return rc;
                TD_DISPATCH(sv, sv_apply_t, sv, r, (unsigned long long *)sv.bid.p);
                if (rc) return rc;
            }
            return TD_OK;
        };
        bool random_like = false;
        auto warm = [&](int bits) -> int {
            const int keep = g_warm_bits;
            g_warm_bits = bits;
            if (bpc == 2) rc = sv_warm_t<uint16_t>(sv, known_range, &warm_rounds, &random_like);
            else if (bpc == 5) rc = sv_warm_t<u32n>(sv, known_range, &warm_rounds, &random_like);
            else rc = sv_warm_t<uint32_t>(sv, known_range, &warm_rounds, &random_like);
            g_warm_bits = keep;
            return rc;
        };
        bool all_placed = false;
        if (sv.zs_done) {
            // block-local start (td_blocks.h): zero cells of the diagonal blocks, then one two-hop pass over the whole
            // matrix for what the blocks left; one small read-back tells whether anything is left for the rounds and the
            // finisher (on tie-heavy instances nothing is: ~30 launches that would all exit at once are not made)
            // (the same TD_HOP_PASSES two-hop passes inside the blocks as the sharded solve makes: the two sequences are the
            // same up to here, and identical altogether when the blocks leave nothing)
            const int passes_a = g_hop_passes;
            if ((rc = sv_phase_a(sv, passes_a))) return rc;
            if (passes_a > 0) {
                int *pin = (int *)c.pinned;
                auto read_left = [&]() -> int {
                    TD_HIP(hipMemcpyAsync(pin, (char *)sv.hop.p + offsetof(HopCtl, left), sizeof(int), hipMemcpyDeviceToHost, c.stream));
                    TD_HIP(hipMemcpyAsync(pin + 1, (int *)sv.misc.p + CTL_FLAG, sizeof(int), hipMemcpyDeviceToHost, c.stream));
                    TD_HIP(hipStreamSynchronize(c.stream));
                    return TD_OK;
                };
                if ((rc = read_left())) return rc;
                if (getenv("TD_DEBUG")) fprintf(stderr, "[td] after the block-local start: %d rows free, flag %d\n", pin[0], pin[1]);
                if (pin[0] > 0 && pin[1] == 0 && g_hop_global && pin[0] <= g_hop_max_rows) {
                    // (nothing has moved a price yet: "tight" is "zero cell" over the whole row too, and the pass needs no row minimum)
                    if ((rc = sv_hop_t<uint8_t>(sv, n, n, 0, 1, true, nullptr, sv.cc_partial))) return rc;
                    if ((rc = read_left())) return rc;
                    if (getenv("TD_DEBUG")) fprintf(stderr, "[td] after the two-hop pass over the whole matrix: %d rows free\n", pin[0]);
                }
                all_placed = pin[0] == 0 && pin[1] == 0;
                round_cap = std::min(max_rounds, g_zs_global_rounds);
            }
        }
        if (!all_placed && (rc = sv_complete_cc(sv))) return rc;   // the rounds and the finishers read whole rows of the narrow copy
        if (!all_placed && (rc = rounds(true))) return rc;
        // Wide, tie-free rows (no constant rows, few rows tied at their minimum) that the eps = 0
        // rounds leave with many free rows.  One small read-back; only non-speculative attempts
        // (u16 / u32 rows), which have synchronised for the width flag already.
        bool wide = false, hard16 = false;
        if (g_warm && !g_solver_eps && !spec && bpc != 1 && sv.nconst == 0 && known_range >= g_warm_min_range) {
            k_freelist<<<1, 1024, 0, c.stream>>>(n, (const int *)sv.r2c.p, (int *)sv.list.p, (int *)sv.misc.p);
            TD_HIP(hipMemcpyAsync(c.pinned, sv.misc.p, CTL_ALL * sizeof(int), hipMemcpyDeviceToHost, c.stream));
            TD_HIP(hipStreamSynchronize(c.stream));
            const int nfree_now = ((int *)c.pinned)[CTL_NFREE], tied0 = ((int *)c.pinned)[CTL_TIED];
            // tied0 counts every 16th row (every row of a model below 1024).  No warm start below TD_WARM_MIN_N rows: a
            // thresholded 60 x 60 model ran 800 warm rounds = 8 ms for rows the serial workgroup finishes in microseconds
            // (tools/r4_outliers.py found it: the 4 sampled rows of so small a model missed its ties)
            wide = n >= g_warm_min_n && !(nfree_now < std::max(g_warm_minfree, n / 64) || (long long)tied0 * (n < 1024 ? 1 : 16) * g_warm_tie_div > n);
            c.stats[2] = nfree_now;
            hard16 = bpc == 2 && g_u16_redo_free > 0 && nfree_now >= g_u16_redo_free;
        }
        if ((wide || hard16) && bpc == 2 && g_wide_u16_n && n >= g_wide_u16_n && g_narrow_price && !np_failed && known_range >= 0 &&
            known_range <= NP_RANGE) {
            // A hard instance (the finisher will dominate) in 2-byte cells: the cooperative finisher k_sapx needs
            // n / E >= 8 * 256 chunks and the 2-byte kernels carry 64-bit prices.  Redo it as 4-byte cells with
            // 32-bit prices (the next width of the loop): one more compress pass and 12 rounds, milliseconds
            // against hundreds (2-D Manhattan grid, n = 8192: 493 -> 224 ms, n = 16 384: 848 -> 640 ms).
            continue;
        }
        if (wide) {
            // eps > 0 phases down to eps = 1 as a price warm start (sv_warm_t), the rounds once more,
            // then the dense finisher
            if ((rc = warm(g_warm_bits))) return rc;
            if (random_like) round_cap = std::max(round_cap, g_rand_rounds);   // (cheap here: a few hundred bidders; halves what is left for the forest)
            if ((rc = rounds(false))) return rc;
            for (int rep = 1; random_like && rep < g_rand_repeat; rep++) {   // more blocks of eps = 0 rounds (the per-round progress words are reused)
                TD_HIP(hipMemsetAsync((int *)sv.misc.p + CTL_PROG, 0, sizeof(int) * 64, c.stream));
                if ((rc = rounds(false))) return rc;
            }
        }
        ShardTab tab{};
        tab.p[0] = sv.cc.p;
        tab.rps = n;
        tab.count = 1;
        sv.placed = false;
        if (!all_placed) TD_DISPATCH(sv, sv_finish_t, sv, tab, (int *)sv.r2c.p);
        if (rc) return rc;
        if (sv.defer_const && !sv.placed) {
            ProfScope ps(TD_K_FINAL);
            k_place_const<<<1, 1024, 0, c.stream>>>(n, (int *)sv.r2c.p, (int *)sv.owner.p, (int *)sv.list.p, (int *)sv.pred.p,
                                                   (int *)sv.misc.p);
        }
        if (dual_bound != nullptr && (rc = sv_complete_cc(sv))) return rc;   // (k_dual reads the narrow rows)
        TD_DISPATCH(sv, sv_totals_t, sv, dual_bound != nullptr);
        if (rc) return rc;
        int flag = 0;
        // a host row_to_col of a small model rides along with the read-back (one round trip, no blocking pageable copy)
        r2c_in_pinned = !is_device_ptr(row_to_col) && (size_t)R2C_PIN_OFF + sizeof(int32_t) * (size_t)n <= c.pinned_cap;
        if (r2c_in_pinned)
            TD_HIP(hipMemcpyAsync((char *)c.pinned + R2C_PIN_OFF, transposed ? sv.owner.p : sv.r2c.p, sizeof(int32_t) * (size_t)n,
                                  hipMemcpyDeviceToHost, c.stream));
        if ((rc = sv_readback(sv, &tot, &dual, max_rounds, (spec || bpc == 5 || bpc == 6) ? &flag : nullptr))) return rc;
        c.stats[1] = warm_rounds;
        if (sv.fused_t && fused_spec && (flag & 1)) {
            // the speculative fused pass met a cell outside 0 .. fill (or, for 1-byte cells, outside base .. base + 253):
            // nothing of it can be kept
            if (getenv("TD_DEBUG")) fprintf(stderr, "[td] speculative fused pass refused (flag %d): general path\n", flag);
            if (bpc == 6 && !(flag & ~9)) {   // 1-byte cells did not fit: the same pass with 4-byte cells, checked this time
                bool ff = false;
                int64_t fr = 0;
                sv.fused8 = false;
                fused_spec = false;
                if ((rc = sv_compress_fused(sv, &ff, &fr))) return rc;
                if (ff) {
                    known_range = fr;
                    continue;
                }
            }
            no_fuse = true;
            sv.fused_t = sv.fused8 = false;
            transposed = false;
            fused_spec = false;
            max_rounds = g_max_rounds;
            early_check = true;
            if ((rc = materialise())) return rc;   // (td_build_assign: the general path reads the matrix)
            goto restart;
        }
        if (bpc == 5 && flag) {   // a price reached the 32-bit limit: the attempt is void, redo with 64-bit prices
            np_failed = true;
            continue;
        }
        if (bpc == 6 && flag) {   // the same guard for the 1-byte cells with the escape: redo the fused pass as 4-byte cells
            bool ff = false;
            int64_t fr = 0;
            sv.fused8 = false;
            if ((rc = sv_compress_fused(sv, &ff, &fr))) return rc;
            if (!ff) return fail(TD_EINTERNAL, "fused transpose pass: 4-byte cells refused after the 1-byte ones fitted");
            known_range = fr;
            continue;
        }
        if (spec && (flag & 4)) {  // many constant columns: solve the transposed problem instead
            want_transpose = true;
            range_hint = c.stats[6] > 254 ? c.stats[6] : -1;
            break;
        }
        if (spec && flag) {  // rows did not fit u8: redo with the width the recorded range needs
            known_range = c.stats[6];
            continue;
        }
        c.stats[6] = (bpc == 5) ? 1 : 0;   // 1: solved in the narrow-price mode
        solved = true;
        break;
    }
    if (!want_transpose) break;
    {
        ProfScope ps(TD_K_FINAL);
        if ((rc = ensure(sv.tbuf, sizeof(int32_t) * (size_t)n * n))) return rc;
        k_transpose<<<dim3((n + 63) / 64, (n + 63) / 64), 256, 0, c.stream>>>(n, sv.d_cost, (int32_t *)sv.tbuf.p);
        TD_HIP(hipGetLastError());
        sv.d_cost = (const int32_t *)sv.tbuf.p;
        transposed = true;
    }
    }
    if (!solved) return fail(TD_ERANGE, "row cost range exceeds 2^32-2");
    // transposed solve: its columns are the caller's rows, owner[] is the caller's row_to_col
    const void *res = transposed ? sv.owner.p : sv.r2c.p;
    c.stats[7] = transposed ? 1 : 0;
    if (r2c_in_pinned) {
        memcpy(row_to_col, (const char *)c.pinned + R2C_PIN_OFF, sizeof(int32_t) * (size_t)n);
        if (total) *total = tot;
        if (dual_bound) *dual_bound = dual;
        return TD_OK;
    }
    if (is_device_ptr(row_to_col)) {
        TD_HIP(hipMemcpyAsync(row_to_col, res, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, c.stream));
    } else {
        TD_HIP(hipMemcpyAsync(row_to_col, res, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c.stream));
    }
    TD_HIP(hipStreamSynchronize(c.stream));
    if (total) *total = tot;
    if (dual_bound) *dual_bound = dual;
    return TD_OK;
}

}  // namespace

// ---- handle-scoped solvers ("re-entrant per handle"): a td_solver owns its grow-only workspace, so a process can
// keep, say, one sized for N = 65 536 and one for ticks side by side; calls stay synchronous and one at a time, like the
// reference (td_assign itself is the solver of the library's own default handle)
struct td_solver {
    td_shard ws;
};

extern "C" int td_solver_create(td_solver **out)
{
    TD_REQUIRE_INIT();
    if (!out) return fail(TD_EINVAL, "null out");
    *out = new td_solver();
    return TD_OK;
}

extern "C" int td_solver_destroy(td_solver *h)
{
    if (!h) return TD_OK;
    if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
    h->ws.free_all();
    delete h;
    return TD_OK;
}

extern "C" int td_solver_assign(td_solver *h, int n, const int32_t *cost, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    if (!h) return fail(TD_EINVAL, "null solver");
    return assign_impl(h->ws, n, cost, row_to_col, total, dual_bound);
}

extern "C" int td_solver_build_assign(td_solver *h, const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S,
                                      int32_t fill, int32_t threshold, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    if (!h) return fail(TD_EINVAL, "null solver");
    Solver *was = g_active;
    g_active = &h->ws;
    const int rc = td_build_assign(cab_to, n_s, dem_from, n_d, dist, S, fill, threshold, row_to_col, total, dual_bound);
    g_active = was;
    return rc;
}

extern "C" int td_build_assign(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                               int32_t threshold, int32_t *row_to_col, int64_t *total, int64_t *dual_bound)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (n_s < 0 || n_d < 0) return fail(TD_EINVAL, "negative sizes n_s=%d n_d=%d", n_s, n_d);
    const int n = std::max(n_s, n_d);
    if (n == 0) {   // simulate.py:38  n == 0 -> (0, [], 0)
        if (total) *total = 0;
        if (dual_bound) *dual_bound = 0;
        return TD_OK;
    }
    if ((n_s && !cab_to) || (n_d && !dem_from) || !row_to_col) return fail(TD_EINVAL, "null array");
    if (dist && S <= 0) return fail(TD_EINVAL, "dist given but S=%d", S);
    // the (small) position arrays and a host distance table: once on the device, in a buffer of the solver's workspace
    Solver &sv = *g_active;
    int rc;
    const size_t words = (size_t)n_s + n_d + ((dist && !is_device_ptr(dist)) ? (size_t)S * S : 0) + 16;
    if ((rc = ensure(sv.gpos, sizeof(int32_t) * words))) return rc;
    int32_t *buf = (int32_t *)sv.gpos.p;
    const int32_t *d_cab = cab_to, *d_dem = dem_from, *d_dist = dist;
    if (n_s && !is_device_ptr(cab_to)) {
        TD_HIP(hipMemcpyAsync(buf, cab_to, sizeof(int32_t) * (size_t)n_s, hipMemcpyHostToDevice, c.stream));
        d_cab = buf;
    }
    if (n_d && !is_device_ptr(dem_from)) {
        TD_HIP(hipMemcpyAsync(buf + n_s, dem_from, sizeof(int32_t) * (size_t)n_d, hipMemcpyHostToDevice, c.stream));
        d_dem = buf + n_s;
    }
    if (dist && !is_device_ptr(dist)) {
        TD_HIP(hipMemcpyAsync(buf + n_s + n_d, dist, sizeof(int32_t) * (size_t)S * S, hipMemcpyHostToDevice, c.stream));
        d_dist = buf + n_s + n_d;
    }
    return td::build_assign_device(d_cab, n_s, d_dem, n_d, d_dist, S, fill, threshold, false, row_to_col, total, dual_bound);
}

extern "C" int td_set_line_metric(int on)
{
    read_tunables();
    const int was = g_line ? 1 : 0;
    g_line = on != 0;
    return was;
}

extern "C" int td_set_blocks(int blocks)
{
    read_tunables();
    const int was = g_blocks;
    g_blocks = std::max(-1, std::min(HOP_BMAX, blocks));
    return was;
}

extern "C" void td_assign_release_workspace(void)
{
    g_default.free_all();
    td::line_release_workspace();
}

// =====================================================================================
// row-sharded solve: device-side pieces; the collective is the caller's
// =====================================================================================
extern "C" {

int td_shard_create(int n, int row0, int nrows, const int32_t *cost_rows, td_shard **out)
{
    TD_REQUIRE_INIT();
    read_tunables();
    if (!out) return fail(TD_EINVAL, "null out");
    if (n < 2 || row0 < 0 || nrows < 0 || row0 + nrows > n) return fail(TD_EINVAL, "bad shard geometry n=%d row0=%d nrows=%d", n, row0, nrows);
    if (n >= (1 << ROW_BITS) - 1) return fail(TD_ERANGE, "n=%d exceeds the packed bid key", n);
    if (nrows > 0 && !cost_rows) return fail(TD_EINVAL, "null cost rows");
    td_shard *s = new td_shard();
    int rc = sv_prepare(*s, n, row0, nrows, nrows ? cost_rows : (const int32_t *)ctx().pinned);
    if (rc) {
        s->free_all();
        delete s;
        return rc;
    }
    *out = s;
    return TD_OK;
}

int td_shard_destroy(td_shard *s)
{
    TD_REQUIRE_INIT();
    if (!s) return TD_OK;
    (void)hipStreamSynchronize(ctx().stream);
    s->free_all();
    delete s;
    return TD_OK;
}

int td_shard_compress(td_shard *s, int bytes_per_cell, int *fits)
{
    TD_REQUIRE_INIT();
    if (!s || !fits) return fail(TD_EINVAL, "null argument");
    bool f = false;
    int rc = sv_compress(*s, bytes_per_cell, &f);
    *fits = f ? 1 : 0;
    // a failed pass records the largest row range exactly; a pass that fits bounds it by the width
    if (!rc && !f) s->range = std::max<int64_t>(s->range, ctx().stats[6]);
    if (!rc && f) {
        const int64_t lim = bytes_per_cell == 1 ? 254 : (bytes_per_cell == 2 ? 65534 : 0xFFFFFFFEll);
        s->fit_bound = s->fit_bound < 0 ? lim : std::min(s->fit_bound, lim);
    }
    return rc;
}

// The 1-byte attempt of the block-local start WITHOUT waiting for its width flag (as td_assign speculates): the flag stays on
// the device, every later kernel of the attempt exits on it, td_shard_state_export reports it in the segment's word 0 and the
// ranks learn it from the one all-gather.  Saves the rank a host synchronisation in front of phase A.
int td_shard_compress_spec(td_shard *s)
{
    TD_REQUIRE_INIT();
    if (!s) return fail(TD_EINVAL, "null shard");
    bool f = false;
    return sv_compress(*s, 1, &f, true);
}

// Constant rows (dummy cabs of a padded model) sit out the sharded solve as they do in td_assign: between
// td_shard_compress and td_shard_begin every rank writes its constant rows into a zeroed mask of n ints (set = 0),
// the caller SUM-all-reduces it and hands the result back (set = 1).  The rows then never bid, the finisher's rank
// skips them and gives them the columns nobody owns at the end (k-th constant row <- k-th free column).
int td_shard_const_rows(td_shard *s, int32_t *mask_full, int set)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !mask_full || !s->bpc) return fail(TD_EINVAL, "td_shard_const_rows: null argument or shard not compressed");
    if (!is_device_ptr(mask_full)) return fail(TD_EINVAL, "td_shard_const_rows: the mask must be device memory");
    if (!set) {
        if (s->nrows > 0) k_const_mask<<<(s->nrows + 255) / 256, 256, 0, c.stream>>>(s->row0, s->nrows, (const int *)s->rconst.p, mask_full);
        TD_HIP(hipGetLastError());
        return TD_OK;
    }
    int rc;
    if ((rc = ensure(s->cmask, sizeof(int) * (size_t)(s->n + 16)))) return rc;
    TD_HIP(hipMemcpyAsync(s->cmask.p, mask_full, sizeof(int) * (size_t)s->n, hipMemcpyDeviceToDevice, c.stream));
    s->have_cmask = true;
    s->defer_const = g_defer_const && !g_solver_eps;
    return TD_OK;
}

// flags bit 0: the caller runs the constant-row exchange (td_shard_const_rows) in every solve — then the 1-byte
// compress pass of a wide shard may initialise the state itself, defer its constant rows and write round 0's bids
// (k_compress_reg<.., BID0>, as td_assign does); td_shard_begin / td_shard_bid(0) / td_shard_rounds then only hand the
// prepared keys over.  Same keys bit for bit, so a sharded run stays identical to td_assign.
int td_shard_options(td_shard *s, int flags)
{
    TD_REQUIRE_INIT();
    if (!s) return fail(TD_EINVAL, "null shard");
    s->want_bid0 = (flags & 1) != 0 && !g_solver_eps;
    if (s->want_bid0) s->defer_const = g_defer_const && !g_solver_eps;
    // bit 1: block-local start (td_blocks.h) — the 1-byte compress pass writes zero-slice bids, td_shard_phase_a runs the
    // local rounds and the two-hop pass, td_shard_state_export / _import carry the ONE exchange that follows
    s->zs_V = ((flags & 3) == 3 && s->want_bid0) ? (g_blocks > 0 ? g_blocks : (g_blocks < 0 ? 8 : 0)) : 0;
    // bit 2: with the block-local start, the compress pass stores the narrow cells of the diagonal slices only; the library
    // writes the rest by itself (k_compress_rest) the first time a call needs whole rows: td_shard_bid / _rounds, td_shard_cc
    // (the source of every pointer td_shard_finish reads through), the dual bound of td_shard_total(_dev)
    s->lazy_cc = (flags & 4) != 0 && s->zs_V > 0;
    return TD_OK;
}

// 1 when the last td_shard_compress wrote the zero-slice bids of the block-local start: td_shard_phase_a is due
int td_shard_blocks_pending(td_shard *s) { return (s && s->zs_done) ? 1 : 0; }

int td_shard_phase_a(td_shard *s)
{
    TD_REQUIRE_INIT();
    if (!s) return fail(TD_EINVAL, "null shard");
    if (!s->zs_done || s->bpc != 1) return fail(TD_EINVAL, "td_shard_phase_a: the compress pass did not prepare a block-local start");
    return sv_phase_a(*s, g_hop_passes);
}

// The ONE exchange after phase A.  Every rank exports a segment of td_shard_state_words() int32 words (device memory):
//   [0] 1 = this rank's rows fit one byte  [1] 1 = phase A ran here  [2] free rows phase A left  [3] constant rows
//   [4..5] largest row range (int64)  [6..15] spare, then the owners of the rank's column slice (rows_per_shard words,
//   global row ids, -1 free) and the constant-row flags of its rows (rows_per_shard words).
// The caller all-gathers the segments (rank order) and hands the concatenation to td_shard_state_import, which fills in
// the other ranks' owner slices, the owned bits of the packed prices and the replicated constant-row mask
// (td_shard_const_rows' job), and returns summary[0..4] = {all fit, all ran phase A, free rows left in total,
// constant rows in total, largest row range}.
int td_shard_state_words(td_shard *s, int rows_per_shard) { return s ? 16 + 2 * rows_per_shard : 0; }
}   // extern "C"

namespace {
__global__ void k_state_export(int nrows, int rps, int col_lo, int fits, int ran, const int *__restrict__ owner,
                               const int *__restrict__ rconst, const HopCtl *__restrict__ hc, const int *__restrict__ ctl,
                               int32_t *__restrict__ seg)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 16) {
        int v = 0;
        if (t == 0) v = fits && ctl[CTL_FLAG] == 0;
        if (t == 1) v = ran;
        if (t == 2) v = (ran && hc) ? hc->left : -1;
        if (t == 3) v = ctl[CTL_NCONST];
        if (t == 4) v = ctl[CTL_RSEEN];
        if (t == 5) v = ctl[CTL_RSEEN + 1];
        if (t != 6) seg[t] = v;   // word 6 is the caller's (written before or after this kernel, on the same stream)
    }
    if (t < rps) {
        seg[16 + t] = (ran && t < nrows) ? owner[col_lo + t] : -1;
        seg[16 + rps + t] = (t < nrows && rconst[t]) ? 1 : 0;
    }
}

__global__ void k_state_import(int n, int world, int rps, int words, int own_rank, const int32_t *__restrict__ all,
                               int *__restrict__ owner, int32_t *__restrict__ pk, int *__restrict__ cmask, long long *__restrict__ summary)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        long long fit = 1, ran = 1, left = 0, nconst = 0, range = 0;
        for (int r = 0; r < world; r++) {
            const int32_t *h = all + (size_t)r * words;
            fit = fit && h[0];
            ran = ran && h[1];
            left += h[2] >= 0 ? h[2] : n;   // unknown (no two-hop pass counted them): assume rows are free
            nconst += h[3];
            const long long rg = (long long)(((unsigned long long)(uint32_t)h[5] << 32) | (uint32_t)h[4]);
            range = rg > range ? rg : range;
        }
        summary[0] = fit;
        summary[1] = ran;
        summary[2] = left;
        summary[3] = nconst;
        summary[4] = range;
        summary[5] = all[6];   // rank 0's spare word 6: the caller's own (solve_sharded: the line-metric attempt's plausibility word)
    }
    if (t < n) {
        const int r = t / rps, k = t - r * rps;
        const int32_t *seg = all + (size_t)r * words;
        if (r != own_rank) {
            const int o = seg[16 + k];
            owner[t] = o;
            pk[t] = (pk[t] & ~1) | (o >= 0 ? 1 : 0);
        }
        cmask[t] = seg[16 + rps + k];
    }
}
}   // namespace

extern "C" {
int td_shard_state_export(td_shard *s, int rows_per_shard, int fits, int32_t *seg)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !seg || rows_per_shard < s->nrows) return fail(TD_EINVAL, "td_shard_state_export: bad arguments");
    if (!is_device_ptr(seg)) return fail(TD_EINVAL, "td_shard_state_export: the segment must be device memory");
    const bool ran = s->state_ready && s->bpc == 1 && s->zs_V > 0;
    int col_lo = 0;
    if (ran) col_lo = (s->row0 / (s->n / s->zs_V)) * (s->n / s->zs_V);
    ProfScope ps(TD_K_ASSIGN);
    k_state_export<<<(std::max(rows_per_shard, 16) + 255) / 256, 256, 0, c.stream>>>(
        s->nrows, rows_per_shard, col_lo, fits ? 1 : 0, ran ? 1 : 0, (const int *)s->owner.p, (const int *)s->rconst.p,
        ran ? (const HopCtl *)s->hop.p : nullptr, (const int *)s->misc.p, seg);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

int td_shard_state_import(td_shard *s, int world, int rank, int rows_per_shard, const int32_t *all, int64_t *summary5)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !all || !summary5 || world < 1 || rank < 0 || rank >= world) return fail(TD_EINVAL, "td_shard_state_import: bad arguments");
    if (!is_device_ptr(all)) return fail(TD_EINVAL, "td_shard_state_import: the segments must be device memory");
    if ((long long)rows_per_shard * world < s->n) return fail(TD_EINVAL, "td_shard_state_import: rows_per_shard * world < n");
    int rc;
    if ((rc = ensure(s->cmask, sizeof(int) * (size_t)(s->n + 16)))) return rc;
    long long *sum_dev = (long long *)((char *)s->misc.p + 2048);   // clear of the control words (0..) and the totals (1024)
    const int words = 16 + 2 * rows_per_shard;
    {
        ProfScope ps(TD_K_ASSIGN);
        k_state_import<<<(s->n + 255) / 256, 256, 0, c.stream>>>(s->n, world, rows_per_shard, words, rank, all, (int *)s->owner.p,
                                                            (int32_t *)s->price.p, (int *)s->cmask.p, sum_dev);
    }
    TD_HIP(hipGetLastError());
    TD_HIP(hipMemcpyAsync(c.pinned, sum_dev, 6 * sizeof(long long), hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    for (int k = 0; k < 6; k++) summary5[k] = ((const long long *)c.pinned)[k];
    s->have_cmask = true;
    s->defer_const = g_defer_const && !g_solver_eps;
    return TD_OK;
}

int td_shard_range(td_shard *s, int64_t *range)
{
    TD_REQUIRE_INIT();
    if (!s || !range) return fail(TD_EINVAL, "null argument");
    *range = s->range > 0 ? s->range : std::max<int64_t>(0, s->fit_bound);
    return TD_OK;
}

int td_shard_begin(td_shard *s, int64_t global_range)
{
    TD_REQUIRE_INIT();
    if (!s) return fail(TD_EINVAL, "null shard");
    const int64_t range = global_range >= 0 ? global_range : (s->range > 0 ? s->range : std::max<int64_t>(0, s->fit_bound));
    // same guard as td_assign: the packed bid key keeps (price << 20 | row), prices stay below n * range;
    // the keys also travel through a signed 64-bit MAX all-reduce
    if ((double)(range + 1) * (double)(s->n + 1) >= 4.0e12)
        return fail(TD_ERANGE, "row cost range %lld with n=%d overflows the packed bid key (price < 2^43)", (long long)range, s->n);
    int rc = TD_OK;
    if (!s->bid0_done && !s->state_ready) TD_DISPATCH(*s, sv_begin_t, *s);   // (else the compress pass has initialised the state in front of itself)
    return rc;
}

int td_shard_keys_len(td_shard *s) { return s ? s->npad : 0; }

namespace {
// one bidding round of a shard; round 0 of a shard whose compress pass wrote the bids: hand them over
int shard_bid_round(td_shard *s, int round, unsigned long long *keys)
{
    int rc = TD_OK;
    if ((rc = sv_complete_cc(*s))) return rc;   // a lazy narrow copy (td_shard_options bit 2): the rounds read whole rows
    if (round == 0 && s->bid0_done) {
        const size_t bytes = sizeof(unsigned long long) * (size_t)s->npad;
        TD_HIP(hipMemcpyAsync(keys, s->bid.p, bytes, hipMemcpyDeviceToDevice, ctx().stream));
        TD_HIP(hipMemsetAsync(s->bid.p, 0, bytes, ctx().stream));   // the finisher's speculative batches use this buffer as zeroed scratch
        return TD_OK;
    }
    TD_DISPATCH(*s, sv_bid_t, *s, round, keys);
    return rc;
}
}  // namespace

int td_shard_bid(td_shard *s, int round, uint64_t *keys)
{
    TD_REQUIRE_INIT();
    if (!s || !keys) return fail(TD_EINVAL, "null argument");
    if (!is_device_ptr(keys)) return fail(TD_EINVAL, "bid keys must be device memory");
    return shard_bid_round(s, round, (unsigned long long *)keys);
}

int td_shard_apply(td_shard *s, int round, uint64_t *keys)
{
    TD_REQUIRE_INIT();
    if (!s || !keys) return fail(TD_EINVAL, "null argument");
    int rc;
    TD_DISPATCH(*s, sv_apply_t, *s, round, (unsigned long long *)keys);
    return rc;
}

// ---- the bidding rounds of a sharded solve as ONE call: bid -> RCCL MAX all-reduce -> apply, every
// round enqueued on the library's stream with no host work in between.  RCCL is opened lazily with
// dlopen (the copy the process has loaded already — torch's — else /opt/rocm's): a single-GPU user
// of the library never loads it.
struct Id128 {   // ncclUniqueId: 128 opaque bytes, passed to ncclCommInitRank BY VALUE
    char b[128];
};
namespace {
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *comm = nullptr;
    int world = 0, rank = 0;
};
RcclApi g_rccl;
constexpr int RCCL_UINT64 = 5;   // ncclUint64 (rccl.h ncclDataType_t)
constexpr int RCCL_MAX = 2;      // ncclMax   (rccl.h ncclRedOp_t)

int rccl_load()
{
    RcclApi &r = g_rccl;
    if (r.h) return TD_OK;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *nm : names)
        if (!r.h) r.h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names)
        if (!r.h) r.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!r.h) return fail(TD_EINTERNAL, "librccl.so not found: %s", dlerror());
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        r.h = nullptr;
        return fail(TD_EINTERNAL, "librccl.so lacks the nccl entry points");
    }
    return TD_OK;
}
int rccl_fail(int e, const char *what)
{
    return fail(TD_EINTERNAL, "RCCL error %d (%s) in %s", e, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?", what);
}
}  // namespace

int td_comm_unique_id(void *id128)
{
    TD_REQUIRE_INIT();
    if (!id128) return fail(TD_EINVAL, "null id");
    int rc = rccl_load();
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id128);
    return e ? rccl_fail(e, "ncclGetUniqueId") : TD_OK;
}

int td_comm_init(int world, int rank, const void *id128)
{
    TD_REQUIRE_INIT();
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(TD_EINVAL, "bad communicator arguments");
    int rc = rccl_load();
    if (rc) return rc;
    if (g_rccl.comm) {
        (void)g_rccl.CommDestroy(g_rccl.comm);
        g_rccl.comm = nullptr;
    }
    Id128 id;
    memcpy(id.b, id128, sizeof(id.b));
    const int e = g_rccl.CommInitRank(&g_rccl.comm, world, id, rank);
    if (e) return rccl_fail(e, "ncclCommInitRank");
    g_rccl.world = world;
    g_rccl.rank = rank;
    return TD_OK;
}

int td_comm_destroy(void)
{
    if (g_rccl.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_rccl.comm);
    g_rccl.comm = nullptr;
    g_rccl.world = 0;
    return TD_OK;
}

int td_shard_rounds(td_shard *s, int rounds, uint64_t *keys)
{
    TD_REQUIRE_INIT();
    if (!s || !keys || rounds < 0 || rounds > 48) return fail(TD_EINVAL, "bad arguments to td_shard_rounds");
    if (!is_device_ptr(keys)) return fail(TD_EINVAL, "bid keys must be device memory");
    if (!g_rccl.comm) return fail(TD_ENOINIT, "td_comm_init has not been called");
    Ctx &c = ctx();
    int rc = TD_OK;
    static const bool force_ar = getenv("TD_SHARD_FORCE_AR") != nullptr;   // measurement: all-reduce with one rank too
    for (int r = 0; r < rounds; r++) {
        if ((rc = shard_bid_round(s, r, (unsigned long long *)keys))) return rc;
        if (g_rccl.world > 1 || force_ar) {
            const int e = g_rccl.AllReduce(keys, keys, (size_t)s->npad, RCCL_UINT64, RCCL_MAX, g_rccl.comm, c.stream);
            if (e) return rccl_fail(e, "ncclAllReduce");
        }
        TD_DISPATCH(*s, sv_apply_t, *s, r, (unsigned long long *)keys);
        if (rc) return rc;
    }
    return TD_OK;
}

int td_shard_cc(td_shard *s, void **ptr, uint64_t *bytes)
{
    TD_REQUIRE_INIT();
    if (!s || !s->bpc) return fail(TD_EINVAL, "shard not compressed");
    {
        const int rc = sv_complete_cc(*s);   // whoever asks for the rows (the finisher's rank, a gathered copy) gets whole rows
        if (rc) return rc;
    }
    if (ptr) *ptr = s->cc.p;
    if (bytes) *bytes = (uint64_t)s->nrows * s->nchunks * 16;
    return TD_OK;
}

// Runs the finisher on THIS rank; shard_ptrs[k] = device-visible base of shard k's compressed
// rows (own memory, hipIpc-mapped peer memory, or a gathered copy).
int td_shard_finish(td_shard *s, int world, const void *const *shard_ptrs, int rows_per_shard)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !shard_ptrs || world < 1 || world > 16) return fail(TD_EINVAL, "bad finisher arguments (world <= 16)");
    ShardTab tab{};
    for (int k = 0; k < world; k++) tab.p[k] = shard_ptrs[k];
    tab.rps = rows_per_shard;
    tab.count = world;
    int rc;
    const int n = s->n;
    if ((rc = ensure(s->r2c_full, sizeof(int) * (size_t)(n + 16)))) return rc;
    int *full = (int *)s->r2c_full.p;
    k_fill_i32<<<(n + 255) / 256, 256, 0, c.stream>>>(full, n, -1);
    k_r2c_from_owner<<<(n + 255) / 256, 256, 0, c.stream>>>(n, n, 0, (const int *)s->owner.p, full);
    const bool deferred = s->defer_const && s->have_cmask;
    if (deferred) k_mark_const<<<1, 1024, 0, c.stream>>>(n, (const int *)s->cmask.p, full, (int *)s->misc.p);
    s->placed = false;
    TD_DISPATCH(*s, sv_finish_t, *s, tab, full);
    if (rc) return rc;
    if (deferred && !s->placed)
        k_place_const<<<1, 1024, 0, c.stream>>>(n, full, (int *)s->owner.p, (int *)s->list.p, (int *)s->pred.p, (int *)s->misc.p);
    // the finisher moved assignments: rebuild this rank's local row_to_col from owner[]
    if (s->nrows > 0) {
        k_fill_i32<<<(s->nrows + 255) / 256, 256, 0, c.stream>>>((int *)s->r2c.p, s->nrows, -1);
        k_r2c_from_owner<<<(n + 255) / 256, 256, 0, c.stream>>>(n, s->nrows, s->row0, (const int *)s->owner.p, (int *)s->r2c.p);
    }
    TD_HIP(hipGetLastError());
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

// Every row that bids has its column (td_shard_state_import's summary[2] == 0) but the model has deferred constant rows:
// each rank gives them the columns nobody owns, k-th constant row <- k-th free column, from the REPLICATED owner[] and
// constant-row mask — the same placement on every rank, no finisher, no exchange.
int td_shard_place_const(td_shard *s)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !s->have_cmask) return fail(TD_EINVAL, "td_shard_place_const: no constant-row mask (td_shard_state_import / td_shard_const_rows first)");
    const int n = s->n;
    int rc;
    if ((rc = ensure(s->r2c_full, sizeof(int) * (size_t)(n + 16)))) return rc;
    int *full = (int *)s->r2c_full.p;
    k_fill_i32<<<(n + 255) / 256, 256, 0, c.stream>>>(full, n, -1);
    k_r2c_from_owner<<<(n + 255) / 256, 256, 0, c.stream>>>(n, n, 0, (const int *)s->owner.p, full);
    k_mark_const<<<1, 1024, 0, c.stream>>>(n, (const int *)s->cmask.p, full, (int *)s->misc.p);
    k_place_const<<<1, 1024, 0, c.stream>>>(n, full, (int *)s->owner.p, (int *)s->list.p, (int *)s->pred.p, (int *)s->misc.p);
    if (s->nrows > 0) {
        k_fill_i32<<<(s->nrows + 255) / 256, 256, 0, c.stream>>>((int *)s->r2c.p, s->nrows, -1);
        k_r2c_from_owner<<<(n + 255) / 256, 256, 0, c.stream>>>(n, s->nrows, s->row0, (const int *)s->owner.p, (int *)s->r2c.p);
    }
    TD_HIP(hipGetLastError());
    return TD_OK;
}

// owner[] (n ints, global row per column): get it after the finisher / set it on the other ranks
int td_shard_owner(td_shard *s, int32_t *owner, int set)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !owner) return fail(TD_EINVAL, "null argument");
    const size_t bytes = sizeof(int32_t) * (size_t)s->n;
    if (set) {
        TD_HIP(hipMemcpyAsync(s->owner.p, owner, bytes, is_device_ptr(owner) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c.stream));
        if (s->nrows > 0) {
            k_fill_i32<<<(s->nrows + 255) / 256, 256, 0, c.stream>>>((int *)s->r2c.p, s->nrows, -1);
            k_r2c_from_owner<<<(s->n + 255) / 256, 256, 0, c.stream>>>(s->n, s->nrows, s->row0, (const int *)s->owner.p, (int *)s->r2c.p);
        }
        TD_HIP(hipGetLastError());
    } else {
        TD_HIP(hipMemcpyAsync(owner, s->owner.p, bytes, is_device_ptr(owner) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c.stream));
    }
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

// partial total (and dual bound) over this shard's rows; the caller sums over ranks
int td_shard_total(td_shard *s, int64_t *partial_total, int64_t *partial_dual)
{
    TD_REQUIRE_INIT();
    if (!s) return fail(TD_EINVAL, "null shard");
    int rc;
    if (partial_dual != nullptr && (rc = sv_complete_cc(*s))) return rc;
    TD_HIP(hipMemsetAsync((char *)s->misc.p + 1024, 0, 16, ctx().stream));
    TD_DISPATCH(*s, sv_totals_t, *s, partial_dual != nullptr);
    if (rc) return rc;
    return sv_readback(*s, partial_total, partial_dual, g_max_rounds);
}

// The same without a host round trip: {partial total, partial dual bound, device-side error / void-attempt flags} as three
// int64 words in DEVICE memory, queued on the library's stream.  The caller SUM-all-reduces the three words and reads them
// once (word 2 != 0 on any rank: an error) — td_shard_total's read-back, the host-to-device copy of its result and the
// read-back of the reduced value were three synchronisations per solve.
int td_shard_total_dev(td_shard *s, int64_t *out3, int want_dual)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !out3) return fail(TD_EINVAL, "null argument");
    if (!is_device_ptr(out3)) return fail(TD_EINVAL, "td_shard_total_dev: the three words must be device memory");
    int rc;
    if (want_dual && (rc = sv_complete_cc(*s))) return rc;
    TD_HIP(hipMemsetAsync((char *)s->misc.p + 1024, 0, 16, c.stream));
    TD_DISPATCH(*s, sv_totals_t, *s, want_dual != 0);
    if (rc) return rc;
    k_pack_totals<<<1, 64, 0, c.stream>>>((const long long *)((const char *)s->misc.p + 1024), (const int *)s->misc.p, (long long *)out3);
    TD_HIP(hipGetLastError());
    return TD_OK;
}

// final column prices (n x int64): get them on the finisher's rank / set them on the others so
// that every rank evaluates its part of the dual bound with the same prices
int td_shard_price(td_shard *s, int64_t *price, int set)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || !price || !s->bpc) return fail(TD_EINVAL, "bad argument");
    if (!is_device_ptr(price)) return fail(TD_EINVAL, "price buffer must be device memory");
    const int g = (s->n + 255) / 256;
    if (s->bpc == 1)
        k_price_io<int32_t><<<g, 256, 0, c.stream>>>(s->n, (int32_t *)s->price.p, (long long *)price, set);
    else
        k_price_io<int64_t><<<g, 256, 0, c.stream>>>(s->n, (int64_t *)s->price.p, (long long *)price, set);
    TD_HIP(hipGetLastError());
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

int td_shard_row_to_col(td_shard *s, int32_t *r2c_local)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!s || (!r2c_local && s->nrows)) return fail(TD_EINVAL, "null argument");
    if (s->nrows == 0) return TD_OK;
    TD_HIP(hipMemcpyAsync(r2c_local, s->r2c.p, sizeof(int32_t) * (size_t)s->nrows,
                          is_device_ptr(r2c_local) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

// ---- peer mapping of shards over xGMI (one process per GPU => hipIpc) -----------------
int td_ipc_export(const void *dev_ptr, void *handle64)
{
    TD_REQUIRE_INIT();
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle size");
    if (!dev_ptr || !handle64) return fail(TD_EINVAL, "null argument");
    hipIpcMemHandle_t h;
    TD_HIP(hipIpcGetMemHandle(&h, const_cast<void *>(dev_ptr)));
    memset(handle64, 0, 64);
    memcpy(handle64, &h, sizeof(h));
    return TD_OK;
}

int td_ipc_open(const void *handle64, void **dev_ptr)
{
    TD_REQUIRE_INIT();
    if (!handle64 || !dev_ptr) return fail(TD_EINVAL, "null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    TD_HIP(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return TD_OK;
}

int td_ipc_close(void *dev_ptr)
{
    TD_REQUIRE_INIT();
    if (!dev_ptr) return TD_OK;
    TD_HIP(hipIpcCloseMemHandle(dev_ptr));
    return TD_OK;
}

int td_memcpy(void *dst, const void *src, uint64_t bytes)
{
    TD_REQUIRE_INIT();
    Ctx &c = ctx();
    if (!dst || !src) return fail(TD_EINVAL, "null argument");
    TD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c.stream));
    TD_HIP(hipStreamSynchronize(c.stream));
    return TD_OK;
}

}  // extern "C"
