/*
 * taxidispatcher_amd.h — C ABI of the MI355X (gfx950) cab<->request assignment path.
 *
 * This is the drop-in boundary for ONE hot path of boguszjelinski/taxidispatcher:
 *     cost-matrix build  ->  (optional) LCM greedy pre-reduce  ->  optimal N x N assignment
 * i.e. what the reference hands to cvxopt.glpk.ilp.  Nothing native exists in the reference
 * for this path (its three C files are pool finders), so each entry point cites the
 * reference *function* it replaces; the Python binding a maintainer would add is shown in
 * INTEGRATION.md and shipped in taxidispatcher_amd/_ffi.py.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `void*` stream is a hipStream_t.
 *   - every data pointer may be a HOST pointer or a DEVICE pointer of the active device;
 *     the library detects which (hipPointerGetAttributes) and stages host buffers itself.
 *   - caller owns every buffer; the library keeps no pointer past return.
 *   - return 0 on success, a negative TD_E* code on failure; td_last_error() gives the text.
 *     No exceptions cross the ABI.
 *   - single-threaded like the reference (one solve at a time per process); calls are
 *     synchronous unless the name ends in _async.
 *   - cost matrices are row-major int32, row = cab (supply), column = request (demand),
 *     exactly the reference's `cost[cab][cust]` / linear index n*cab+cust
 *     (solver.py:13, greedy_opt.py:22-24, Simulator.java:378-380).
 */
#ifndef TAXIDISPATCHER_AMD_H
#define TAXIDISPATCHER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define TD_API __attribute__((visibility("default")))
#else
#define TD_API
#endif

#define TD_OK 0
#define TD_EINVAL (-1)   /* bad argument */
#define TD_EHIP (-2)     /* a HIP runtime call failed */
#define TD_ENOINIT (-3)  /* td_init not called */
#define TD_ERANGE (-4)   /* cost range would overflow the solver's integer arithmetic */
#define TD_EINTERNAL (-5)

/* ---- life cycle -------------------------------------------------------------------- */
TD_API int td_init(int device);            /* select GPU, create stream + workspace */
TD_API void td_shutdown(void);
TD_API const char *td_last_error(void);
/* Stream rule: every entry point enqueues on ONE stream (the library's own unless td_set_stream
 * gave it the caller's). Device INPUTS must be complete in that stream's order, device OUTPUTS of
 * the asynchronous entry points (td_cost_build, td_gen_uniform with a device destination) are
 * ready in that stream's order: a caller working on another stream either shares its stream
 * with td_set_stream or fences with its own synchronisation / td_synchronize. Entry points that
 * return values to the host (td_assign, td_lcm, td_pool2, td_count_sum) return after their
 * results are complete. */
TD_API int td_set_stream(void *hip_stream); /* run on the caller's stream (NULL -> library stream) */
TD_API int td_synchronize(void);
TD_API int td_version(void);

/* ---- a-2 cost-matrix build ---------------------------------------------------------
 * Replaces calculate_cost: greedy_opt.py:86-99 (fill=big_cost, threshold<0),
 * simulate.py:17-33 (threshold=DROP_TIME), Simulator.java:493-520 (same + id != -1, pass
 * ids as NULL when all valid), and — with ids — procedure.py:6-12 (cells addressed by id,
 * fill = n*n).
 *   n = max(n_s, n_d);  cost is n*n int32, pre-filled with `fill`;
 *   cost[c][d] = dist[cab_to[c]*S + dem_from[d]]   (dist == NULL  =>  |cab_to[c]-dem_from[d]|)
 *   written only when threshold < 0 or value < threshold.
 *   cab_id / dem_id: NULL => positional.  Non-NULL => a pair is skipped when either id is -1
 *   (Simulator.java:508); with by_id != 0 the cell written is cost[cab_id[c]][dem_id[d]]
 *   (procedure.py:12).
 * One GPU thread per (cab, request) pair quad; position arrays read coalesced.
 */
TD_API int td_cost_build(const int32_t *cab_to, const int32_t *cab_id, int n_s,
                  const int32_t *dem_from, const int32_t *dem_id, int n_d,
                  const int32_t *dist, int S, int32_t fill, int32_t threshold, int by_id,
                  int32_t *cost /* n*n */);

/* Row window of the same matrix: writes rows [row0, row0 + nrows) of the n x n model into
 * cost_rows (nrows * n int32).  This is how a row shard builds its block in place from the
 * replicated position arrays: every rank passes the full cab_to / dem_from (<= 256 KiB
 * each) and its own window; no communication.  With by_id the window applies to the cab ids. */
TD_API int td_cost_build_rows(const int32_t *cab_to, const int32_t *cab_id, int n_s,
                       const int32_t *dem_from, const int32_t *dem_id, int n_d,
                       const int32_t *dist, int S, int32_t fill, int32_t threshold, int by_id,
                       int row0, int nrows, int32_t *cost_rows /* nrows*n */);

/* ---- a-4 optimal assignment --------------------------------------------------------
 * Replaces solve(n, cost) at solver.py:11-27 (and the ilp call at procedure.py:27,
 * greedy_opt.py:117, simulate.py:52, heuristic.py:37): min sum c[i][j] x[i][j], every row
 * and column used exactly once.  Output is row_to_col[n] (x[n*i + row_to_col[i]] == 1);
 * td_expand_x gives the reference's n*n 0/1 vector.
 *   total       optimal objective (exact integer, equals GLPK's optimum)
 *   dual_bound  may be NULL; else an LP-duality lower bound computed on the device from the
 *               final prices: dual_bound == total certifies optimality.
 * One solver per process: td_assign works in ONE grow-only device workspace owned by the library
 * (no hipMalloc after the first call of a size), like the reference, which solves one model at a
 * time per process (Simulator.java:195-205 waits for its child).  Calls are serialised by the
 * caller; consecutive calls of any sizes are independent (tests/test_gpu_parity.py::
 * test_workspace_reuse_across_sizes).  Row shards (td_shard_*) each own a workspace, so several
 * shards may live in one process.
 */
TD_API int td_assign(int n, const int32_t *cost, int32_t *row_to_col, int64_t *total,
              int64_t *dual_bound);
/* Line-metric instances.  The reference's distance table is a line (greedy_opt.py:122-127: dist[i][j] = |i - j|),
 * so its square cost matrices are |a_i - b_j|: sorted by a and b they are Monge and the sorted matching is
 * optimal.  td_assign tries that matching first (O(n) anchor reads, one sort of 2n keys) and keeps it only
 * when ONE pass over the matrix proves it optimal on the actual cells (row minima of c[i][j] - v[j] all on
 * the matched cells; exact 64-bit integers); every other matrix goes to the general solver unchanged.
 * td_set_line_metric(0) switches the attempt off (1 = on, the default; env TD_LINE=0 does the same);
 * returns the previous setting. */
TD_API int td_set_line_metric(int on);

/* ---- handle-scoped solvers ("the library must be re-entrant per handle") -----------------------------
 * td_assign / td_build_assign solve in the library's default workspace.  A td_solver owns a workspace of its own (grow-only,
 * released by td_solver_destroy), so one process can keep several — e.g. one sized for N = 65 536 and one for ticks — without
 * one call regrowing what the other needs.  Same arguments, results and errors as td_assign / td_build_assign; calls are
 * synchronous and one at a time, like the reference (strictly single-threaded). */
typedef struct td_solver td_solver;
TD_API int td_solver_create(td_solver **out);
TD_API int td_solver_destroy(td_solver *h);
TD_API int td_solver_assign(td_solver *h, int n, const int32_t *cost, int32_t *row_to_col, int64_t *total, int64_t *dual_bound);
TD_API int td_solver_build_assign(td_solver *h, const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S,
                                  int32_t fill, int32_t threshold, int32_t *row_to_col, int64_t *total, int64_t *dual_bound);

/* ---- API #1 of the reference in ONE call: cost build + optimal assignment --------------------------------------
 * What procedure.py:5-29, greedy_opt.py:102-118 and simulate.py:36-53 expose: solve(distances, demand, cabs) builds the cost
 * matrix (td_cost_build's positional rule: cost[i][j] = dist[cab_to[i]][dem_from[j]] if below `threshold`, else `fill`;
 * rows / columns beyond n_s / n_d are `fill`; dist == NULL => |a - b|) and solves it.  Same row_to_col, total and
 * dual bound as td_cost_build + td_assign — but a model padded with dummy requests (n_s - n_d beyond the shape rule's
 * margin, fill >= 255: every Simulator.java / simulate.py tick) never exists as an int32 matrix: the fused transposing
 * compress pass makes each cell from the position arrays straight into its 1- / 4-byte working copy, and the total is
 * summed from the position arrays again (csrc/td_assign.hip: CellSrc).  Other shapes are built into a library buffer
 * and solved by td_assign.  Arrays may be host or device memory; td_tick's remainder goes through the same path. */
TD_API int td_build_assign(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                           int32_t threshold, int32_t *row_to_col, int64_t *total, int64_t *dual_bound);

/* n*n bytes of 0/1 in the reference's order i = n*cab + cust (solver.py:36-39) */
TD_API int td_expand_x(int n, const int32_t *row_to_col, uint8_t *x);

/* ---- a-5 LCM greedy pre-reduce -----------------------------------------------------
 * Replaces LCM: greedy_opt.py:61-82, simulate.py:76-98, heuristic.py:24-33,
 * Simulator.java:523-549.  Repeatedly takes the first minimum in row-major order and masks
 * its row and column.
 *   threshold >= 0 : stop before taking a cell  > threshold     (greedy_opt.py:68-69)
 *   stop_value_on  : stop before taking a cell == stop_value, and cells >= stop_value are
 *                    never candidates                              (Simulator.java:529-538)
 *   stop_size >= 0 : stop after taking when n - pairs == stop_size (Simulator.java:544-545)
 *   sum_below      : a taken cell is summed only if value < sum_below (greedy_opt.py:74)
 *   max_pairs      : capacity of rows[] / cols[] (n is always enough)
 * Outputs pairs in the order the reference takes them.
 */
TD_API int td_lcm(int n, const int32_t *cost, int32_t mask, int32_t threshold, int stop_value_on,
           int32_t stop_value, int stop_size, int64_t sum_below, int max_pairs,
           int32_t *rows, int32_t *cols, int32_t *n_pairs, int64_t *total, int32_t *last_min);

/* Row-sharded LCM: rank r owns cost rows [row0, row0 + nrows).  Per pick every shard
 * reports its smallest live cell {value, global row, column} (value = INT64_MAX: none), the caller
 * takes the minimum in the reference's order (value, row, column) over all shards — one all-gather of
 * 24 bytes per rank — applies td_lcm's stop rules and tells every shard the pick.  Pairs, total and
 * last_min equal td_lcm's.  Host driver: taxidispatcher_amd/sharded.py lcm_sharded. */
typedef struct td_lcm_shard td_lcm_shard;
TD_API int td_lcm_shard_create(int n, int row0, int nrows, const int32_t *cost_rows, int stop_value_on,
                               int32_t stop_value, td_lcm_shard **out);
TD_API int td_lcm_shard_destroy(td_lcm_shard *s);
TD_API int td_lcm_shard_local_min(td_lcm_shard *s, int64_t *out3);
TD_API int td_lcm_shard_take(td_lcm_shard *s, int row, int col);
/* The same result in ROUNDS of locally dominant cells instead of one exchange per pick (Simulator.java:523-549 takes up
 * to 700 picks per tick): a live cell that is the first minimum of its row AND of its column under the order (value, row,
 * column) is taken by the sequential greedy before anything else of its row or column, so all of them are taken at once.
 * Per round: round_colmin (per column the smallest live cell below `limit` of this shard's rows, key = (value << 32) | global
 * row as a SIGNED int64, INT64_MAX = none) -> MIN all-reduce of the n keys -> round_apply (taken[c] = the key of a row whose
 * first minimum is its column's minimum, INT64_MIN = none) -> MAX all-reduce -> round_commit (mask the taken columns
 * everywhere, retire the taken rows, re-scan the rows that lost their cached column).  The picks sorted by (value, row, column)
 * are td_lcm's pair list; the stop rules are applied to that order by the host driver (sharded.py lcm_sharded).  Vectors
 * may be host or device memory. */
TD_API int td_lcm_shard_round_colmin(td_lcm_shard *s, int64_t limit, int64_t *colmin);
TD_API int td_lcm_shard_round_apply(td_lcm_shard *s, int64_t limit, const int64_t *colmin, int64_t *taken);
TD_API int td_lcm_shard_round_commit(td_lcm_shard *s, const int64_t *taken);

/* ---- one dispatcher tick in ONE call (BASELINE configs[4]) ---------------------------------
 * Replaces, for one time step, Simulator.java:163-208 after createTempDemand / createTempSupply:
 * calculate_cost (:493-520; simulate.py:17-33) -> LCM down to `stop_size` rows (:523-549, stops on big_cost or
 * when MAX_NON_LCM rows are left; skipped when stop_size < 0 or >= n) -> removal of the matched cabs and requests
 * (analyzePairs :613-674, filter_out greedy_opt.py:32-37 / simulate.py:64-69: a compaction kernel, the pair list
 * never leaves the device for it) -> calculate_cost of the remainder -> optimal assignment (solver.py:26).
 *   cab_to[n_s], dem_from[n_d], dist (S x S or NULL = |a-b|): host or device; fill = big_cost; threshold = DROP_TIME
 *   (< 0: none).  Outputs (HOST arrays): the LCM pairs in the reference's order (capacity max(n_s, n_d)), *n_pairs,
 *   *lcm_last_min (LCM_min_val, Simulator.java:188), kept_cabs / kept_dems (may be NULL: positions of the cabs /
 *   requests left for the solver, in order), *n_rest = max of their counts, row_to_col[n_rest] and the total of
 *   the remainder's optimal assignment (dummy cells count fill, like td_assign).  An empty model returns 0 pairs,
 *   n_rest 0.  When the LCM ran and ended on `fill` (*lcm_last_min == fill) the tick has no input for the solver,
 *   as in Simulator.java:188-189: the pairs, the kept lists and *n_rest are reported, row_to_col is left untouched and
 *   *total is 0.  row_to_col indexes the kept lists: ask for kept_cabs / kept_dems whenever it is used.
 *   The cost matrices are library buffers in HBM (td_tick_release_workspace frees them). */
TD_API int td_tick(const int32_t *cab_to, int n_s, const int32_t *dem_from, int n_d, const int32_t *dist, int S, int32_t fill,
                   int32_t threshold, int stop_size, int32_t *lcm_rows, int32_t *lcm_cols, int32_t *n_pairs,
                   int32_t *lcm_last_min, int32_t *kept_cabs, int32_t *kept_dems, int32_t *n_rest, int32_t *row_to_col,
                   int64_t *total);
TD_API void td_tick_release_workspace(void);

/* ---- f-3 pool of two (the step right before the path in every tick) -------------------
 * Replaces findPool: Simulator.java:681-758 (and pool.c:64-131): every ordered pair (A, B) of
 * requests is a candidate with cost = min(plan1, plan2) (:693-717); plans are taken in STABLE
 * order of cost (insertion order A-major, then B) and kept iff neither customer is in an earlier
 * kept plan (:729-739).  Implemented as the lowest-cost method with symmetric masking on the
 * n x n pair-cost matrix (same kernels as td_lcm).  Outputs up to n/2 plans in the reference's
 * order: cust_a[i] picks up cust_b[i]; plan[i] = 1 (CLNT_B_ENDS) iff cost1 < cost2 else 0.
 */
TD_API int td_pool2(int n, const int32_t *from, const int32_t *to, const int32_t *dist, int S,
                    int32_t *cust_a, int32_t *cust_b, int32_t *plan, int32_t *cost, int32_t *n_pairs);

/* ---- f-4 pools of up to 4 passengers ---------------------------------------------------
 * Replaces pool_n.c:101-207 (findPool / drop_customers / removeDuplicates; Pool.java:32-113 is the
 * same enumeration) for ONE first-pick-up slice [first0, first1) — the unit findpool.c:138-141 hands
 * to each of its 8 children (child t: first0 = t * (n/8 + 1), pool_n.c:243-246) — and the merge of
 * the children's lists (findpool.c:73-98,166-172).
 *   requests i = 0..n-1: from[i], to[i], max_wait[i] (pick-up path up to i may not be longer),
 *   max_loss[i] (percent a pooled ride may exceed the direct one);  dist: S x S or NULL => |a-b|.
 *   pools: max_pools records of 2k+1 ints (k pick-ups, k drop-offs, cost) in the reference's output
 *   order (stable by cost, de-duplicated);  n_happy: happy plans before de-duplication;  max_happy:
 *   capacity of the plan buffer (<= 0: 4 Mi plans; TD_ERANGE when exceeded — the reference's
 *   pool[10000] simply overflows there).
 *   k = 2, 3 or 4 passengers (TD_EINVAL otherwise: with k = 1 the reference's duplicate test compares its 4 padded slots
 *   and keeps a single pool, which is not reproduced).
 */
TD_API int td_pool_n(int k, int n, const int32_t *from, const int32_t *to, const int32_t *max_wait,
                     const int32_t *max_loss, const int32_t *dist, int S, int first0, int first1,
                     int64_t max_happy, int max_pools, int32_t *pools, int32_t *n_pools, int64_t *n_happy);
TD_API int td_pool_merge(int k, int n_requests, int n_in, const int32_t *pools_in /* n_in * (2k+1) */,
                         int sort_by_cost, int max_pools, int32_t *pools_out, int32_t *n_out);

/* ---- a-7 objective evaluation  (greedy_opt.py:21-29 count_sum) ---------------------- */
TD_API int td_count_sum(int n, const int32_t *cost, const int32_t *row_to_col, int64_t big_cost,
                 int64_t *sum, int32_t *n_real);

/* ---- a-10 synthetic instances (bench / tests) --------------------------------------
 * perf.jl:5  t = rand(lo:hi, n, n)  as a counter-based hash so that host oracle, one GPU
 * and each row shard generate identical cells: cell(i,j) = lo + mulhi32(hi32(splitmix64(
 * seed*0x100000001B3 + i*n + j)), hi-lo+1).  Writes rows [row0, row0+nrows).
 */
TD_API int td_gen_uniform(int n, uint64_t seed, int32_t lo, int32_t hi, int row0, int nrows,
                   int32_t *cost /* nrows*n */);

/* ---- multi-GPU: row-sharded solve ---------------------------------------
 * One process per GPU.  Rank r owns cost rows [row0, row0+nrows) x all n columns; prices and
 * column owners are replicated.  The exchange step between ranks (one MAX all-reduce of the
 * packed 64-bit bid keys per bidding round) is done by the CALLER with torch.distributed /
 * RCCL on the key buffer, or by the library itself (td_shard_rounds below).  Host driver:
 * taxidispatcher_amd/sharded.py.
 *   td_shard_compress   narrow working copy of the local rows (1, 2 or 4 bytes per cell); every
 *                       rank must end up with the same width (caller reduces `fits` with MIN)
 *   td_shard_const_rows constant rows (dummy cabs of a padded model, greedy_opt.py:88-90) sit out the solve as in
 *                       td_assign: after td_shard_compress, before td_shard_begin, every rank marks its constant
 *                       rows in a zeroed device mask of n ints (set = 0), the caller SUM-all-reduces the mask and
 *                       gives it back (set = 1); the finisher's rank then hands those rows the left-over columns
 *   td_shard_options    flags bit 0 = "this caller runs td_shard_const_rows in every solve": a wide shard's 1-byte
 *                       compress pass may then initialise the state, defer the constant rows and write round 0's
 *                       bids itself, as td_assign does (same keys; td_shard_bid(0) only hands them over)
 *   td_shard_bid        one Jacobi bidding round over the local free rows; writes keys[j] =
 *                       (price << 20 | global_row + 1) with atomicMax, 0 = no bid
 *   td_shard_apply      applies the globally reduced keys (identical on every rank) and zeroes them
 *   td_shard_finish     augmenting-path finisher on the calling rank; shard_ptrs[k] is the base of
 *                       shard k's compressed rows as visible from this device (own memory, peer
 *                       memory mapped with td_ipc_open over xGMI, or a gathered copy)
 *   td_shard_owner      get (set=0) / set (set=1) the replicated owner[] (n ints)
 *   td_shard_total      this shard's part of the total (and of the dual bound); caller sums
 */
typedef struct td_shard td_shard;
TD_API int td_shard_create(int n, int row0, int nrows, const int32_t *cost_rows, td_shard **out);
TD_API int td_shard_destroy(td_shard *s);
TD_API int td_shard_compress(td_shard *s, int bytes_per_cell, int *fits);
/* largest row cost range (max - min) this shard has seen so far; the caller reduces it with MAX over
 * the ranks and hands the result to td_shard_begin, which applies td_assign's TD_ERANGE guard
 * ((range + 1) * (n + 1) must stay below 4e12: the packed bid key holds price << 20 | row) */
TD_API int td_shard_range(td_shard *s, int64_t *range);
TD_API int td_shard_begin(td_shard *s, int64_t global_range /* < 0: use this shard's own */);
TD_API int td_shard_keys_len(td_shard *s);
TD_API int td_shard_bid(td_shard *s, int round, uint64_t *keys);
TD_API int td_shard_apply(td_shard *s, int round, uint64_t *keys);
TD_API int td_shard_cc(td_shard *s, void **ptr, uint64_t *bytes);
/* All bidding rounds in ONE call: per round  td_shard_bid -> RCCL MAX all-reduce of the keys -> td_shard_apply,
 * enqueued back to back on the library's stream (no host work between the rounds).  The communicator is
 * the library's own: rank 0 makes a 128-byte id (td_comm_unique_id), the caller broadcasts it with whatever
 * it has (torch.distributed), every rank calls td_comm_init.  librccl.so is dlopen'ed on first use. */
TD_API int td_comm_unique_id(void *id128);
TD_API int td_comm_init(int world, int rank, const void *id128);
TD_API int td_comm_destroy(void);
TD_API int td_shard_rounds(td_shard *s, int rounds, uint64_t *keys);
TD_API int td_shard_finish(td_shard *s, int world, const void *const *shard_ptrs, int rows_per_shard);
TD_API int td_shard_owner(td_shard *s, int32_t *owner, int set);
TD_API int td_shard_price(td_shard *s, int64_t *price /* n, device */, int set);
TD_API int td_shard_total(td_shard *s, int64_t *partial_total, int64_t *partial_dual);
/* the same into three int64 words of DEVICE memory, no host round trip: {partial total, partial dual bound, error / void-attempt
 * flags}; the caller SUM-all-reduces them and reads them once (word 2 != 0: an error on some rank) */
TD_API int td_shard_total_dev(td_shard *s, int64_t *out3 /* device */, int want_dual);
TD_API int td_shard_const_rows(td_shard *s, int32_t *mask_full, int set);
TD_API int td_shard_options(td_shard *s, int flags);
TD_API int td_shard_row_to_col(td_shard *s, int32_t *r2c_local);
/* BLOCK-LOCAL START of the sharded solve (csrc/td_blocks.h; what SURVEY 8e's "one exchange per round" costs at
 * N = 65 536 over 8 GPUs is the exchanges, not the rounds).  With td_shard_options(flags = 3) the 1-byte compress
 * pass of a shard that owns whole diagonal blocks (n / 8 rows x the same columns) writes, for every row, a bid for
 * the first ZERO cell of the row's own column slice; td_shard_phase_a then runs a few bidding rounds and a two-hop
 * augmentation pass on the zero cells of those blocks — no price moves, every pair is tight, nothing is exchanged.
 * The ranks then meet ONCE: td_shard_state_export writes this rank's segment (td_shard_state_words int32 words of
 * device memory: fits / ran / free rows left / constant rows / range, the owners of its column slice, the
 * constant-row flags of its rows), the caller all-gathers the segments in rank order, td_shard_state_import fills in
 * the other slices (owners, owned bits, the replicated constant-row mask of td_shard_const_rows) and returns
 * summary[0..5] = {all ranks fit, all ran phase A, free rows left in total, constant rows, largest row range, rank 0's segment
 * word 6 (free for the caller: solve_sharded carries the line-metric attempt's plausibility word there)}.
 * The ordinary rounds (td_shard_bid / _apply / _rounds) and td_shard_finish take what is still free; when
 * summary[2] == 0 the solve is complete.  flags bit 2 (flags = 7): the compress pass stores the 1-byte cells of the
 * diagonal slices only (all that phase A reads: 1/8 of the narrow copy); the library writes the other cells by itself
 * the first time a call needs whole rows (td_shard_bid / _rounds, td_shard_cc — where the pointers td_shard_finish reads
 * through come from —, the dual bound of td_shard_total) — never, when phase A leaves nothing.  td_assign starts the same way for n >= 12 288 (td_set_blocks),
 * so a sharded run and td_assign with the same block count stay bit-identical. */
TD_API int td_shard_compress_spec(td_shard *s);   /* the 1-byte compress pass without waiting for its width flag (it travels in the segment) */
TD_API int td_shard_blocks_pending(td_shard *s);
TD_API int td_shard_phase_a(td_shard *s);
TD_API int td_shard_state_words(td_shard *s, int rows_per_shard);
TD_API int td_shard_state_export(td_shard *s, int rows_per_shard, int fits, int32_t *seg /* device */);
TD_API int td_shard_state_import(td_shard *s, int world, int rank, int rows_per_shard, const int32_t *all /* device */,
                                 int64_t *summary6 /* host, 6 words */);
/* summary[2] == 0 with constant rows in the model: every rank places them itself from the replicated owner[] and mask
 * (k-th constant row <- k-th column nobody owns, what td_shard_finish does on its rank) — no finisher, no exchange */
TD_API int td_shard_place_const(td_shard *s);
/* diagonal blocks td_assign starts in: 0 = never, > 0 = that many (n must be a multiple of 16 * blocks, n >= 12 288),
 * -1 = by size (8 from n = 12 288 on, where the 1-byte compress pass can write the bids).  Returns the previous setting. */
TD_API int td_set_blocks(int blocks);
/* The sorted matching of td_assign's line-metric path (cost = |a_i - b_j|, perf.jl's G2 family) over ROW SHARDS.
 * Replaces the same call as td_assign (simulator.py:199 / munkres.c solve / greedy_opt.py:95), for matrices one
 * GPU cannot hold.  `ws` is td_line_shard_ws_words(n) 64-bit words of device memory owned by the caller.  Every
 * rank runs phases 0, 1, 2, 3 in order and after EACH phase SUM-all-reduces ws[*seg_off .. *seg_off + *seg_len)
 * with the other ranks (the segments are written disjointly and zero elsewhere, so the sum is the exchange; with
 * one rank there is nothing to do).  phase 0: anchors from the rank that owns row 0; 1: row keys; 2: replicated
 * sort, local matched cells and neighbours; 3: replicated prices, certificate pass over the local rows.
 * td_line_shard_result (after phase 3's exchange): *accepted = 1 when every rank's rows certify the matching
 * (then it is optimal, whatever the matrix was), *total its cost, row_to_col[0..nrows) (host or device) the
 * columns of the local rows.  *accepted = 0: use the general sharded solve (td_shard_*). */
TD_API int64_t td_line_shard_ws_words(int n);
TD_API int td_line_shard_phase(int phase, int n, int row0, int nrows, const int32_t *cost_rows, int64_t *ws, int64_t *seg_off,
                               int64_t *seg_len);
TD_API int td_line_shard_result(int n, int row0, int nrows, const int64_t *ws, int32_t *row_to_col, int64_t *total, int32_t *accepted);
TD_API int td_ipc_export(const void *dev_ptr, void *handle64);
TD_API int td_ipc_open(const void *handle64, void **dev_ptr);
TD_API int td_ipc_close(void *dev_ptr);
TD_API int td_memcpy(void *dst, const void *src, uint64_t bytes);

/* ---- profiling hooks used by bench.py ---------------------------------------------- */
#define TD_K_COST_BUILD 0
#define TD_K_GEN 1
#define TD_K_COMPRESS 2
#define TD_K_BID 3
#define TD_K_ASSIGN 4
#define TD_K_SAP 5
#define TD_K_FINAL 6
#define TD_K_LCM 7
#define TD_K_LINE 8   /* line-metric recogniser: anchors, keys, sort, prices */
#define TD_K_CERT 9   /* ... and its certificate pass over the int32 matrix */
#define TD_K_COUNT 10
TD_API int td_profile_enable(int on);                      /* HIP-event timing per kernel class */
TD_API int td_profile_get(int kernel, double *total_ms, int64_t *launches);
TD_API int td_profile_reset(void);
/* counters of the last td_assign: [0]=bid rounds, [1]=rounds of the eps > 0 price warm start,
 * [2]=free rows left to the serial finisher, [3]=its dijkstra steps, [4]=cost storage bytes per cell,
 * [5]=augmentations committed by the parallel finisher, [6]=1 when 4-byte cells were solved with 32-bit prices and labels,
 * [7]=1 when the transposed formulation was solved
 * (many constant columns, see DESIGN.md "rectangular models"),
 * [8]=1 when the matrix was recognised as a line metric: sorted matching, proven by the certificate pass
 * (then [0..6] are 0 except [4]=4, [7]=1 when it was the transpose that was recognised (constant trailing columns),
 * [9] = number of constant rows of the unbalanced model; see DESIGN.md "line-metric instances") */
TD_API int td_last_stats(int64_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif
