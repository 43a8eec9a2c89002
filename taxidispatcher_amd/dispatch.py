"""Host-side mirror of the reference's Python interface for the assignment hot path.

Same function names, argument meaning and return shapes as the reference scripts, so a user
of `procedure.py` / `greedy_opt.py` / `simulate.py` / `solver.py` can switch the import and keep
the calling code.  Every function runs on the MI355X through the C ABI (`_ffi`); nothing here
computes an assignment, a cost matrix or an LCM on the CPU.

Reference map (file:line in boguszjelinski/taxidispatcher):
    calculate_cost        greedy_opt.py:86-99, simulate.py:17-33 (drop_time), Simulator.java:493-520
    calculate_cost_by_id  procedure.py:6-12
    solve                 greedy_opt.py:102-118 / simulate.py:36-53  -> (n, x, cost)
    procedure_solve       procedure.py:5-29                           -> x
    build_assign          the same API #1 without handing the matrix back (td_build_assign)
    solve_cost            solver.py:11-27                             -> x
    LCM                   greedy_opt.py:61-82 / simulate.py:76-98
    LCM_heuristic         heuristic.py:24-33
    LCM_simulator         Simulator.java:523-549
    count_sum             greedy_opt.py:21-29
    filter_out            greedy_opt.py:32-37 (by id) / simulate.py:64-69 (by position)
    combined              greedy_opt.py:136-160
"""
import ctypes

import numpy as np

from . import _ffi

BIG_COST = 250000  # greedy_opt.py:5, simulate.py:12, Simulator.java:115


# ----------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------
def _records(rows):
    """(id, from, to) iterables -> three int32 arrays."""
    rows = list(rows)
    if not rows:
        z = np.zeros(0, np.int32)
        return z, z, z
    a = _ffi.as_i32(rows).reshape(len(rows), -1)
    return np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1]), np.ascontiguousarray(a[:, 2])


def _require_i32(t, what):
    """device tensors go to the kernels as they are: they must be int32 (numpy inputs are converted)"""
    if hasattr(t, "data_ptr") and str(getattr(t, "dtype", "")) != "torch.int32":
        raise _ffi.TdError("%s: a torch tensor handed to the library must be int32, got %s" % (what, t.dtype))


def _dist_arg(distances):
    """distances: None (=> |a-b|), an S x S array-like, or a (device_ptr, S) tuple."""
    if distances is None:
        return None, 0, None
    if isinstance(distances, tuple):
        return distances[0], int(distances[1]), None
    if hasattr(distances, "data_ptr") and getattr(distances, "is_cuda", False):
        _require_i32(distances, "distances")
        if distances.dim() != 2 or distances.shape[0] != distances.shape[1]:
            raise _ffi.TdError("distances must be a square S x S table")
        return _ffi.addr(distances), int(distances.shape[0]), distances   # addr() fences torch's stream
    d = _ffi.as_i32(distances)
    if d.ndim != 2 or d.shape[0] != d.shape[1]:
        raise _ffi.TdError("distances must be a square S x S table")
    return d.ctypes.data, int(d.shape[0]), d


def cost_build(cab_to, dem_from, distances=None, fill=BIG_COST, threshold=-1, cab_id=None, dem_id=None,
               by_id=False, out=None, sync=True):
    """Thin wrapper over td_cost_build. Returns (n, cost) with cost an int32 n x n numpy array,
    or writes into `out` (numpy array or torch CUDA tensor) and returns (n, out).
    sync=False: a device `out` is NOT waited for — only for callers whose next use of it is another call of this
    library (same stream, e.g. a tick loop: cost_build -> LCM -> assign)."""
    lib = _ffi.lib()
    cab_to = _ffi.as_i32(cab_to)
    dem_from = _ffi.as_i32(dem_from)
    n_s, n_d = int(cab_to.size), int(dem_from.size)
    n = max(n_s, n_d)
    cab_id = None if cab_id is None else _ffi.as_i32(cab_id)
    dem_id = None if dem_id is None else _ffi.as_i32(dem_id)
    dptr, S, keep = _dist_arg(distances)
    if out is None:
        out = np.empty((n, n), np.int32)
    if n:
        _ffi.check(lib.td_cost_build(_ffi.addr(cab_to), _ffi.addr(cab_id), n_s, _ffi.addr(dem_from),
                                     _ffi.addr(dem_id), n_d, dptr, S, int(fill), int(threshold), int(bool(by_id)),
                                     _ffi.addr(out)))
        if sync and getattr(out, "is_cuda", False):
            # device output: written asynchronously on the library's stream; torch works on its own
            _ffi.check(lib.td_synchronize())
    del keep
    return n, out


def assign(cost, n=None, want_dual=False):
    """Thin wrapper over td_assign. cost: n x n int32 (numpy or torch CUDA tensor).
    Returns (row_to_col int32[n], total[, dual_bound])."""
    lib = _ffi.lib()
    if isinstance(cost, np.ndarray) or not hasattr(cost, "data_ptr"):
        cost = _ffi.as_i32(cost)
    else:
        _require_i32(cost, "cost")
    if n is None:
        n = int(cost.shape[0])
    r2c = np.empty(n, np.int32)
    total = ctypes.c_int64(0)
    dual = ctypes.c_int64(0)
    _ffi.check(lib.td_assign(n, _ffi.addr(cost), _ffi.addr(r2c), ctypes.byref(total),
                             ctypes.byref(dual) if want_dual else None))
    if want_dual:
        return r2c, int(total.value), int(dual.value)
    return r2c, int(total.value)


def set_line_metric(on):
    """Switch td_assign's line-metric attempt (sorted matching + certificate pass, td_line.hip) on or off;
    returns the previous setting. On by default."""
    return bool(_ffi.lib().td_set_line_metric(1 if on else 0))


def build_assign(cab_to, dem_from, distances=None, fill=BIG_COST, threshold=-1, want_dual=False):
    """td_build_assign: cost build + optimal assignment in one call (the reference's solve(distances, demand, cabs),
    procedure.py:5-29 / greedy_opt.py:102-118 / simulate.py:36-53, without handing the matrix back).  A model padded
    with dummy requests never exists as an int32 matrix on the device.  Returns (n, row_to_col int32[n], total[, dual])."""
    lib = _ffi.lib()
    cab = cab_to if hasattr(cab_to, "data_ptr") else _ffi.as_i32(cab_to)
    dem = dem_from if hasattr(dem_from, "data_ptr") else _ffi.as_i32(dem_from)
    n_s, n_d = int(cab.shape[0]), int(dem.shape[0])
    n = max(n_s, n_d)
    dptr, S, keep = _dist_arg(distances)
    r2c = np.empty(n, np.int32)
    total, dual = ctypes.c_int64(0), ctypes.c_int64(0)
    _ffi.check(lib.td_build_assign(_ffi.addr(cab) if n_s else None, n_s, _ffi.addr(dem) if n_d else None, n_d, dptr, S, int(fill),
                                   int(threshold), _ffi.addr(r2c) if n else None, ctypes.byref(total),
                                   ctypes.byref(dual) if want_dual else None))
    del keep
    if want_dual:
        return n, r2c, int(total.value), int(dual.value)
    return n, r2c, int(total.value)


class Solver:
    """A handle-scoped solver (td_solver_*): its own grow-only workspace on the GPU, same calls and answers as assign() /
    build_assign().  Several can live in one process (re-entrant per handle); calls stay synchronous."""

    def __init__(self):
        self.lib = _ffi.lib()
        h = ctypes.c_void_p()
        _ffi.check(self.lib.td_solver_create(ctypes.byref(h)))
        self.h = h

    def assign(self, cost, n=None, want_dual=False):
        if isinstance(cost, np.ndarray) or not hasattr(cost, "data_ptr"):
            cost = _ffi.as_i32(cost)
        else:
            _require_i32(cost, "cost")
        if n is None:
            n = int(cost.shape[0])
        r2c = np.empty(n, np.int32)
        total, dual = ctypes.c_int64(0), ctypes.c_int64(0)
        _ffi.check(self.lib.td_solver_assign(self.h, n, _ffi.addr(cost), _ffi.addr(r2c), ctypes.byref(total),
                                             ctypes.byref(dual) if want_dual else None))
        return (r2c, int(total.value), int(dual.value)) if want_dual else (r2c, int(total.value))

    def build_assign(self, cab_to, dem_from, distances=None, fill=BIG_COST, threshold=-1, want_dual=False):
        cab = cab_to if hasattr(cab_to, "data_ptr") else _ffi.as_i32(cab_to)
        dem = dem_from if hasattr(dem_from, "data_ptr") else _ffi.as_i32(dem_from)
        n_s, n_d = int(cab.shape[0]), int(dem.shape[0])
        n = max(n_s, n_d)
        dptr, S, keep = _dist_arg(distances)
        r2c = np.empty(n, np.int32)
        total, dual = ctypes.c_int64(0), ctypes.c_int64(0)
        _ffi.check(self.lib.td_solver_build_assign(self.h, _ffi.addr(cab) if n_s else None, n_s, _ffi.addr(dem) if n_d else None, n_d, dptr, S,
                                                   int(fill), int(threshold), _ffi.addr(r2c) if n else None, ctypes.byref(total),
                                                   ctypes.byref(dual) if want_dual else None))
        del keep
        return (n, r2c, int(total.value), int(dual.value)) if want_dual else (n, r2c, int(total.value))

    def close(self):
        if self.h:
            self.lib.td_solver_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def expand_x(n, row_to_col):
    """row_to_col -> the reference's n*n 0/1 vector, index n*cab + cust (solver.py:36-39)."""
    x = np.zeros(n * n, np.uint8)
    if n:
        _ffi.check(_ffi.lib().td_expand_x(n, _ffi.addr(_ffi.as_i32(row_to_col)), _ffi.addr(x)))
    return x


def last_stats():
    out = (ctypes.c_int64 * 16)()
    _ffi.check(_ffi.lib().td_last_stats(out, 16))
    d = {"bid_rounds": out[0], "warm_rounds": out[1], "sap_free_rows": out[2], "sap_steps": out[3], "bytes_per_cell": out[4],
         "parallel_sap_rows": out[5], "narrow_price": out[6], "transposed": out[7], "line_metric": out[8], "line_dummies": out[9], "forest_levels": out[10]}
    return d


# ----------------------------------------------------------------------------------------
# reference-shaped API
# ----------------------------------------------------------------------------------------
def calculate_cost(distances, demand, cabs, big_cost=BIG_COST, drop_time=None):
    """greedy_opt.py:86-99. With drop_time (simulate.py:27: DROP_TIME=10) a cell is written only
    if distance < drop_time. n == 0 -> (0, 0) as simulate.py:21."""
    _, _, c_to = _records(cabs)
    _, d_frm, _ = _records(demand)
    if max(c_to.size, d_frm.size) == 0:
        return 0, 0
    return cost_build(c_to, d_frm, distances, fill=big_cost, threshold=-1 if drop_time is None else drop_time)


def calculate_cost_by_id(distances, demand, cabs):
    """procedure.py:6-12: fill n*n, cells addressed by the records' ids."""
    c_id, _, c_to = _records(cabs)
    d_id, d_frm, _ = _records(demand)
    n = max(c_to.size, d_frm.size)
    if n == 0:
        return 0, np.zeros((0, 0), np.int32)
    return cost_build(c_to, d_frm, distances, fill=n * n, threshold=-1, cab_id=c_id, dem_id=d_id, by_id=True)


def solve_cost(n, cost):
    """solver.py:11-27 solve(n, cost) -> x ; n == 0 -> (0, []) (solver.py:12)."""
    if n == 0:
        return 0, []
    r2c, _ = assign(cost, n)
    return expand_x(n, r2c)


def procedure_solve(distances, demand, cabs):
    """procedure.py:5-29 solve(distances, demand, cabs) -> x (length n*n, x[n*cab+cust] == 1)."""
    c_id, _, c_to = _records(cabs)
    d_id, d_frm, _ = _records(demand)
    n = max(c_to.size, d_frm.size)
    if n == 0:
        return np.zeros(0, np.uint8)
    ids_ok = (distances is not None and c_id.size and d_id.size and c_id.min() >= 0 and d_id.min() >= 0 and c_id.max() < n and d_id.max() < n
              and np.unique(c_id).size == c_id.size and np.unique(d_id).size == d_id.size)
    if not ids_ok:   # (ids outside 0..n-1 or repeated: the scatter of procedure.py:12 decides, build the matrix the same way)
        n, cost = calculate_cost_by_id(distances, demand, cabs)
        r2c, _ = assign(cost, n)
        return expand_x(n, r2c)
    # procedure.py:9-12 addresses the cells by the records' ids: the same matrix as the positional rule over arrays ORDERED BY ID
    # (a missing id is a stand outside the table: td_cost_build never indexes outside it, the cell stays at the fill value)
    to_by_id = np.full(n, -1, np.int32)
    frm_by_id = np.full(n, -1, np.int32)
    to_by_id[c_id] = c_to
    frm_by_id[d_id] = d_frm
    _, r2c, _ = build_assign(to_by_id, frm_by_id, distances, fill=n * n, threshold=-1)
    return expand_x(n, r2c)


def solve(distances, demand, cabs, big_cost=BIG_COST, drop_time=None):
    """greedy_opt.py:102-118 / simulate.py:36-53 -> (n, x, cost); n == 0 -> (0, [], 0)."""
    n, cost = calculate_cost(distances, demand, cabs, big_cost, drop_time)
    if n == 0:
        return 0, [], 0
    r2c, _ = assign(cost, n)
    return n, expand_x(n, r2c), cost


def _lcm(n, c, mask, threshold, stop_value_on, stop_value, stop_size, sum_below, max_pairs=None):
    lib = _ffi.lib()
    if not hasattr(c, "data_ptr"):
        c = _ffi.as_i32(c).reshape(n, n)
    rows = np.empty(max(n, 1), np.int32)
    cols = np.empty(max(n, 1), np.int32)
    k = ctypes.c_int32(0)
    tot = ctypes.c_int64(0)
    lm = ctypes.c_int32(0)
    _ffi.check(lib.td_lcm(n, _ffi.addr(c), int(mask), int(threshold), int(stop_value_on), int(stop_value),
                          int(stop_size), int(sum_below), n if max_pairs is None else int(max_pairs), _ffi.addr(rows), _ffi.addr(cols), ctypes.byref(k),
                          ctypes.byref(tot), ctypes.byref(lm)))
    return int(tot.value), rows[:k.value].copy(), cols[:k.value].copy(), int(lm.value)


def LCM(n, c, threshold=10, big_cost=BIG_COST, with_pairs=False):
    """greedy_opt.py:61-82 (THRESHOLD=10) / simulate.py:76-98 (THRESHOLD=20, with_pairs=True).
    `c` is the cab-major n x n cost (what np.array(matrix(cost).T) is in the reference).
    Returns (total_cost, allocated_supply, allocated_demand[, allocated])."""
    total, rows, cols, _ = _lcm(n, c, big_cost, threshold, 0, 0, -1, big_cost)
    if with_pairs:
        return total, list(map(int, rows)), list(map(int, cols)), list(zip(map(int, rows), map(int, cols)))
    return total, list(map(int, rows)), list(map(int, cols))


def LCM_heuristic(n, c):
    """heuristic.py:24-33: n iterations, every taken cell summed, mask value 100."""
    total, rows, cols, _ = _lcm(n, c, 100, -1, 0, 0, -1, 2**62)
    return total, list(map(int, rows)), list(map(int, cols))


def LCM_simulator(cost, max_non_lcm=600, big_cost=BIG_COST, as_arrays=False):
    """Simulator.java:523-549 -> (pairs [(cab, request)], LCM_min_val); as_arrays=True: (rows, cols, LCM_min_val)
    as int32 arrays (a tick loop that only indexes with them skips ~700 Python tuples per tick)."""
    cost_a = cost if hasattr(cost, "data_ptr") else _ffi.as_i32(cost)
    n = int(cost_a.shape[0])
    _, rows, cols, lm = _lcm(n, cost_a, big_cost, -1, 1, big_cost, max_non_lcm, big_cost)
    if as_arrays:
        return rows, cols, lm
    return list(zip(map(int, rows), map(int, cols))), lm


def tick(cab_to, dem_from, distances=None, big_cost=BIG_COST, drop_time=10, max_non_lcm=600):
    """One dispatcher tick in one C-ABI call (td_tick): calculate_cost -> LCM down to max_non_lcm rows ->
    removal of the matched cabs / requests on the device -> calculate_cost -> optimal assignment
    (Simulator.java:163-208,493-549,613-674; greedy_opt.py:32-37).  cab_to / dem_from: the stands the free cabs
    stand at / the requests start from.  Returns a dict: lcm_rows, lcm_cols (int32 arrays, the reference's pick
    order), lcm_min_val, kept_cabs, kept_dems (positions handed to the solver, in order), n_rest, row_to_col
    (int32[n_rest], indices into kept_cabs -> kept_dems; >= len(kept_dems) or a cab index >= len(kept_cabs): dummy),
    total (the remainder's optimum, dummy cells count big_cost), solved (False when the LCM ran and ended on big_cost:
    like Simulator.java:188-189 the tick then has no input for the solver; row_to_col is empty and total 0)."""
    lib = _ffi.lib()
    cab = cab_to if hasattr(cab_to, "data_ptr") else _ffi.as_i32(cab_to)
    dem = dem_from if hasattr(dem_from, "data_ptr") else _ffi.as_i32(dem_from)
    n_s, n_d = int(cab.shape[0]), int(dem.shape[0])
    n = max(n_s, n_d)
    dptr, S, keep = _dist_arg(distances)
    rows = np.empty(max(n, 1), np.int32)
    cols = np.empty(max(n, 1), np.int32)
    kc = np.empty(max(n_s, 1), np.int32)
    kd = np.empty(max(n_d, 1), np.int32)
    r2c = np.empty(max(n, 1), np.int32)
    k, lm, n2, tot = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0)
    _ffi.check(lib.td_tick(_ffi.addr(cab) if n_s else None, n_s, _ffi.addr(dem) if n_d else None, n_d, dptr, S, int(big_cost),
                           -1 if drop_time is None else int(drop_time), -1 if max_non_lcm is None else int(max_non_lcm),
                           _ffi.addr(rows), _ffi.addr(cols), ctypes.byref(k), ctypes.byref(lm), _ffi.addr(kc), _ffi.addr(kd),
                           ctypes.byref(n2), _ffi.addr(r2c), ctypes.byref(tot)))
    del keep
    kk = k.value
    lcm_ran = max_non_lcm is not None and 0 <= int(max_non_lcm) < n
    solved = n2.value > 0 and not (lcm_ran and lm.value == int(big_cost))
    return {"lcm_rows": rows[:kk], "lcm_cols": cols[:kk], "lcm_min_val": lm.value, "kept_cabs": kc[:n_s - kk],
            "kept_dems": kd[:n_d - kk], "n_rest": n2.value, "row_to_col": r2c[:n2.value if solved else 0], "total": tot.value,
            "solved": solved}


def find_pool(frm, to, distances=None):
    """Simulator.java:681-758 findPool on the GPU: requests given by their from/to stands ->
    list of (custA, custB, plan, cost) in the order the reference keeps them."""
    lib = _ffi.lib()
    frm, to = _ffi.as_i32(frm), _ffi.as_i32(to)
    n = int(frm.size)
    if n < 2:
        return []
    dptr, S, keep = _dist_arg(distances)
    a = np.empty(n // 2 + 1, np.int32)
    b = np.empty(n // 2 + 1, np.int32)
    plan = np.empty(n // 2 + 1, np.int32)
    cost = np.empty(n // 2 + 1, np.int32)
    k = ctypes.c_int32(0)
    _ffi.check(lib.td_pool2(n, _ffi.addr(frm), _ffi.addr(to), dptr, S, _ffi.addr(a), _ffi.addr(b), _ffi.addr(plan),
                            _ffi.addr(cost), ctypes.byref(k)))
    del keep
    return [(int(a[i]), int(b[i]), int(plan[i]), int(cost[i])) for i in range(k.value)]


def find_pool_n(k, demand, distances=None, child=None, children=8, max_happy=0):
    """pool_n.c on the GPU.  demand: rows (id, from, to, max_wait, max_loss) like the reference's
    demand file (pool_n.c:18,29-54; the id column is not used, requests are addressed by position).
    child = t: only the first-pick-up slice findpool.c gives its child t of `children`
    (pool_n.c:243-246); None: all requests as first pick-up.  Returns (int32 array [m, 2k+1] of
    pick-ups, drop-offs, cost in the reference's output order, number of happy plans)."""
    lib = _ffi.lib()
    if not 2 <= int(k) <= 4:
        raise _ffi.TdError("pool size %d (2..4)" % int(k))
    d = _ffi.as_i32(np.asarray(demand).reshape(len(demand), -1))
    n = int(d.shape[0])
    frm, to, wait, loss = (np.ascontiguousarray(d[:, c]) for c in (1, 2, 3, 4))
    if child is None:
        first0, first1 = 0, n
    else:
        step = n // children + 1
        first0 = step * child
        first1 = min(n, first0 + step)
    dptr, S, keep = _dist_arg(distances)
    cap = max(1, n // k + 1)
    out = np.zeros((cap, 2 * k + 1), np.int32)
    m = ctypes.c_int32(0)
    nh = ctypes.c_int64(0)
    _ffi.check(lib.td_pool_n(int(k), n, _ffi.addr(frm), _ffi.addr(to), _ffi.addr(wait), _ffi.addr(loss), dptr, S, int(first0),
                             int(first1), int(max_happy), cap, _ffi.addr(out), ctypes.byref(m), ctypes.byref(nh)))
    del keep
    return out[:m.value].copy(), int(nh.value)


def merge_pools(k, n_requests, lists):
    """findpool.c:73-98,166-172: the children's lists (in child order) merged into one list of pools
    that share no request; sorted by cost first for 4-passenger pools only, as the reference does."""
    lib = _ffi.lib()
    parts = [np.asarray(x, np.int32).reshape(-1, 2 * k + 1) for x in lists]
    allp = np.ascontiguousarray(np.concatenate(parts, 0)) if parts else np.zeros((0, 2 * k + 1), np.int32)
    cap = max(1, n_requests // k + 1)
    out = np.zeros((cap, 2 * k + 1), np.int32)
    m = ctypes.c_int32(0)
    _ffi.check(lib.td_pool_merge(int(k), int(n_requests), int(allp.shape[0]), _ffi.addr(allp) if allp.size else None,
                                 1 if k == 4 else 0, cap, _ffi.addr(out), ctypes.byref(m)))
    return out[:m.value].copy()


def count_sum(nn, cost, res, big_cost=BIG_COST):
    """greedy_opt.py:21-29 with positional lists: because cost[taxi][trip] IS
    dist[supply[taxi].to][demand[trip].from] for every real cell, the sum over x==1 cells with
    cost < big_cost equals the reference's sum of dist[...] terms. `res` is x (n*n) or row_to_col."""
    res = np.asarray(res)
    if nn == 1:
        # x = [1] and row_to_col = [0] have the same size: with one row the only assignment is column 0
        r2c = np.zeros(1, np.int32)
    elif res.size == nn * nn:
        r2c = np.full(nn, -1, np.int32)
        ii, jj = np.nonzero(res.reshape(nn, nn) == 1)
        r2c[ii] = jj
    else:
        r2c = _ffi.as_i32(res)
    s = ctypes.c_int64(0)
    k = ctypes.c_int32(0)
    if not hasattr(cost, "data_ptr"):
        cost = _ffi.as_i32(cost)
    _ffi.check(_ffi.lib().td_count_sum(nn, _ffi.addr(cost), _ffi.addr(r2c), int(big_cost), ctypes.byref(s),
                                       ctypes.byref(k)))
    return int(s.value)


def filter_out(input, allocated, element=None):
    """element given: greedy_opt.py:32-37 (drop rows whose row[element] is in `allocated`);
    element None: simulate.py:64-69 (drop by position). Host list bookkeeping, O(n)."""
    alloc = set(int(a) for a in allocated)
    if element is None:
        output = [row for i, row in enumerate(input) if i not in alloc]
    else:
        output = [row for row in input if row[element] not in alloc]
    return len(output), output


def combined(distances, demand, cabs, threshold=10, big_cost=BIG_COST):
    """greedy_opt.py:136-160: optimal solve, then LCM(threshold) + optimal on the remainder.
    Returns (nn, res, n2, res2 + lcm) — the four numbers the reference appends to its log."""
    nn, x, cost_table = solve(distances, demand, cabs, big_cost)
    if nn == 0:
        return 0, 0, 0, 0
    res = count_sum(nn, cost_table, x, big_cost)
    lcm, allocated_cabs, allocated_cust = LCM(nn, cost_table, threshold, big_cost)
    # ids equal positions in rand_list (greedy_opt.py:47-50) but filter by the records' own id
    cabs = list(cabs)
    demand = list(demand)
    cab_ids = [cabs[i][0] for i in allocated_cabs if i < len(cabs)]
    cust_ids = [demand[i][0] for i in allocated_cust if i < len(demand)]
    _, rest_demand = filter_out(demand, cust_ids, 0)
    _, rest_cabs = filter_out(cabs, cab_ids, 0)
    n2, x2, cost_table2 = solve(distances, rest_demand, rest_cabs, big_cost)
    res2 = count_sum(n2, cost_table2, x2, big_cost) if n2 else 0
    return nn, res, n2, res2 + lcm
