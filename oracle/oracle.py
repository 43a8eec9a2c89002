"""ctypes wrapper of the CPU oracle (oracle/td_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
taxidispatcher_amd/.  Builds oracle/_build/liboracle.so with gcc on first use.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
INT32_MAX = 2**31 - 1


def build(force=False):
    src = os.path.join(_HERE, "td_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fPIC", "-fvisibility=hidden", "-shared",
                               "-o", _LIB, src])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.oracle_assign.restype = ctypes.c_int64
        L.oracle_dual_bound_scaled.restype = ctypes.c_int64
        L.oracle_count_sum.restype = ctypes.c_int64
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def cost_build(cab_to, dem_from, dist=None, fill=250000, threshold=-1, cab_id=None, dem_id=None):
    cab_to, dem_from = _i32(cab_to), _i32(dem_from)
    n = max(cab_to.size, dem_from.size)
    cost = np.empty((n, n), np.int32)
    d = None if dist is None else _i32(dist)
    S = 0 if d is None else d.shape[0]
    ci = None if cab_id is None else _i32(cab_id)
    di = None if dem_id is None else _i32(dem_id)
    lib().oracle_cost_build(_p(cab_to), _p(ci), cab_to.size, _p(dem_from), _p(di), dem_from.size, _p(d), S,
                            ctypes.c_int32(fill), ctypes.c_int32(threshold), _p(cost))
    return n, cost


def cost_build_by_id(cab_id, cab_to, dem_id, dem_from, dist=None):
    cab_id, cab_to, dem_id, dem_from = map(_i32, (cab_id, cab_to, dem_id, dem_from))
    n = max(cab_to.size, dem_from.size)
    cost = np.empty((n, n), np.int32)
    d = None if dist is None else _i32(dist)
    S = 0 if d is None else d.shape[0]
    lib().oracle_cost_build_by_id(_p(cab_id), _p(cab_to), cab_to.size, _p(dem_id), _p(dem_from), dem_from.size,
                                  _p(d), S, _p(cost))
    return n, cost


def lcm(cost, mask=250000, max_iter=None, threshold=-1, stop_value_on=0, stop_value=0, stop_size=-1,
        sum_below=2**62, java_scan=0):
    cost = _i32(cost)
    n = cost.shape[0]
    rows = np.empty(max(n, 1), np.int32)
    cols = np.empty(max(n, 1), np.int32)
    tot = ctypes.c_int64(0)
    lm = ctypes.c_int32(0)
    k = lib().oracle_lcm(n, _p(cost), ctypes.c_int32(mask), n if max_iter is None else max_iter,
                         ctypes.c_int32(threshold), stop_value_on, ctypes.c_int32(stop_value), stop_size,
                         ctypes.c_int64(sum_below), java_scan, _p(rows), _p(cols), ctypes.byref(tot),
                         ctypes.byref(lm))
    return int(tot.value), rows[:k].copy(), cols[:k].copy(), int(lm.value)


def assign(cost):
    """exact optimum -> (total, row_to_col, u, v) with sum(u)+sum(v) == total"""
    cost = _i32(cost)
    n = cost.shape[0]
    r = np.zeros(n, np.int32)
    u = np.zeros(n, np.int64)
    v = np.zeros(n, np.int64)
    t = lib().oracle_assign(n, _p(cost), _p(r), _p(u), _p(v))
    return int(t), r, u, v


def certificate(cost, row_to_col, u, v):
    cost = _i32(cost)
    p = ctypes.c_int64(0)
    d = ctypes.c_int64(0)
    rc = lib().oracle_certificate(cost.shape[0], _p(cost), _p(_i32(row_to_col)),
                                  _p(np.ascontiguousarray(u, dtype=np.int64)),
                                  _p(np.ascontiguousarray(v, dtype=np.int64)), ctypes.byref(p), ctypes.byref(d))
    return rc, int(p.value), int(d.value)


def is_unique(cost, row_to_col, u, v):
    cost = _i32(cost)
    return bool(lib().oracle_is_unique(cost.shape[0], _p(cost), _p(_i32(row_to_col)),
                                       _p(np.ascontiguousarray(u, dtype=np.int64)),
                                       _p(np.ascontiguousarray(v, dtype=np.int64))))


def count_sum(cost, row_to_col, big_cost=250000):
    cost = _i32(cost)
    k = ctypes.c_int32(0)
    s = lib().oracle_count_sum(cost.shape[0], _p(cost), _p(_i32(row_to_col)), ctypes.c_int64(big_cost),
                               ctypes.byref(k))
    return int(s), int(k.value)


def gen_uniform(n, seed, lo=10, hi=40, row0=0, nrows=None):
    nrows = n if nrows is None else nrows
    cost = np.empty((nrows, n), np.int32)
    lib().oracle_gen_uniform(n, ctypes.c_uint64(seed), ctypes.c_int32(lo), ctypes.c_int32(hi), row0, nrows,
                             _p(cost))
    return cost


def solve_x(n, cost):
    """solver.py:11-27 restated: x vector of length n*n, x[n*cab+cust] == 1 on the optimum."""
    if n == 0:
        return 0, []
    _, r, _, _ = assign(np.asarray(cost).reshape(n, n))
    x = np.zeros(n * n, np.uint8)
    x[np.arange(n) * n + r] = 1
    return x


def pool_n(k, frm, to, wait, loss, dist=None, first0=0, first1=None, cap=200000):
    """f-4 (pool_n.c): kept plans of one first-pick-up slice as an int32 array [m, 2k+1]
    (pick-ups, drop-offs, cost) in the reference's output order, and the number of happy plans."""
    frm, to, wait, loss = map(_i32, (frm, to, wait, loss))
    n = int(frm.size)
    first1 = n if first1 is None else first1
    out = np.zeros((cap, 2 * k + 1), np.int32)
    nh = ctypes.c_long(0)
    d = None if dist is None else _i32(dist)
    L = lib()
    L.oracle_pool_n.restype = ctypes.c_long
    m = L.oracle_pool_n(k, n, _p(frm), _p(to), _p(wait), _p(loss), _p(d), 0 if d is None else d.shape[0],
                        int(first0), int(first1), ctypes.c_long(cap), _p(out), ctypes.byref(nh))
    if m < 0:
        raise OverflowError("more than %d happy plans" % cap)
    return out[:m].copy(), int(nh.value)
