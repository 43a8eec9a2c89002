"""CPU tests: the oracle (oracle/td_oracle.c) against the reference's known-answer instances,
an independent exact solver (scipy) and plain-numpy restatements of the reference loops."""
import json
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import oracle

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


def numpy_lcm(cost, threshold, big_cost, mask=None, sum_all=False):
    """greedy_opt.py:61-82 restated line by line (small n only)."""
    mask = big_cost if mask is None else mask
    n = cost.shape[0]
    d = np.array(cost).flatten()
    rows, cols, total = [], [], 0
    for _ in range(n):
        elem = int(d.argmin(0))
        if threshold is not None and d[elem] > threshold:
            break
        row = int(elem / n)
        col = elem - row * n
        rows.append(row)
        cols.append(col)
        if sum_all or d[elem] < big_cost:
            total += int(d[elem])
        for j in range(n):
            d[n * row + j] = mask
            d[j * n + col] = mask
    return total, rows, cols


def java_lcm(cost, max_non_lcm, big_cost):
    """Simulator.java:523-549 restated."""
    n = cost.shape[0]
    c = cost.copy()
    pairs, size, lm = [], n, big_cost
    for _ in range(n):
        lm, smin, dmin = big_cost, -1, -1
        for s in range(n):
            for d in range(n):
                if c[s, d] < lm:
                    lm, smin, dmin = int(c[s, d]), s, d
        if lm == big_cost:
            break
        pairs.append((smin, dmin))
        c[:, dmin] = big_cost
        c[smin, :] = big_cost
        size -= 1
        if size == max_non_lcm:
            break
    return pairs, lm


def test_known_answers():
    g = GOLD["pdf_table5"]
    t, r, u, v = oracle.assign(np.array(g["cost"], np.int32))
    assert t == g["total"] == 101
    assert oracle.is_unique(np.array(g["cost"], np.int32), r, u, v) == g["unique"] is False
    g = GOLD["procedure_py"]
    t, r, u, v = oracle.assign(np.array(g["cost"], np.int32))
    assert t == g["total"] == 17
    assert g["cost"] == [[3, 3, 0, 2], [1, 1, 2, 4], [5, 5, 2, 0], [16, 16, 16, 16]]
    # cab0 -> cust2 and cab2 -> cust3 are forced in every optimum
    assert r[0] == 2 and r[2] == 3
    g = GOLD["julia_3x4"]
    assert g["total"] == 1


def test_procedure_cost_by_id():
    g = GOLD["procedure_py"]
    cabs = np.array(g["cabs"])
    dem = np.array(g["demand"])
    n, cost = oracle.cost_build_by_id(cabs[:, 0], cabs[:, 2], dem[:, 0], dem[:, 1])
    assert n == 4 and cost.tolist() == g["cost"]
    S = g["n_stands"]
    dist = np.abs(np.arange(S)[:, None] - np.arange(S)[None, :])
    n, cost2 = oracle.cost_build_by_id(cabs[:, 0], cabs[:, 2], dem[:, 0], dem[:, 1], dist)
    assert cost2.tolist() == g["cost"]


def test_gen_uniform_pinned():
    g = GOLD["gen_uniform_seed1_n8"]
    m = oracle.gen_uniform(8, 1, 10, 40)
    assert m[0].tolist() == g["first_row"] and int(m.sum()) == g["sum"]
    assert m.min() >= 10 and m.max() <= 40
    # row-sharded generation is identical to the full one
    assert np.array_equal(oracle.gen_uniform(8, 1, 10, 40, row0=3, nrows=2), m[3:5])
    big = oracle.gen_uniform(512, 7, 10, 40)
    assert set(np.unique(big)) == set(range(10, 41))


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 5), (2, 0, 3), (7, 0, 3), (50, 1, 40), (100, 1, 40), (200, 10, 41),
                                     (150, 0, 1000000), (400, 10, 41)])
def test_assign_vs_scipy(n, lo, hi):
    rng = np.random.default_rng(n * 31 + hi)
    for _ in range(3):
        c = rng.integers(lo, hi, (n, n)).astype(np.int32)
        t, r, u, v = oracle.assign(c)
        ri, ci = linear_sum_assignment(c)
        assert t == int(c[ri, ci].sum())
        assert sorted(r.tolist()) == list(range(n))
        rc, p, d = oracle.certificate(c, r, u, v)
        assert rc == 0 and p == d == t


def test_assign_negative_and_structured():
    rng = np.random.default_rng(5)
    c = rng.integers(-1000, 1000, (60, 60)).astype(np.int32)
    t, r, u, v = oracle.assign(c)
    ri, ci = linear_sum_assignment(c)
    assert t == int(c[ri, ci].sum())
    # 1-D geometry has a closed form: sorted matching (SURVEY 8c fixture 7)
    for n in (50, 400):
        a = rng.integers(0, 10 * n, n)
        b = rng.integers(0, 10 * n, n)
        c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
        assert oracle.assign(c)[0] == int(np.abs(np.sort(a) - np.sort(b)).sum())


def test_uniqueness():
    c = np.array([[1, 5], [5, 1]], np.int32)
    t, r, u, v = oracle.assign(c)
    assert t == 2 and oracle.is_unique(c, r, u, v)
    c = np.array([[1, 1], [1, 1]], np.int32)
    t, r, u, v = oracle.assign(c)
    assert t == 2 and not oracle.is_unique(c, r, u, v)


def test_cost_build_variants():
    rng = np.random.default_rng(3)
    S = 50
    dist = np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]).astype(np.int32)
    cab_to = rng.integers(0, S, 13)
    dem_from = rng.integers(0, S, 7)
    n, c = oracle.cost_build(cab_to, dem_from, dist, fill=250000, threshold=-1)
    assert n == 13 and c.shape == (13, 13)
    assert np.array_equal(c[:, :7], np.abs(cab_to[:, None] - dem_from[None, :]))
    assert (c[:, 7:] == 250000).all()
    n, c2 = oracle.cost_build(cab_to, dem_from, None, fill=250000, threshold=10)
    ref = np.abs(cab_to[:, None] - dem_from[None, :])
    ref = np.where(ref < 10, ref, 250000)
    assert np.array_equal(c2[:, :7], ref)
    ids = np.arange(13)
    ids[4] = -1
    n, c3 = oracle.cost_build(cab_to, dem_from, dist, fill=250000, threshold=10, cab_id=ids)
    assert (c3[4] == 250000).all() and np.array_equal(np.delete(c3, 4, 0), np.delete(c2, 4, 0))


def test_lcm_matches_numpy_and_java_restatements():
    rng = np.random.default_rng(11)
    for n in (5, 17, 40):
        c = rng.integers(0, 30, (n, n)).astype(np.int32)
        tot, rows, cols, _ = oracle.lcm(c, mask=250000, threshold=10, sum_below=250000)
        t2, r2, c2 = numpy_lcm(c, 10, 250000)
        assert (tot, rows.tolist(), cols.tolist()) == (t2, r2, c2)
        # heuristic.py:24-33
        c = rng.integers(1, 40, (n, n)).astype(np.int32)
        tot, rows, cols, _ = oracle.lcm(c, mask=100, threshold=-1)
        t2, r2, c2 = numpy_lcm(c, None, 250000, mask=100, sum_all=True)
        assert (tot, rows.tolist(), cols.tolist()) == (t2, r2, c2)
        # Simulator.java
        c = rng.integers(0, 12, (n, n)).astype(np.int32)
        c[c >= 10] = 250000
        stop = n // 2
        _, rows, cols, lm = oracle.lcm(c, mask=250000, stop_value_on=1, stop_value=250000, stop_size=stop,
                                       sum_below=250000, java_scan=1)
        pairs, lm2 = java_lcm(c, stop, 250000)
        assert list(zip(rows.tolist(), cols.tolist())) == pairs and lm == lm2


def test_statistical_kat_heuristic_gap():
    """taxi_dispatching.pdf p.4: LCM is ~78 % worse than the optimum on 100x100 costs 1..39
    (heuristic.py:5-6,21,39). 30 cases, tolerance +-10 points."""
    rng = np.random.default_rng(2020)
    gaps = []
    for _ in range(30):
        c = rng.integers(1, 40, (100, 100)).astype(np.int32)
        lcm_total = oracle.lcm(c, mask=100, threshold=-1)[0]
        opt = oracle.assign(c)[0]
        assert opt <= lcm_total  # heuristic.py:40 "!!!" detector
        gaps.append(100.0 * (lcm_total - opt) / opt)
    assert 68 < float(np.mean(gaps)) < 88


def test_statistical_kat_combined_method():
    """PDF p.6: LCM(threshold 10) + optimum on the rest is ~2.4 % worse; model 400 -> ~146
    (greedy_opt.py:7-9,131-160). 8 cases, loose tolerance."""
    rng = np.random.default_rng(7)
    S, n0 = 4000, 400
    gaps, sizes = [], []
    for _ in range(8):
        def rand_list():
            frm = rng.integers(0, S, n0)
            to = rng.integers(0, S, n0)
            keep = frm != to
            return frm[keep], to[keep]
        d_frm, _ = rand_list()
        _, c_to = rand_list()
        n, cost = oracle.cost_build(c_to, d_frm, None, fill=250000)
        t, r, _, _ = oracle.assign(cost)
        opt = oracle.count_sum(cost, r)[0]
        lcm, rows, cols, _ = oracle.lcm(cost, mask=250000, threshold=10, sum_below=250000)
        rest_c = np.delete(c_to, rows[rows < c_to.size])
        rest_d = np.delete(d_frm, cols[cols < d_frm.size])
        n2, cost2 = oracle.cost_build(rest_c, rest_d, None, fill=250000)
        t2, r2, _, _ = oracle.assign(cost2)
        opt2 = oracle.count_sum(cost2, r2)[0]
        gaps.append(100.0 * (opt2 + lcm - opt) / opt)
        sizes.append(n2)
    assert 0.5 < float(np.mean(gaps)) < 5.0
    assert 125 < float(np.mean(sizes)) < 170


def test_oracle_under_address_sanitizer():
    """`make -C oracle asan` + a run of the oracle's entry points under ASan/UBSan in a child process
    (the GPU side cannot be sanitised on this pool, the CPU checker can)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    code = r'''
import numpy as np, os, sys
sys.path.insert(0, %r)
from oracle import oracle
oracle._LIB = os.path.join(%r, "oracle", "_build", "liboracle_asan.so")
oracle._lib = None
import ctypes
oracle.build = lambda force=False: oracle._LIB
rng = np.random.default_rng(0)
for n in (1, 2, 7, 64, 130):
    c = rng.integers(0, 50, (n, n)).astype(np.int32)
    t, r2c, u, v = oracle.assign(c)
    assert sorted(r2c.tolist()) == list(range(n)) and t == int(c[np.arange(n), r2c].sum())
    oracle.is_unique(c, r2c, u, v)
    oracle.lcm(c, mask=250000, threshold=10, sum_below=250000)
    oracle.lcm(c, mask=250000, stop_value_on=1, stop_value=250000, stop_size=max(0, n - 3), sum_below=250000, java_scan=1)
n, cost = oracle.cost_build(rng.integers(0, 50, 33), rng.integers(0, 50, 21), None, 250000, 10)
oracle.cost_build(rng.integers(0, 9, 5), rng.integers(0, 9, 8), rng.integers(0, 9, (9, 9)).astype(np.int32), 250000, -1)
oracle.gen_uniform(37, 3, 10, 40)
print("asan ok")
''' % (root, root)
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_pool_n_oracle_matches_reference_outputs():
    """f-4: the oracle's restatement of pool_n.c against the outputs of the reference binary itself
    (tests/golden/pool_n, made by tests/golden/make_pool_fixtures.py from oracle/_ref/pool_n)."""
    import pool_fixtures as pf
    assert len(pf.cases()) >= 10
    for name, k in pf.cases():
        d, exp = pf.load(name, k)
        n = len(d)
        for child in range(8):
            a, b = pf.child_slice(n, child)
            out, nh = oracle.pool_n(k, d[:, 1], d[:, 2], d[:, 3], d[:, 4], None, a, b)
            assert out.tolist() == exp[child], (name, k, child)
            assert nh >= len(exp[child])
