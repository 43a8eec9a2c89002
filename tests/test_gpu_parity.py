"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI
(taxidispatcher_amd/_ffi.py -> libtaxidispatcher_amd.so) and is checked against the oracle.
Integer work: bit-exact."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))
BIG = 250000


@pytest.fixture(autouse=True)
def _general_solver_only(general_solver):
    """This module pins the GENERAL solver: |a-b| instances would otherwise be answered by the line-metric
    attempt (sorted matching + certificate, tests/test_gpu_line.py) before the solver under test runs."""
    yield


def make_instance(kind, n, rng):
    if kind == "g1":      # perf.jl:5
        return rng.integers(10, 41, (n, n)).astype(np.int32)
    if kind == "g4":      # heuristic.py:21
        return rng.integers(1, 40, (n, n)).astype(np.int32)
    if kind == "wide":
        return rng.integers(0, 1000000, (n, n)).astype(np.int32)
    if kind == "neg":
        return rng.integers(-5000, 5000, (n, n)).astype(np.int32)
    if kind == "const":
        return np.full((n, n), 7, np.int32)
    S = 50 if kind == "g3" else 10 * n
    a = rng.integers(0, S, n)
    b = rng.integers(0, S, n)
    c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    if kind == "g3":      # Simulator.java:493-520 shape: threshold 10, dummy columns
        nd = max(1, int(n * 0.363))
        c[c >= 10] = BIG
        c[:, nd:] = BIG
    return c


def check_assignment(td, c):
    n = c.shape[0]
    r2c, total, dual = td.assign(c, want_dual=True)
    t_ref, r_ref, u, v = oracle.assign(c)
    assert total == t_ref, "total differs from the optimum"
    assert dual == total, "device LP-duality certificate does not close"
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total
    if n <= 600 and oracle.is_unique(c, r_ref, u, v):
        assert np.array_equal(r2c, r_ref), "unique optimum but per-cab assignment differs"
    return r2c, total


def test_julia_3x4_rectangular_on_gpu(td):
    """julia.jl:3-15: 3 cabs x 4 requests, rows == 1, columns <= 1 -> optimum 1.  The reference's own way to a square
    model is a dummy cab row (solver.py pads with one value): a constant row costs the same wherever it lands, so
    total - that constant is the rectangular optimum.  Constant rows are deferred by td_assign (k_place_const)."""
    c34 = np.array(GOLD["julia_3x4"]["cost"], np.int32)
    for pad in (0, 16, 250000):
        c = np.vstack([c34, np.full((1, 4), pad, np.int32)])
        r2c, total, dual = td.assign(c, want_dual=True)
        assert total - pad == GOLD["julia_3x4"]["total"] == 1 and dual == total
        assert sorted(np.asarray(r2c).tolist()) == [0, 1, 2, 3]
        assert int(c34[np.arange(3), np.asarray(r2c)[:3]].sum()) == 1
        assert oracle.assign(c)[0] == total


def test_known_answers(td):
    for key in ("pdf_table5", "procedure_py"):
        c = np.array(GOLD[key]["cost"], np.int32)
        r2c, total = check_assignment(td, c)
        assert total == GOLD[key]["total"]
    c = np.array(GOLD["procedure_py"]["cost"], np.int32)
    r2c, _ = td.assign(c)
    assert r2c[0] == 2 and r2c[2] == 3  # forced in every optimum


def test_procedure_py_api(td):
    """procedure.py:32-57 end to end: same inputs, same printed pairs up to ties."""
    g = GOLD["procedure_py"]
    S = g["n_stands"]
    dist = np.zeros((S, S))
    for i in range(S):
        for j in range(i, S):
            dist[j][i] = j - i
            dist[i][j] = dist[j][i]
    demand = [tuple(r) for r in g["demand"]]
    cabs = [tuple(r) for r in g["cabs"]]
    x = td.procedure_solve(dist, demand, cabs)
    n_cust = 4
    assert len(x) == 16 and int(x.sum()) == 4
    taken = [(dem, trip) for dem in range(3) for trip in range(n_cust) if x[n_cust * dem + trip] == 1]
    assert len(taken) == 3 and (0, 2) in taken and (2, 3) in taken
    n, cost = td.calculate_cost_by_id(dist, demand, cabs)
    assert cost.tolist() == g["cost"]
    real = sum(cost[c][d] for c, d in taken)
    assert real == g["real_total"]


@pytest.mark.parametrize("n", [1, 2, 3, 5, 15, 16, 17, 31, 63, 64, 65, 100, 127, 128, 129, 257, 400, 600, 1000])
def test_assign_sizes(td, n):
    rng = np.random.default_rng(1000 + n)
    for kind in ("g1", "g4", "g2", "g3", "wide", "neg", "const"):
        check_assignment(td, make_instance(kind, n, rng))


@pytest.mark.parametrize("kind,n", [("g1", 1300), ("g3", 1300), ("g2", 1300), ("g1", 2048), ("wide", 2048),
                                    ("g1", 4096), ("g3", 2500)])
def test_assign_larger(td, kind, n):
    rng = np.random.default_rng(n)
    check_assignment(td, make_instance(kind, n, rng))


def test_assign_unique_optimum_per_cab(td):
    """Strictly dominant diagonal => unique optimum => per-cab parity is required."""
    rng = np.random.default_rng(9)
    for n in (8, 100, 513):
        perm = rng.permutation(n)
        c = rng.integers(50, 90, (n, n)).astype(np.int32)
        c[np.arange(n), perm] = rng.integers(0, 5, n)
        r2c, total = check_assignment(td, c)
        t_ref, r_ref, u, v = oracle.assign(c)
        assert oracle.is_unique(c, r_ref, u, v)
        assert np.array_equal(r2c, perm) and np.array_equal(r_ref, perm)


def test_assign_extreme_values(td):
    rng = np.random.default_rng(4)
    n = 40
    c = rng.integers(0, 2**31 - 1, (n, n)).astype(np.int32)       # needs the u32 path
    check_assignment(td, c)
    c = rng.integers(-2**31, 2**31 - 1, (n, n)).astype(np.int32)
    check_assignment(td, c)
    c = rng.integers(0, 300, (n, n)).astype(np.int32)              # u16 path
    check_assignment(td, c)
    c = rng.integers(0, 70000, (n, n)).astype(np.int32)            # u32 path, moderate
    check_assignment(td, c)
    # row offsets far apart but narrow per-row range -> still the u8 path
    c = (rng.integers(0, 200, (n, n)) + rng.integers(0, 10**9, (n, 1))).astype(np.int32)
    check_assignment(td, c)
    assert td.last_stats()["bytes_per_cell"] == 1


def test_empty_and_device_pointers(td):
    import torch
    assert td.solve_cost(0, []) == (0, [])
    assert td.calculate_cost(None, [], []) == (0, 0)
    n = 300
    c = oracle.gen_uniform(n, 3)
    ct = torch.from_numpy(c).cuda()
    r2c_t = torch.empty(n, dtype=torch.int32, device="cuda")
    tot = ctypes.c_int64(0)
    from taxidispatcher_amd import _ffi
    _ffi.check(_ffi.lib().td_assign(n, ct.data_ptr(), r2c_t.data_ptr(), ctypes.byref(tot), None))
    assert tot.value == oracle.assign(c)[0]
    r = r2c_t.cpu().numpy()
    assert sorted(r.tolist()) == list(range(n)) and int(c[np.arange(n), r].sum()) == tot.value


def test_gen_uniform_bit_exact(td):
    from taxidispatcher_amd import _ffi
    for n, row0, nrows in [(8, 0, 8), (1000, 0, 1000), (1001, 5, 77), (64, 63, 1)]:
        out = np.empty((nrows, n), np.int32)
        _ffi.check(_ffi.lib().td_gen_uniform(n, 1, 10, 40, row0, nrows, out.ctypes.data))
        assert np.array_equal(out, oracle.gen_uniform(n, 1, 10, 40, row0, nrows))
    g = GOLD["gen_uniform_seed1_n8"]
    out = np.empty((8, 8), np.int32)
    _ffi.check(_ffi.lib().td_gen_uniform(8, 1, 10, 40, 0, 8, out.ctypes.data))
    assert out[0].tolist() == g["first_row"] and int(out.sum()) == g["sum"]


@pytest.mark.parametrize("n_s,n_d", [(1, 1), (3, 4), (13, 7), (7, 13), (64, 64), (600, 218), (257, 1001), (1300, 1300)])
def test_cost_build_variants(td, n_s, n_d):
    rng = np.random.default_rng(n_s * 7 + n_d)
    for S in (50, 300):
        cab_to = rng.integers(0, S, n_s)
        dem_from = rng.integers(0, S, n_d)
        dist = rng.integers(0, 60, (S, S)).astype(np.int32)   # a general (asymmetric) table
        for d in (None, dist):
            for thr in (-1, 10):
                n, c = td.cost_build(cab_to, dem_from, d, fill=BIG, threshold=thr)
                n_o, c_o = oracle.cost_build(cab_to, dem_from, d, BIG, thr)
                assert n == n_o and np.array_equal(c, c_o)
        cab_id = np.arange(n_s)
        dem_id = np.arange(n_d)
        cab_id[rng.integers(0, n_s)] = -1
        dem_id[rng.integers(0, n_d)] = -1
        n, c = td.cost_build(cab_to, dem_from, dist, fill=BIG, threshold=10, cab_id=cab_id, dem_id=dem_id)
        n_o, c_o = oracle.cost_build(cab_to, dem_from, dist, BIG, 10, cab_id, dem_id)
        assert np.array_equal(c, c_o)


def test_reference_shaped_calculate_cost_and_solve(td):
    """greedy_opt.py:86-118 / simulate.py:17-53 call shapes."""
    rng = np.random.default_rng(12)
    S = 4000
    demand = [(i, int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(57)]
    cabs = [(i, int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(60)]
    n, cost = td.calculate_cost(None, demand, cabs)
    assert n == 60 and cost[3][5] == abs(cabs[3][2] - demand[5][1]) and cost[0][59] == BIG
    nn, x, cost_table = td.solve(None, demand, cabs)
    assert nn == 60 and len(x) == 3600 and int(np.asarray(x).sum()) == 60
    res = td.count_sum(nn, cost_table, x)
    t_ref, r_ref, _, _ = oracle.assign(cost_table)
    assert res == oracle.count_sum(cost_table, r_ref)[0]
    # simulate.py variant with DROP_TIME on a 50-stand world
    S = 50
    dist = np.abs(np.arange(S)[:, None] - np.arange(S)[None, :])
    demand = [(i, int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(30)]
    cabs = [(i, int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(45)]
    nn, x, cost_table = td.solve(dist, demand, cabs, drop_time=10)
    _, c_o = oracle.cost_build([c[2] for c in cabs], [d[1] for d in demand], dist, BIG, 10)
    assert np.array_equal(cost_table, c_o)
    r2c = np.asarray(x).reshape(nn, nn).argmax(1)
    assert int(c_o[np.arange(nn), r2c].sum()) == oracle.assign(c_o)[0]
    assert td.solve(dist, [], [], drop_time=10) == (0, [], 0)


def test_cooperative_finisher_large_nontied(td):
    """n = 8192 with 32-bit rows: the rows the speculative batches leave are finished by k_sapx (one
    search spread over 8 workgroups, grid barrier per step). |a-b| costs have a closed-form optimum
    (sorted matching); the wide uniform instance is certified by the LP dual."""
    import torch
    n = 8192
    rng = np.random.default_rng(4)
    a = rng.integers(0, 10 * n, n).astype(np.int32)
    b = rng.integers(0, 10 * n, n).astype(np.int32)
    ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
    td.cost_build(a, b, None, fill=BIG, threshold=-1, out=ct)
    r2c, total, dual = td.assign(ct, n, want_dual=True)
    expected = int(np.abs(np.sort(a.astype(np.int64)) - np.sort(b.astype(np.int64))).sum())
    assert total == expected == dual
    r2c = np.asarray(r2c)
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(np.abs(a.astype(np.int64) - b.astype(np.int64)[r2c]).sum()) == expected
    assert td.last_stats()["sap_free_rows"] > 0   # the serial finisher had work to do
    c = rng.integers(0, 10**6, (n, n)).astype(np.int32)
    r2c, total, dual = td.assign(c, n, want_dual=True)
    r2c = np.asarray(r2c)
    assert total == dual and sorted(r2c.tolist()) == list(range(n))
    assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total


def test_rectangular_models_constant_rows_and_columns(td):
    """Padded rectangular models (solver.py pads to a square with big_cost; Simulator.java:244-252):
    constant rows are deferred, many constant columns switch td_assign to the transposed
    formulation - total, dual certificate and permutation must match the oracle either way."""
    rng = np.random.default_rng(99)
    seen_t = 0
    for trial in range(60):
        n = int(rng.choice([64, 65, 100, 257, 600, 1300]))
        real_r = int(rng.integers(1, n + 1))
        real_c = int(rng.integers(1, n + 1))
        if trial % 3 == 0:
            real_r, real_c = n, int(rng.integers(1, max(2, n // 2)))     # dummy columns dominate
        c = np.full((n, n), BIG, np.int32)
        blk = rng.integers(0, int(rng.choice([10, 50, 100000])), (real_r, real_c)).astype(np.int32)
        if trial % 2 == 0:   # DROP_TIME style threshold inside the real block
            blk[rng.random(blk.shape) < 0.7] = BIG
        c[:real_r, :real_c] = blk
        if trial % 5 == 0:   # constant rows / columns need not share one value
            c[real_r:, :] = rng.integers(0, 1000, (n - real_r, 1)).astype(np.int32)
        if trial % 7 == 0:
            c = c[rng.permutation(n)][:, rng.permutation(n)]
        c = np.ascontiguousarray(c)
        r2c, total, dual = td.assign(c, n, want_dual=True)
        t_o = oracle.assign(c)[0]
        assert total == t_o == dual, (trial, n, real_r, real_c, total, t_o, dual)
        r2c = np.asarray(r2c)
        assert sorted(r2c.tolist()) == list(range(n))
        assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == t_o
        seen_t += td.last_stats()["transposed"]
    assert seen_t >= 10   # the transposed path was exercised


@pytest.mark.parametrize("n,pad,top", [(700, 1000, 3 * 10**8), (1500, 300, 2**22 + 5000), (3000, 1000, 10**6), (2500, 255, 5000)])
def test_padded_model_whose_real_cells_exceed_the_pad_value(td, n, pad, top):
    """the speculative fused transpose + compress pass takes the value of the (constant) last column for the
    model's range and goes on with 32-bit prices; a model padded with a SMALL value whose real cells are larger (up to
    beyond 2^28, the 32-bit kernels' BIG) must be caught on the device (k_tr_finish against the assumed range) and
    redone on the general path — same optimum as the oracle, closing certificate.  The line attempt is ON here (the
    default flow: its probe is what estimates the pad value)."""
    was = td.set_line_metric(True)
    try:
        rng = np.random.default_rng(n)
        real_c = n // 3
        c = np.full((n, n), pad, np.int32)
        c[:, :real_c] = rng.integers(0, top, (n, real_c)).astype(np.int32)
        r2c, total, dual = td.assign(c, n, want_dual=True)
        t_o = oracle.assign(c)[0]
        assert total == t_o == dual, (total, t_o, dual, td.last_stats())
        r2c = np.asarray(r2c)
        assert sorted(r2c.tolist()) == list(range(n))
        assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == t_o
    finally:
        td.set_line_metric(was)


def test_lcm_randomized_differential(td):
    """td_lcm against the oracle over random shapes of every stop rule: threshold, stop value,
    stop size, pair cap, masked cells, exhausted lists; instances with few value levels take the
    level-list kernels, wide ones the row-scan loop - all five outputs must agree."""
    from taxidispatcher_amd import dispatch
    rng = np.random.default_rng(20260)
    for trial in range(160):
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 100, 129, 257, 400, 1000]))
        kind = trial % 4
        if kind == 0:      # distances 0..R with masked cells, thresholded (greedy_opt.py / simulate.py)
            R = int(rng.choice([3, 15, 40, 200]))
            c = rng.integers(0, R + 1, (n, n)).astype(np.int32)
            c[rng.random((n, n)) < rng.choice([0.0, 0.3, 0.9])] = BIG
            args = dict(mask=BIG, threshold=int(rng.choice([0, 2, 10, 20])), sum_below=BIG)
        elif kind == 1:    # Simulator.java: stop value = big cost, stop at a model size
            c = rng.integers(0, 10, (n, n)).astype(np.int32)
            c[rng.random((n, n)) < rng.choice([0.2, 0.8, 0.97])] = BIG
            args = dict(mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=int(rng.choice([-1, 0, n // 3, n - 1, n, n + 5])),
                        sum_below=BIG)
        elif kind == 2:    # heuristic.py: every cell a candidate, small mask value
            c = rng.integers(1, 40, (n, n)).astype(np.int32)
            args = dict(mask=100, threshold=-1)
        else:              # wide values, negative values, pair cap
            lo = int(rng.choice([-50, 0]))
            c = rng.integers(lo, lo + int(rng.choice([30, 60, 100000])), (n, n)).astype(np.int32)
            args = dict(mask=BIG, threshold=int(rng.choice([-1, 25])), sum_below=int(rng.choice([BIG, 20])),
                        max_iter=int(rng.integers(0, n + 1)))
        t_o, r_o, c_o, lm_o = oracle.lcm(c, **args)
        got = dispatch._lcm(n, c, args["mask"], args.get("threshold", -1), args.get("stop_value_on", 0),
                            args.get("stop_value", 0), args.get("stop_size", -1), args.get("sum_below", 2**62),
                            max_pairs=args.get("max_iter"))
        assert (got[0], got[3]) == (t_o, lm_o), (trial, n, args, got[0], t_o, got[3], lm_o)
        assert np.array_equal(got[1], r_o) and np.array_equal(got[2], c_o), (trial, n, args)


def _sorted_cell_greedy(c, threshold):
    """Test-local restatement of the thresholded lowest-cost method for SPARSE candidates:
    cells at or below the threshold in (value, row, col) order, taken when row and column are
    free (what repeated np.argmin + masking does, greedy_opt.py:61-82). The oracle's O(n^2) scan
    per pick is too slow at n = 20000; this one is checked against it at small n below."""
    n = c.shape[0]
    r, k = np.nonzero(c <= threshold)
    v = c[r, k]
    order = np.lexsort((k, r, v))
    rt, ct = np.zeros(n, bool), np.zeros(n, bool)
    rows, cols, tot = [], [], 0
    for i in order:
        a, b = r[i], k[i]
        if not rt[a] and not ct[b]:
            rt[a] = ct[b] = True
            rows.append(a)
            cols.append(b)
            tot += int(v[i])
    return tot, np.array(rows, np.int32), np.array(cols, np.int32)


def test_lcm_large_hashed_tables(td):
    """n above 16384: the row / column tables of the list greedy are shared by several rows."""
    from taxidispatcher_amd import dispatch
    for n, per_row in ((300, 6), (20000, 40)):
        rng = np.random.default_rng(5)
        c = np.full((n, n), BIG, np.int32)
        k = per_row * n
        c[rng.integers(0, n, k), rng.integers(0, n, k)] = rng.integers(0, 12, k).astype(np.int32)
        t_o, r_o, c_o = _sorted_cell_greedy(c, 10)
        if n <= 300:
            t_x, r_x, c_x, _ = oracle.lcm(c, mask=BIG, threshold=10, sum_below=BIG)
            assert t_x == t_o and np.array_equal(r_x, r_o) and np.array_equal(c_x, c_o)
        got = dispatch._lcm(n, c, BIG, 10, 0, 0, -1, BIG)
        assert got[0] == t_o
        assert np.array_equal(got[1], r_o) and np.array_equal(got[2], c_o)


@pytest.mark.parametrize("n", [1, 5, 17, 64, 100, 400, 1300])
def test_lcm_variants(td, n):
    rng = np.random.default_rng(77 + n)
    # greedy_opt.py:61-82, threshold 10
    c = rng.integers(0, 30, (n, n)).astype(np.int32)
    tot, rows, cols = td.LCM(n, c, threshold=10)
    t_o, r_o, c_o, _ = oracle.lcm(c, mask=BIG, threshold=10, sum_below=BIG)
    assert (tot, rows, cols) == (t_o, r_o.tolist(), c_o.tolist())
    # simulate.py:76-98, threshold 20, pair list
    tot, rows, cols, pairs = td.LCM(n, c, threshold=20, with_pairs=True)
    t_o, r_o, c_o, _ = oracle.lcm(c, mask=BIG, threshold=20, sum_below=BIG)
    assert (tot, pairs) == (t_o, list(zip(r_o.tolist(), c_o.tolist())))
    # heuristic.py:24-33
    c = rng.integers(1, 40, (n, n)).astype(np.int32)
    tot, rows, cols = td.LCM_heuristic(n, c)
    t_o, r_o, c_o, _ = oracle.lcm(c, mask=100, threshold=-1)
    assert (tot, rows, cols) == (t_o, r_o.tolist(), c_o.tolist())
    # Simulator.java:523-549 on a thresholded tick-like matrix
    a = rng.integers(0, 50, n)
    b = rng.integers(0, 50, n)
    c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    c[c >= 10] = BIG
    c[:, max(1, int(0.7 * n)):] = BIG
    stop = max(0, n - max(1, n // 3))
    pairs, lm = td.LCM_simulator(c, max_non_lcm=stop)
    _, r_o, c_o, lm_o = oracle.lcm(c, mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=stop, sum_below=BIG,
                                   java_scan=1)
    assert pairs == list(zip(r_o.tolist(), c_o.tolist())) and lm == lm_o
    # exhausted: every real cell consumed -> LCM_min_val == big_cost (Simulator.java:188)
    pairs, lm = td.LCM_simulator(c, max_non_lcm=-1)
    _, r_o, c_o, lm_o = oracle.lcm(c, mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=-1, sum_below=BIG,
                                   java_scan=1)
    assert pairs == list(zip(r_o.tolist(), c_o.tolist())) and lm == lm_o


def test_combined_pipeline(td):
    """greedy_opt.py:136-160 (a-8): n, opt, n2, opt2+lcm."""
    rng = np.random.default_rng(5)
    S, n0 = 4000, 400

    def rand_list():
        out, count = [], 0
        for _ in range(n0):
            frm, to = int(rng.integers(0, S)), int(rng.integers(0, S))
            if frm != to:
                out.append((count, frm, to))
                count += 1
        return out
    demand, cabs = rand_list(), rand_list()
    nn, res, n2, res2 = td.combined(None, demand, cabs, threshold=10)
    n_o, cost = oracle.cost_build([c[2] for c in cabs], [d[1] for d in demand], None, BIG)
    t, r, _, _ = oracle.assign(cost)
    opt = oracle.count_sum(cost, r)[0]
    lcm, rows, cols, _ = oracle.lcm(cost, mask=BIG, threshold=10, sum_below=BIG)
    rest_c = np.delete(np.array([c[2] for c in cabs]), rows)
    rest_d = np.delete(np.array([d[1] for d in demand]), cols)
    n2_o, cost2 = oracle.cost_build(rest_c, rest_d, None, BIG)
    t2, r2, _, _ = oracle.assign(cost2)
    assert (nn, res, n2, res2) == (n_o, opt, n2_o, oracle.count_sum(cost2, r2)[0] + lcm)
    assert res <= res2


def test_solver_cli_roundtrip(td, tmp_path):
    """cost.txt -> python -m taxidispatcher_amd.solver -> solv_out.txt (Simulator.java:195-207)."""
    from taxidispatcher_amd import solver
    rng = np.random.default_rng(2)
    a = rng.integers(0, 50, 40)
    b = rng.integers(0, 50, 25)
    n, cost = oracle.cost_build(a, b, None, BIG, 10)
    solver.write_cost(str(tmp_path / "cost.txt"), cost)
    assert solver.main([str(tmp_path / "cost.txt"), str(tmp_path / "solv_out.txt")]) == 0
    x = solver.read_solution(str(tmp_path / "solv_out.txt"), n)
    xm = x.reshape(n, n)
    assert (xm.sum(0) == 1).all() and (xm.sum(1) == 1).all()
    assert int((xm * cost).sum()) == oracle.assign(cost)[0]
    # OPT count is tie-invariant (Simulator.java:378-383)
    t, r, _, _ = oracle.assign(cost)
    assert int(((xm == 1) & (cost < BIG)).sum()) == oracle.count_sum(cost, r)[1]


def test_large_properties(td):
    """BASELINE sizes through size-independent properties: known optimum 10*N for the perf.jl
    distribution (row-minimum bound), closed-form sorted matching for 1-D geometry, and the
    device certificate."""
    import torch
    from taxidispatcher_amd import _ffi
    n = 16384
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(_ffi.lib().td_gen_uniform(n, 1, 10, 40, 0, n, cost.data_ptr()))
    r2c, total, dual = td.assign(cost, n, want_dual=True)
    assert total == 10 * n == dual
    assert sorted(r2c.tolist()) == list(range(n))
    picked = cost[torch.arange(n, device="cuda"), torch.from_numpy(r2c.astype(np.int64)).cuda()]
    assert int(picked.sum().item()) == total and int(picked.max().item()) == 10
    del cost
    rng = np.random.default_rng(3)
    n = 4096
    a = rng.integers(0, 10 * n, n)
    b = rng.integers(0, 10 * n, n)
    ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
    td.cost_build(a, b, None, fill=BIG, threshold=-1, out=ct)
    r2c, total, dual = td.assign(ct, n, want_dual=True)
    assert total == int(np.abs(np.sort(a) - np.sort(b)).sum()) == dual


def test_randomized_differential(td):
    """300 random small instances: random size, value range, sign, tie density, big_cost
    sentinels and unbalanced padding — total and certificate against the oracle."""
    rng = np.random.default_rng(20201)
    for it in range(300):
        n = int(rng.integers(1, 90))
        mode = it % 6
        if mode == 0:
            c = rng.integers(0, int(rng.integers(1, 6)), (n, n))                   # dense ties
        elif mode == 1:
            c = rng.integers(-10**int(rng.integers(1, 9)), 10**int(rng.integers(1, 9)), (n, n))
        elif mode == 2:
            c = rng.integers(0, 10, (n, n))
            c[rng.random((n, n)) < 0.6] = BIG                                     # thresholded cells
        elif mode == 3:
            ns = int(rng.integers(1, n + 1))
            c = np.full((n, n), BIG)
            c[:ns, :] = rng.integers(0, 50, (ns, n))                              # dummy rows
        elif mode == 4:
            a, b = rng.integers(0, 50, n), rng.integers(0, 50, n)
            c = np.abs(a[:, None] - b[None, :])                                   # stand geometry
        else:
            c = rng.integers(0, 70000, (n, n))
            c[:, : n // 2] //= 1000                                               # mixed widths per column
        c = c.astype(np.int32)
        r2c, total, dual = td.assign(c, want_dual=True)
        ref = oracle.assign(c)[0]
        assert total == ref == dual, (it, n, mode, total, ref, dual)
        assert sorted(r2c.tolist()) == list(range(n))
        assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total


@pytest.mark.parametrize("n,width", [(20000, 1), (9000, 2), (5000, 4), (33000, 1)])
def test_assign_multi_chunk_finisher_paths(td, n, width):
    """Sizes where a finisher thread owns several 16-byte chunks (u8 > 16384, u16 > 8192,
    u32 > 4096 columns).  perf.jl-like instance (optimum 10*n known from the row-minimum bound),
    widened to u16 / u32 storage by one expensive column per row."""
    import torch
    from taxidispatcher_amd import _ffi
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(_ffi.lib().td_gen_uniform(n, 5, 10, 40, 0, n, cost.data_ptr()))
    if width == 2:
        cost[:, 0] = 5000          # row range > 254 -> u16 storage
    elif width == 4:
        cost[:, 0] = 250000        # big_cost sentinel -> u32 storage
    r2c, total, dual = td.assign(cost, n, want_dual=True)
    assert td.last_stats()["bytes_per_cell"] == width
    assert total == dual
    assert sorted(r2c.tolist()) == list(range(n))
    picked = cost[torch.arange(n, device="cuda"), torch.from_numpy(r2c.astype(np.int64)).cuda()]
    assert int(picked.sum().item()) == total
    if width == 1:
        assert total == 10 * n
    else:   # column 0 must be taken by exactly one row; everything else at its row minimum
        assert total == 10 * (n - 1) + (5000 if width == 2 else 250000)
    del cost
    torch.cuda.empty_cache()


def test_range_guard(td):
    """Costs spanning the whole int32 range are solved exactly while n * range fits the packed bid
    key and refused loudly (TD_ERANGE) beyond it — never a silently wrong answer."""
    rng = np.random.default_rng(6)
    n = 900
    c = rng.integers(-2**31, 2**31 - 1, (n, n)).astype(np.int32)
    r2c, total, dual = td.assign(c, want_dual=True)           # 900 * 2^32 = 3.9e12 < 4e12: still fits
    assert total == oracle.assign(c)[0] == dual
    n = 1200
    c = rng.integers(-2**31, 2**31 - 1, (n, n)).astype(np.int32)
    with pytest.raises(td.TdError, match="overflows the packed bid key"):
        td.assign(c)
    # the same size with a narrower range is fine
    c = rng.integers(0, 10**6, (n, n)).astype(np.int32)
    r2c, total = td.assign(c)
    assert total == oracle.assign(c)[0]


def test_eps_scaling_auction_mode_is_exact(td):
    """The literal eps-scaling auction (TD_SOLVER=eps, comparison mode) reaches the same optimum.
    Tunables are read once per process, so it runs in a child process."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import taxidispatcher_amd as td
from oracle import oracle
td.init(0)
rng = np.random.default_rng(4)
for n, lo, hi in [(5, 0, 9), (64, 10, 41), (200, 1, 40), (150, 0, 100000)]:
    c = rng.integers(lo, hi, (n, n)).astype(np.int32)
    r2c, tot = td.assign(c)
    assert tot == oracle.assign(c)[0], (n, tot)
    assert sorted(r2c.tolist()) == list(range(n))
print("EPS_OK", td.last_stats()["bid_rounds"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TD_SOLVER="eps", TD_EPS0_MULT="4")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "EPS_OK" in out.stdout, out.stdout + out.stderr


def test_deterministic_output(td):
    """Same input -> bit-identical row_to_col, run after run (atomicMax on packed keys, min-id
    claims and ordered free lists make the result independent of wave scheduling)."""
    rng = np.random.default_rng(77)
    for kind, n in [("g1", 1500), ("g3", 900), ("g2", 300), ("wide", 500), ("g4", 257)]:
        c = make_instance(kind, n, rng)
        first, t0 = td.assign(c)
        for _ in range(4):
            again, t1 = td.assign(c)
            assert t1 == t0 and np.array_equal(first, again), (kind, n)


def test_n65536_full_size_single_gpu(td):
    """BASELINE configs[3] size on ONE MI355X (16 GiB int32 + 4 GiB narrow copy fit 288 GB): the
    perf.jl instance at N = 65 536, optimum 10*N by the row-minimum bound, certificate on device."""
    import torch
    from taxidispatcher_amd import _ffi
    n = 65536
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(_ffi.lib().td_gen_uniform(n, 2, 10, 40, 0, n, cost.data_ptr()))
    r2c, total, dual = td.assign(cost, n, want_dual=True)
    assert total == 10 * n == dual
    assert np.array_equal(np.sort(r2c), np.arange(n))
    idx = torch.from_numpy(r2c.astype(np.int64)).cuda()
    assert int(cost[torch.arange(n, device="cuda"), idx].max().item()) == 10
    del cost
    torch.cuda.empty_cache()


def test_workspace_reuse_across_sizes(td):
    """The single-solver rule of td_assign (header): one grow-only workspace, consecutive calls of
    any sizes and widths are independent of each other."""
    rng = np.random.default_rng(77)
    seq = [("g1", 1500), ("g3", 700), ("g2", 90), ("g1", 33), ("wide", 400), ("g1", 1500), ("g4", 100), ("g3", 1301)]
    first = {}
    for kind, n in seq:
        c = make_instance(kind, n, np.random.default_rng(n * 7 + len(kind)))
        r2c, total = check_assignment(td, c)
        key = (kind, n)
        if key in first:     # the same instance again, after other sizes used the workspace: same answer, bit for bit
            assert first[key][1] == total and np.array_equal(first[key][0], r2c)
        first[key] = (r2c, total)
    del rng


def test_wide_two_byte_rows_redone_with_narrow_prices(td):
    """2-D Manhattan grid, n = 8192: rows fit 2-byte cells, the instance is wide and tie-free, so the solve is
    redone as 4-byte cells with 32-bit prices (TD_WIDE_U16_N) for the cooperative finisher; exact by the device
    certificate, the oracle on a sub-instance of the same family."""
    import torch
    rng = np.random.default_rng(31)
    for n, check_oracle in ((600, True), (8192, False)):
        ax, ay, bx, by = (rng.integers(0, 4000, n).astype(np.int32) for _ in range(4))
        c = np.abs(ax[:, None] - bx[None, :]) + np.abs(ay[:, None] - by[None, :])
        ct = torch.from_numpy(c.astype(np.int32)).cuda()
        r2c, total, dual = td.assign(ct, n, want_dual=True)
        assert total == dual
        assert sorted(r2c.tolist()) == list(range(n))
        assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total
        st = td.last_stats()
        if check_oracle:
            assert total == oracle.assign(c.astype(np.int32))[0]
        else:
            assert st["bytes_per_cell"] == 4 and st["narrow_price"] == 1 and st["warm_rounds"] > 0
