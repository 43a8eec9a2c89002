"""GPU tests of the line-metric attempt of td_assign (td_line.hip): the reference's distance table is
dist[i][j] = |i - j| (greedy_opt.py:122-127), so a square cost matrix is |a_i - b_j| and the sorted
matching is optimal.  Accepted answers must equal the optimum; everything else must fall through to
the general solver with the same results as before."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def line_cost(rng, n, S, mode=0):
    a = rng.integers(0, S, n)
    b = rng.integers(0, S, n)
    if mode == 1:
        b = b // 3 + S            # all requests beyond every cab
    elif mode == 2:
        a[:] = a[0]               # every cab on one stand
    elif mode == 3:
        b[:] = b[0]
    elif mode == 4:
        a = a // 2 + 2 * S        # all cabs beyond every request
    return a, b, np.abs(a[:, None] - b[None, :]).astype(np.int32)


def sorted_optimum(a, b):
    return int(np.abs(np.sort(a) - np.sort(b)).sum())


@pytest.fixture()
def line_on(td):
    was = td.set_line_metric(True)
    yield td
    td.set_line_metric(was)


@pytest.mark.parametrize("S", [2, 3, 50, 1000, 10**6, 2**29])
def test_line_instances_accepted_and_optimal(line_on, S):
    td = line_on
    rng = np.random.default_rng(S % 9973)
    for n in (2, 3, 5, 17, 64, 129, 500, 1023, 1024):
        for mode in range(5):
            a, b, cost = line_cost(rng, n, S, mode)
            r2c, total, dual = td.assign(cost, want_dual=True)
            assert sorted(r2c.tolist()) == list(range(n))
            assert int(cost[np.arange(n), r2c].astype(np.int64).sum()) == total == dual == sorted_optimum(a, b), (n, S, mode)
            if mode not in (2, 3) and len(set(a.tolist())) > 1 and len(set(b.tolist())) > 1 and S > 3:
                # (one distinct position on a side = constant rows or columns: the general solver's trivial case)
                assert td.last_stats()["line_metric"] == 1, (n, S, mode)


def test_line_against_oracle_and_general_solver(line_on):
    """same instances three ways: line attempt, general solver (attempt off), CPU oracle"""
    td = line_on
    rng = np.random.default_rng(11)
    for n, S in ((40, 50), (200, 2000), (333, 50), (600, 6000)):
        a, b, cost = line_cost(rng, n, S)
        _, t_line = td.assign(cost)
        assert td.last_stats()["line_metric"] == 1
        td.set_line_metric(False)
        _, t_gen = td.assign(cost)
        assert td.last_stats()["line_metric"] == 0
        td.set_line_metric(True)
        assert t_line == t_gen == oracle.assign(cost)[0]


def test_non_line_matrices_fall_through(line_on):
    td = line_on
    rng = np.random.default_rng(12)
    for n in (2, 7, 50, 256, 700):
        cost = rng.integers(0, 1000, (n, n)).astype(np.int32)
        r2c, total, dual = td.assign(cost, want_dual=True)
        ref = oracle.assign(cost)[0]
        assert total == ref == dual
        assert int(cost[np.arange(n), r2c].sum()) == total
        if n >= 50:
            assert td.last_stats()["line_metric"] == 0


def test_perturbed_line_matrix_is_still_solved_exactly(line_on):
    """one cell off the line metric: either the certificate still holds (the optimum did not move) or the
    general solver takes over; the total is the oracle's in both cases"""
    td = line_on
    rng = np.random.default_rng(13)
    seen = set()
    for trial in range(40):
        n = int(rng.integers(5, 120))
        a, b, cost = line_cost(rng, n, 10 * n)
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        cost[i, j] = max(0, int(cost[i, j]) - int(rng.integers(1, 5 * n)))
        r2c, total = td.assign(cost)
        assert total == oracle.assign(cost)[0]
        assert int(cost[np.arange(n), r2c].sum()) == total
        seen.add(td.last_stats()["line_metric"])
    assert seen == {0, 1}


def test_padded_line_matrix(line_on):
    """unbalanced scenario (greedy_opt.py:88-90): big_cost rows / columns break the line structure"""
    td = line_on
    rng = np.random.default_rng(14)
    for nc, nr in ((30, 50), (50, 30), (200, 180)):
        S = 400
        cabs = [(i, 0, int(rng.integers(0, S))) for i in range(nc)]
        dem = [(i, int(rng.integers(0, S)), 0) for i in range(nr)]
        n, cost = td.calculate_cost(None, dem, cabs)
        r2c, total = td.assign(cost)
        assert total == oracle.assign(np.asarray(cost))[0]


def test_device_resident_line_solve_16384(line_on):
    """the bench instance class at full size: |a-b| on S = 10 N stands, written on the device"""
    import torch
    td = line_on
    n = 16384
    rng = np.random.default_rng(15)
    a = rng.integers(0, 10 * n, n).astype(np.int32)
    b = rng.integers(0, 10 * n, n).astype(np.int32)
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    td.cost_build(a, b, None, fill=250000, threshold=-1, out=cost)
    r2c, total, dual = td.assign(cost, want_dual=True)
    assert total == dual == sorted_optimum(a, b)
    assert sorted(r2c.tolist()) == list(range(n))
    assert td.last_stats()["line_metric"] == 1


def test_default_path_on_every_instance_family(line_on):
    """the parity suite's instance families through the DEFAULT configuration (attempt on)"""
    from test_gpu_parity import check_assignment, make_instance
    td = line_on
    rng = np.random.default_rng(16)
    for kind in ("g1", "g4", "g2", "g3", "wide", "neg", "const"):
        for n in (1, 2, 3, 17, 64, 200, 515):
            c = make_instance(kind, n, rng)
            check_assignment(td, c)
            if kind == "g2" and n >= 17:
                assert td.last_stats()["line_metric"] == 1


def padded_line_cost(rng, n_cabs, n_req, S, big=250000):
    """greedy_opt.py:86-99 for |i-j| stands: n = max, the missing side is big_cost"""
    a = rng.integers(0, S, n_cabs)
    b = rng.integers(0, S, n_req)
    n = max(n_cabs, n_req)
    c = np.full((n, n), big, np.int32)
    c[:n_cabs, :n_req] = np.abs(a[:, None] - b[None, :])
    return c


@pytest.mark.parametrize("side", ["fewer_cabs", "fewer_requests"])
def test_unbalanced_line_models(line_on, side):
    """k = 1..256 missing cabs (constant rows) or requests (constant columns, solved on the transpose): the
    certified plan must give the oracle's total; k = 257 is past LINE_KMAX (32 until round 3) and goes to the general solver"""
    td = line_on
    rng = np.random.default_rng(21 if side == "fewer_cabs" else 22)
    for n in (5, 12, 40, 130, 400, 515):
        for k in (1, 2, 3, 8, 17, 32, 33, 100, 256, 257):
            if n - k < 2:
                continue
            for S in (4, 60, 10 * n):
                c = padded_line_cost(rng, n - k, n, S) if side == "fewer_cabs" else padded_line_cost(rng, n, n - k, S)
                r2c, total, dual = td.assign(c, want_dual=True)
                assert sorted(r2c.tolist()) == list(range(n))
                assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total == dual
                assert total == oracle.assign(c)[0], (side, n, k, S)
                st = td.last_stats()
                if k <= 256 and S == 10 * n:
                    assert st["line_metric"] == 1 and st["line_dummies"] == k, (side, n, k, S, st)
                    assert st["transposed"] == (1 if side == "fewer_requests" else 0)
                if k == 257:
                    assert st["line_metric"] == 0


def test_unbalanced_line_large_against_general_solver(line_on):
    """n = 4096 with 3 missing cabs / 2 missing requests: certified plan vs the general solver's optimum"""
    import torch
    td = line_on
    rng = np.random.default_rng(23)
    n = 4096
    for n_cabs, n_req in ((n - 3, n), (n, n - 2)):
        c = torch.from_numpy(padded_line_cost(rng, n_cabs, n_req, 10 * n)).cuda()
        r2c, t_line, dual = td.assign(c, want_dual=True)
        assert td.last_stats()["line_metric"] == 1 and dual == t_line
        assert sorted(r2c.tolist()) == list(range(n))
        td.set_line_metric(False)
        _, t_gen = td.assign(c)
        td.set_line_metric(True)
        assert t_line == t_gen
