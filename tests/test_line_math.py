"""CPU tier: the mathematics td_line.hip relies on, restated in numpy and checked against the oracle's exact
solver on random instances (no GPU, no product code).  For the reference's line metric dist[i][j] = |i - j|
(greedy_opt.py:122-127):

  balanced    the sorted matching is optimal and prices v[tau(k+1)] = v[tau(k)] + c[s(k)][tau(k+1)] - c[s(k)][tau(k)]
              make every matched cell its row's minimum of c - v (the certificate k_line_cert checks);
              the sort keys c[i][p]^2 - c[i][q]^2 order the rows by position for ANY two columns at different positions;
  unbalanced  (k constant rows) the best non-crossing plan is k prefix-min scans, the skipped columns get price 0 and
              the matched ones min(L, R) with L_i = min(0, L_{i-1} + f_{i-1}), R_i = min(0, R_{i+1} + b_i).
"""
import numpy as np

from oracle import oracle

BIG = 250000


def certificate_holds(c, r2c, v):
    red = c.astype(np.int64) - v[None, :]
    n = c.shape[0]
    return bool((red.min(axis=1) == red[np.arange(n), r2c]).all())


def balanced_plan(c):
    """keys from columns (0, q) and rows (0, i2) as k_line_probe / k_line_keys choose them; None if degenerate"""
    n = c.shape[0]
    c = c.astype(np.int64)
    d = np.abs(c[0] - c[0, 0])
    q = int(np.argmax(d))
    if d[q] == 0:
        return None
    d = np.abs(c[:, 0] - c[0, 0])
    i2 = int(np.argmax(d))
    if d[i2] == 0:
        d = np.abs(c[:, q] - c[0, q])
        i2 = int(np.argmax(d))
        if d[i2] == 0:
            return None
    rk = c[:, 0] ** 2 - c[:, q] ** 2
    ck = c[0] ** 2 - c[i2] ** 2
    if rk[i2] < rk[0]:
        ck = -ck
    sig, tau = np.argsort(rk, kind="stable"), np.argsort(ck, kind="stable")
    f = c[sig[:-1], tau[1:]] - c[sig[:-1], tau[:-1]]
    v = np.zeros(n, np.int64)
    v[tau[1:]] = np.cumsum(f)
    r2c = np.empty(n, np.int64)
    r2c[sig] = tau
    return r2c, v


def test_sorted_matching_and_prefix_prices_certify_the_optimum():
    rng = np.random.default_rng(1)
    seen = 0
    for _ in range(400):
        n = int(rng.integers(2, 60))
        S = int(rng.choice([3, 7, 50, 1000, 10**6]))
        a, b = rng.integers(0, S, n), rng.integers(0, S, n)
        c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
        plan = balanced_plan(c)
        if plan is None:      # one distinct position on a side: the general solver's trivial case
            assert len(set(a.tolist())) == 1 or len(set(b.tolist())) == 1 or S <= 7
            continue
        r2c, v = plan
        seen += 1
        assert sorted(r2c.tolist()) == list(range(n))
        assert certificate_holds(c, r2c, v)
        total = int(c[np.arange(n), r2c].astype(np.int64).sum())
        assert total == oracle.assign(c)[0] == int(np.abs(np.sort(a) - np.sort(b)).sum())
    assert seen > 300


def unbalanced_plan(a_sorted, b_sorted):
    """m = len(a) real rows, n = len(b) columns, k = n - m constant rows: shifts by prefix-min scans, prices by clamp scans"""
    m, n = len(a_sorted), len(b_sorted)
    k = n - m
    A, B = a_sorted.astype(np.int64), b_sorted.astype(np.int64)
    band = np.abs(A[:, None] - B[np.arange(m)[:, None] + np.arange(k + 1)[None, :]])
    P = np.zeros((k + 1, m + 1), np.int64)
    P[:, 1:] = np.cumsum(band.T, axis=1)
    E = P[0].copy()
    args = []
    for d in range(1, k + 1):
        g = E - P[d]
        pm = np.minimum.accumulate(g)
        arg = np.zeros(m + 1, np.int64)
        cur = 0
        for t in range(m + 1):
            if g[t] < g[cur]:
                cur = t
            arg[t] = cur
        args.append(arg)
        E = P[d] + pm
    bounds = [0] * (k + 2)
    bounds[k + 1] = m
    t = m
    for d in range(k, 0, -1):
        t = int(args[d - 1][t])
        bounds[d] = t
    shift = np.zeros(m, np.int64)
    for d in range(1, k + 1):
        shift[bounds[d]:] += 1
    mcol = np.arange(m) + shift
    skipped = np.array([bounds[d] + d - 1 for d in range(1, k + 1)], np.int64)
    C = np.full((n, n), BIG, np.int64)
    C[:m] = np.abs(A[:, None] - B[None, :])
    F = C[np.arange(m - 1), mcol[1:]] - C[np.arange(m - 1), mcol[:-1]]
    Bk = C[np.arange(1, m), mcol[:-1]] - C[np.arange(1, m), mcol[1:]]
    L, R = np.zeros(m, np.int64), np.zeros(m, np.int64)
    for i in range(1, m):
        L[i] = min(0, L[i - 1] + F[i - 1])
    for i in range(m - 2, -1, -1):
        R[i] = min(0, R[i + 1] + Bk[i])
    v = np.zeros(n, np.int64)
    v[mcol] = np.minimum(L, R)
    r2c = np.empty(n, np.int64)
    r2c[:m] = mcol
    r2c[m:] = skipped
    return C, r2c, v, int(E[m])


def test_unbalanced_plan_by_prefix_min_scans_and_clamp_prices():
    rng = np.random.default_rng(2)
    for _ in range(300):
        n = int(rng.integers(3, 48))
        k = int(rng.integers(1, min(8, n - 2) + 1))
        S = int(rng.choice([3, 10, 50, 1000]))
        a = np.sort(rng.integers(0, S, n - k))
        b = np.sort(rng.integers(0, S, n))
        C, r2c, v, plan_cost = unbalanced_plan(a, b)
        assert sorted(r2c.tolist()) == list(range(n))
        assert (v <= 0).all() and (v[r2c[n - k:]] == 0).all()
        assert certificate_holds(C, r2c, v)
        total = int(C[np.arange(n), r2c].sum())
        assert total == plan_cost + k * BIG == oracle.assign(C.astype(np.int32))[0]
