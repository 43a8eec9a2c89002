"""GPU tests of the incremental shortest-path forest finisher (csrc/td_forest.h, k_forest): the general solver on
wide, tie-free 4-byte rows of n >= 2048 — the |a-b| geometry of greedy_opt.py:86-99,122-133 with the line-metric
recogniser switched off, 2-D grids, uniform 0..10^6 (the general S x S table of procedure.py:35-40).
Every total is checked against the oracle's exact solver (or the closed form) AND against the device's LP
certificate; last_stats()["forest_levels"] proves which finisher answered."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu
BIG = 250000


@pytest.fixture(autouse=True)
def _general_solver_only(general_solver):
    yield


def _check(td, c, expected=None):
    n = c.shape[0]
    r2c, total, dual = td.assign(c, want_dual=True)
    r2c = np.asarray(r2c)
    assert dual == total, "device LP-duality certificate does not close"
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(c[np.arange(n), r2c].astype(np.int64).sum()) == total
    if expected is None:
        expected = oracle.assign(c)[0]
    assert total == expected
    return td.last_stats()


def _line(n, rng, S=None):
    S = S or 100 * n   # range > 65 535: 4-byte cells (2-byte rows of n < 4096 go to the single-workgroup finisher)
    a, b = rng.integers(0, S, n), rng.integers(0, S, n)
    return np.abs(a[:, None] - b[None, :]).astype(np.int32), int(np.abs(np.sort(a) - np.sort(b)).sum())


@pytest.mark.parametrize("n", [2048, 2051, 2500])
def test_forest_line_geometry_vs_closed_form_and_oracle(td, n):
    """|a - b| with the recogniser off: the forest must reproduce the sorted-matching optimum;
    n = 2051 / 2500 leave the last column slice partly empty and n % 4 != 0."""
    rng = np.random.default_rng(n)
    c, expected = _line(n, rng)
    st = _check(td, c, expected)
    assert st["forest_levels"] > 0 and st["bytes_per_cell"] == 4
    if n == 2048:
        assert oracle.assign(c)[0] == expected


def test_forest_2d_grid_and_uniform_vs_oracle(td):
    rng = np.random.default_rng(11)
    n = 2304
    ax, ay, bx, by = (rng.integers(0, 40000, n) for _ in range(4))   # range > 65 535: 4-byte cells
    c = (np.abs(ax[:, None] - bx[None, :]) + np.abs(ay[:, None] - by[None, :])).astype(np.int32)
    st = _check(td, c)
    assert st["forest_levels"] > 0
    c = rng.integers(0, 10**6, (n, n)).astype(np.int32)
    st = _check(td, c)
    assert st["forest_levels"] > 0


def test_forest_asymmetric_general_table(td):
    """A general (asymmetric, non-metric) S x S table gathered by stand — PDF Table 2 is asymmetric,
    procedure.py:35-40 consumes dist as a general lookup."""
    rng = np.random.default_rng(5)
    n, S = 2200, 700
    dist = rng.integers(0, 300000, (S, S)).astype(np.int32)
    a, b = rng.integers(0, S, n), rng.integers(0, S, n)
    c = dist[a][:, b].copy()
    st = _check(td, c)
    assert st["bytes_per_cell"] in (2, 4)


def test_forest_with_64_bit_prices(td):
    """Rows whose range is beyond the narrow-price mode (> 2^22): 4-byte cells, 64-bit prices and labels."""
    rng = np.random.default_rng(6)
    n = 2048
    c = rng.integers(0, 50_000_000, (n, n)).astype(np.int32)
    st = _check(td, c)
    assert st["forest_levels"] > 0 and st["narrow_price"] == 0


def test_forest_padded_rectangular_line_model(td):
    """greedy_opt.py:88-90: n = max(cabs, requests), the missing side is big_cost — constant rows sit out the
    solve (deferred) while the forest finishes the real rows; 60 missing cabs is beyond the line recogniser's
    unbalanced plan anyway."""
    rng = np.random.default_rng(8)
    n, k = 2400, 60
    a, b = rng.integers(0, 100 * n, n - k), rng.integers(0, 100 * n, n)
    c = np.full((n, n), BIG, np.int32)
    c[: n - k] = np.abs(a[:, None] - b[None, :])
    st = _check(td, c)
    assert st["forest_levels"] > 0


def test_forest_wide_slices_above_16384(td):
    """n > 16 384: 128 columns per workgroup, predecessor walk in global memory."""
    import torch
    n = 16500
    rng = np.random.default_rng(9)
    a = rng.integers(0, 10 * n, n).astype(np.int32)
    b = rng.integers(0, 10 * n, n).astype(np.int32)
    ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
    td.cost_build(a, b, None, fill=BIG, threshold=-1, out=ct)
    r2c, total, dual = td.assign(ct, n, want_dual=True)
    expected = int(np.abs(np.sort(a.astype(np.int64)) - np.sort(b.astype(np.int64))).sum())
    assert total == expected == dual
    r2c = np.asarray(r2c)
    assert sorted(r2c.tolist()) == list(range(n))
    assert td.last_stats()["forest_levels"] > 0


def test_forest_is_deterministic(td):
    rng = np.random.default_rng(12)
    c, _ = _line(3000, rng)
    r1, t1 = td.assign(c)[:2]
    r2, t2 = td.assign(c)[:2]
    assert t1 == t2 and np.array_equal(np.asarray(r1), np.asarray(r2))
