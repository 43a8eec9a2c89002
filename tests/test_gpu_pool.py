"""f-4 on the GPU: td_pool_n / td_pool_merge against the reference binary's own outputs (fixtures),
the oracle on random instances up to n = 600, and the 8-way fan-out + merge of findpool.c."""
import numpy as np
import pytest

import pool_fixtures as pf
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_pool_n_matches_reference_fixtures(td):
    for name, k in pf.cases():
        d, exp = pf.load(name, k)
        for child in range(8):
            got, nh = td.find_pool_n(k, d, child=child)
            assert got.tolist() == exp[child], (name, k, child)


def random_demand(rng, n, max_wait, losses, stands=50):
    frm = rng.integers(0, stands, n)
    to = np.clip(frm + rng.integers(1, 9, n) * rng.choice([-1, 1], n), 0, stands - 1)
    to = np.where(to == frm, np.where(frm > 0, frm - 1, 1), to)
    return np.stack([np.arange(n), frm, to, rng.integers(0, max_wait + 1, n), rng.choice(losses, n)], 1)


def test_pool_size_outside_2_to_4_is_refused(td):
    """with k = 1 the reference's duplicate test compares padded slots (pool_n.c:175-193); the library refuses
    the sizes it cannot answer like the reference instead of answering differently"""
    from taxidispatcher_amd import _ffi
    d = random_demand(np.random.default_rng(1), 20, 5, [50])
    for k in (0, 1, 5):
        with pytest.raises(_ffi.TdError):
            td.find_pool_n(k, d)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_pool_n_random_vs_oracle(td, k):
    rng = np.random.default_rng(40 + k)
    for n, mw, losses in ((4, 9, [90]), (17, 5, [10, 50]), (150, 3, [1, 30]), (600, 1 if k == 4 else 2, [1, 5, 20])):
        if n < k:
            continue
        d = random_demand(rng, n, mw if k < 4 or n < 600 else 0, losses)
        for first in ((0, n), (n // 3, 2 * n // 3)):
            exp, nh_o = oracle.pool_n(k, d[:, 1], d[:, 2], d[:, 3], d[:, 4], None, first[0], first[1], cap=3000000)
            from taxidispatcher_amd import _ffi
            import ctypes
            lib = _ffi.lib()
            cols = [np.ascontiguousarray(d[:, c].astype(np.int32)) for c in (1, 2, 3, 4)]
            out = np.zeros((n // k + 1, 2 * k + 1), np.int32)
            m, nh = ctypes.c_int32(0), ctypes.c_int64(0)
            _ffi.check(lib.td_pool_n(k, n, *[_ffi.addr(c) for c in cols], None, 0, first[0], first[1], 0, out.shape[0],
                                     _ffi.addr(out), ctypes.byref(m), ctypes.byref(nh)))
            assert nh.value == nh_o, (k, n)
            assert out[:m.value].tolist() == exp.tolist(), (k, n, first)
    # a general (asymmetric) distance table
    S = 30
    dist = rng.integers(0, 12, (S, S)).astype(np.int32)
    np.fill_diagonal(dist, 0)
    d = random_demand(rng, 80, 6, [20, 60], stands=S)
    exp, nh_o = oracle.pool_n(k, d[:, 1], d[:, 2], d[:, 3], d[:, 4], dist, 0, 80, cap=3000000)
    got, nh = td.find_pool_n(k, d, dist)
    assert nh == nh_o and got.tolist() == exp.tolist()


def test_pool_fanout_and_merge(td):
    """findpool.c: 8 children (first-pick-up slices) + merge.  Children come from the reference
    fixtures; the merge is checked against the host restatement of findpool.c's merge."""
    for name, k in pf.cases():
        d, exp = pf.load(name, k)
        n = len(d)
        lists = [td.find_pool_n(k, d, child=c)[0] for c in range(8)]
        merged = td.merge_pools(k, n, lists)
        ref = pf.merge_restatement(k, [exp[c] for c in range(8)])
        assert merged.tolist() == ref, (name, k)
        reqs = merged[:, :k].ravel().tolist()
        assert len(set(reqs)) == len(reqs)          # no request in two pools


def test_pool_n_limits(td):
    from taxidispatcher_amd import _ffi
    d = random_demand(np.random.default_rng(1), 300, 9, [90])
    with pytest.raises(_ffi.TdError):               # far more happy plans than the buffer holds
        td.find_pool_n(4, d, max_happy=1000)
    got, nh = td.find_pool_n(2, d[:1])
    assert got.shape[0] == 0
