"""GPU tests that close the configurations BASELINE.json names but round 1 did not exercise:
configs[0] 20 x 20 through the procedure.py-shaped API, configs[2] at full size for the |a-b| and the
simulator-like families and a bit-exact cost build at 16 384^2, configs[3] with its real shard
geometry (8 row shards of 8 192 x 65 536) driven in one process.  Plus the row-window cost build,
the sharded range guard and the input validation of td_pool2."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu
BIG = 250000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _general_solver_only(general_solver):
    """This module pins the GENERAL solver: |a-b| instances would otherwise be answered by the line-metric
    attempt (sorted matching + certificate, tests/test_gpu_line.py) before the solver under test runs."""
    yield


def test_config0_20x20_procedure_api(td):
    """BASELINE configs[0] / SURVEY G0: 20 cabs x 20 requests through procedure.py's call shape
    (procedure.py:5-29): dist = |i - j| on S = 50 stands, ids 0..19, cells addressed by id, fill n*n."""
    S, n = 50, 20
    dist = np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]).astype(np.int32)
    rng = np.random.default_rng(1)
    ids = rng.permutation(n)          # ids are a permutation of the positions, as the by-id build allows
    demand = [(int(ids[i]), int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(n)]
    cabs = [(i, int(rng.integers(0, S)), int(rng.integers(0, S))) for i in range(n)]
    n_got, cost = td.calculate_cost_by_id(dist, demand, cabs)
    exp = np.full((n, n), n * n, np.int32)
    for c in cabs:
        for d in demand:
            exp[c[0], d[0]] = dist[c[2], d[1]]            # procedure.py:12
    assert n_got == n and np.array_equal(cost, exp)
    _, cost_o = oracle.cost_build_by_id([c[0] for c in cabs], [c[2] for c in cabs], [d[0] for d in demand],
                                        [d[1] for d in demand], dist)
    assert np.array_equal(cost_o, exp)
    x = td.procedure_solve(dist, demand, cabs)
    assert x.shape == (n * n,) and int(x.sum()) == n
    r2c = np.nonzero(x.reshape(n, n) == 1)[1]             # procedure.py:54-57 reads x[n*cab + cust]
    assert sorted(r2c.tolist()) == list(range(n))
    total = int(exp[np.arange(n), r2c].sum())
    assert total == oracle.assign(exp)[0]


def test_cost_build_bit_exact_16384(td):
    """a-2 at the headline size: every one of the 2^28 cells against a torch expression on the device
    (|a-b| analytic, thresholded S = 50 table)."""
    import torch
    from taxidispatcher_amd import _ffi
    n = 16384
    rng = np.random.default_rng(5)
    a = rng.integers(0, 10 * n, n).astype(np.int32)
    b = rng.integers(0, 10 * n, n).astype(np.int32)
    out = torch.empty((n, n), dtype=torch.int32, device="cuda")
    td.cost_build(a, b, None, fill=BIG, threshold=-1, out=out)
    exp = (torch.from_numpy(a).cuda()[:, None] - torch.from_numpy(b).cuda()[None, :]).abs()
    assert bool((out == exp).all())
    del exp
    # thresholded table variant (Simulator.java:493-520): 600 more cabs than requests
    S = 50
    dist = rng.integers(0, 30, (S, S)).astype(np.int32)
    a = rng.integers(0, S, n).astype(np.int32)
    b = rng.integers(0, S, n - 600).astype(np.int32)
    td.cost_build(a, b, dist, fill=BIG, threshold=10, out=out)
    dt = torch.from_numpy(dist).cuda()
    v = dt[torch.from_numpy(a).cuda().long()[:, None], torch.from_numpy(b).cuda().long()[None, :]]
    exp = torch.full((n, n), BIG, dtype=torch.int32, device="cuda")
    exp[:, :n - 600] = torch.where(v < 10, v, torch.full_like(v, BIG))
    assert bool((out == exp).all())
    _ffi.check(_ffi.lib().td_synchronize())


def test_cost_build_rows_window(td):
    """a row shard builds its block in place from the replicated position arrays."""
    from taxidispatcher_amd import _ffi
    lib = _ffi.lib()
    rng = np.random.default_rng(6)
    for n_s, n_d, S, thr, table in ((700, 531, 50, 10, True), (1000, 1000, 10000, -1, False), (333, 800, 40, -1, True)):
        a = rng.integers(0, S, n_s).astype(np.int32)
        b = rng.integers(0, S, n_d).astype(np.int32)
        dist = rng.integers(0, 60, (S, S)).astype(np.int32) if table else None
        n, full = oracle.cost_build(a, b, dist, BIG, thr)
        for world in (2, 3, 8):
            rps = (n + world - 1) // world
            for r in range(world):
                row0 = min(n, r * rps)
                nrows = max(0, min(rps, n - row0))
                blk = np.full((max(nrows, 1), n), -7, np.int32)
                _ffi.check(lib.td_cost_build_rows(_ffi.addr(a), None, n_s, _ffi.addr(b), None, n_d,
                                                  None if dist is None else dist.ctypes.data, 0 if dist is None else S,
                                                  BIG, thr, 0, row0, nrows, _ffi.addr(blk)))
                if nrows:
                    assert np.array_equal(blk[:nrows], full[row0:row0 + nrows]), (n_s, n_d, world, r)
    # by-id variant: the window applies to the cab ids
    n = 40
    ids_c, ids_d = rng.permutation(n).astype(np.int32), rng.permutation(n).astype(np.int32)
    a = rng.integers(0, 50, n).astype(np.int32)
    b = rng.integers(0, 50, n).astype(np.int32)
    _, full = oracle.cost_build_by_id(ids_c, a, ids_d, b, None)
    blk = np.empty((10, n), np.int32)
    _ffi.check(lib.td_cost_build_rows(_ffi.addr(a), _ffi.addr(ids_c), n, _ffi.addr(b), _ffi.addr(ids_d), n, None, 0,
                                      n * n, -1, 1, 20, 10, _ffi.addr(blk)))
    assert np.array_equal(blk, full[20:30])
    assert lib.td_cost_build_rows(_ffi.addr(a), None, n, _ffi.addr(b), None, n, None, 0, BIG, -1, 0, 35, 10,
                                  _ffi.addr(blk)) == -1   # window outside the model: TD_EINVAL


@pytest.mark.parametrize("kind", ["g2", "g2-line", "g3"])
def test_headline_size_other_families(td, kind):
    """SURVEY 8d lists G2 (|a-b|, greedy_opt.py's own cost model) and G3 (simulator-like) for the
    headline size N = 16 384: exact total with the device certificate; G2 also against the closed form,
    by the general solver ("g2") and by the default path with the line-metric attempt on ("g2-line")."""
    import torch
    n = 16384
    if kind == "g2-line":
        td.set_line_metric(True)    # the module fixture restores the setting
        kind = "g2"
    rng = np.random.default_rng(1)
    ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
    if kind == "g2":
        a = rng.integers(0, 10 * n, n).astype(np.int32)
        b = rng.integers(0, 10 * n, n).astype(np.int32)
        td.cost_build(a, b, None, fill=BIG, threshold=-1, out=ct)
        expect = int(np.abs(np.sort(a).astype(np.int64) - np.sort(b).astype(np.int64)).sum())
    else:
        a = rng.integers(0, 50, n).astype(np.int32)
        b = rng.integers(0, 50, int(0.363 * n)).astype(np.int32)
        td.cost_build(a, b, None, fill=BIG, threshold=10, out=ct)
        expect = None
    r2c, total, dual = td.assign(ct, want_dual=True)
    assert total == dual
    assert sorted(r2c.tolist()) == list(range(n))
    got = int(ct[torch.arange(n, device="cuda"), torch.from_numpy(r2c).cuda().long()].sum().item())
    assert got == total
    if expect is not None:
        assert total == expect
    else:   # every real cell costs < 10, so the number of dummy cells in the optimum is minimal: total mod BIG < BIG
        assert total % BIG < 10 * n


def test_config3_shard_geometry_in_process(td):
    """BASELINE configs[3] with its REAL geometry: 8 row shards of 8 192 x 65 536 (2 GiB int32 each) in one process,
    driven through the steps of solve_sharded (torch.cat / torch.maximum stand in for the collectives).  Both sequences:
    the block-local start (csrc/td_blocks.h: phase A on every shard's diagonal block, ONE exchange, then whatever is
    left) and the plain one (a MAX all-reduce of the packed keys per round, finisher on rank 0).  Each must give 10*N
    with a closing certificate and the same row_to_col as td_assign with the same block setting."""
    import torch
    from taxidispatcher_amd import _ffi, sharded
    lib = _ffi.lib()
    n, world = 65536, 8
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2**30:
        pytest.skip("needs ~45 GiB of free HBM")
    full = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(lib.td_gen_uniform(n, 7, 10, 40, 0, n, full.data_ptr()))
    _ffi.check(lib.td_synchronize())
    for blocks in (True, False):
        shards = []
        try:
            for r in range(world):
                row0, nrows, rps = sharded.shard_bounds(n, world, r)
                shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
            r2c, tot, dual, info = sharded.solve_shards_in_process(shards, blocks=blocks)
        finally:
            for s in shards:
                s.close()
        assert info["path"] == ("blocks" if blocks else "auction")
        assert tot == 10 * n == dual
        assert sorted(r2c.tolist()) == list(range(n))
        was = lib.td_set_blocks(8 if blocks else 0)
        try:
            ref, ref_total = td.assign(full)
        finally:
            lib.td_set_blocks(was)
        assert ref_total == 10 * n
        if not blocks or info["left"] == 0:   # (rows the blocks leave free go to the rounds here, to a two-hop pass over the whole matrix in td_assign)
            assert np.array_equal(ref, r2c), "sharded and unsharded runs must be bit-identical"


def test_shard_range_guard(td):
    """the sharded path must refuse what td_assign refuses (TD_ERANGE): a row range that
    would overflow the packed (price << 20 | row) bid key."""
    from taxidispatcher_amd import _ffi, sharded
    n = 2048
    rng = np.random.default_rng(3)
    c = rng.integers(0, 2**31 - 2, (n, n)).astype(np.int32)
    with pytest.raises(_ffi.TdError):
        td.assign(c)
    sh = sharded.HipShard(n, 0, n, c, share_torch_stream=False)
    try:
        for width in (1, 2, 4):
            if sh.compress(width):
                break
        assert sh.range() > 2**30
        with pytest.raises(_ffi.TdError):
            sh.begin(sh.range())
        with pytest.raises(_ffi.TdError):
            sh.begin(-1)
    finally:
        sh.close()
    # a range that fits passes, and the reduced range of another rank is honoured
    c2 = rng.integers(0, 1000, (n, n)).astype(np.int32)
    sh = sharded.HipShard(n, 0, n, c2, share_torch_stream=False)
    try:
        for width in (1, 2, 4):
            if sh.compress(width):
                break
        assert 0 < sh.range() <= 65534
        sh.begin(sh.range())
        with pytest.raises(_ffi.TdError):
            sh.begin(2**31)
    finally:
        sh.close()


def test_pool2_rejects_stands_outside_the_table(td):
    from taxidispatcher_amd import _ffi
    S = 20
    dist = np.abs(np.arange(S)[:, None] - np.arange(S)[None, :]).astype(np.int32)
    frm = np.array([1, 5, 19, 3], np.int32)
    to = np.array([2, 25, 0, 7], np.int32)      # 25 is not a stand of the table
    with pytest.raises(_ffi.TdError):
        td.find_pool(frm, to, dist)
    to[1] = 6
    assert len(td.find_pool(frm, to, dist)) == 2


def test_count_sum_single_row(td):
    """nn == 1, x = [1] must count cost[0][0] (a 1 x 1 LCM remainder in combined())."""
    assert td.count_sum(1, np.array([[7]], np.int32), np.array([1])) == 7
    assert td.count_sum(1, np.array([[BIG]], np.int32), np.array([1])) == 0


def test_wrappers_refuse_wrong_dtype_or_strided_tensors(td):
    """torch CUDA tensors go to the kernels as they are, so the wrappers check them."""
    import torch
    from taxidispatcher_amd import _ffi
    c64 = torch.zeros((8, 8), dtype=torch.int64, device="cuda")
    with pytest.raises(_ffi.TdError):
        td.assign(c64)
    c = torch.randint(0, 50, (16, 16), dtype=torch.int32, device="cuda")
    with pytest.raises(_ffi.TdError):
        td.assign(c.t())                       # a transposed view is not contiguous
    r2c, total = td.assign(c.t().contiguous())
    assert total == oracle.assign(c.t().cpu().numpy())[0]
    with pytest.raises(_ffi.TdError):
        td.find_pool(np.array([1, 2, 3]), np.array([2, 3, 4]), torch.zeros((5, 5), dtype=torch.float32, device="cuda"))


def test_narrow_price_mode_and_its_fallback(td):
    """4-byte cells with a small row range are solved with 32-bit prices and labels first; a price at
    the limit voids the attempt and the solve is redone with 64-bit prices (same optimum).  The limit
    is lowered through TD_NP_PLIMIT in a child process to force the fallback."""
    rng = np.random.default_rng(12)
    n = 1500
    a, b = rng.integers(0, 10 * n, n), rng.integers(0, 10 * n, n)
    c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    c[:, 0] += 70000                                   # range > 65 534: 4-byte cells
    ref = oracle.assign(c)[0]
    r2c, total, dual = td.assign(c, want_dual=True)
    assert total == ref == dual
    st = td.last_stats()
    assert st["bytes_per_cell"] == 4 and st["narrow_price"] == 1
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import taxidispatcher_amd as td
td.init(0)
rng = np.random.default_rng(12)
n = 1500
a, b = rng.integers(0, 10 * n, n), rng.integers(0, 10 * n, n)
c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
c[:, 0] += 70000
r2c, total, dual = td.assign(c, want_dual=True)
st = td.last_stats()
print("RESULT", total, dual, st["bytes_per_cell"], st["narrow_price"])
''' % ROOT
    env = dict(os.environ, TD_NP_PLIMIT="2000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == ref == int(line[2])
    assert line[3] == "4" and line[4] == "0"          # redone with 64-bit prices
