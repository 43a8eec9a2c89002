"""CPU test of the multi-GPU (row-sharded) path: world_size 2 and 3 over gloo on 127.0.0.1.
The collective driver is the product code (taxidispatcher_amd.sharded.solve_sharded); the
per-rank device work is replaced by the numpy shard model of tests/shard_model.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _instance(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "g1":
        return rng.integers(10, 41, (n, n)).astype(np.int32)
    if kind == "g3":
        a, b = rng.integers(0, 50, n), rng.integers(0, 50, n)
        c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
        c[c >= 10] = 250000
        c[:, max(1, int(0.4 * n)):] = 250000
        return c
    if kind == "padded":   # greedy_opt.py:88-90: fewer cabs than requests, the missing cabs are rows of big_cost
        c = rng.integers(10, 41, (n, n)).astype(np.int32)
        c[rng.permutation(n)[:n // 3]] = 250000
        return c
    a, b = rng.integers(0, 10 * n, n), rng.integers(0, 10 * n, n)
    return np.abs(a[:, None] - b[None, :]).astype(np.int32)


def _worker(rank, world, port, kind, n, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shard_model import ModelShard
    from taxidispatcher_amd import sharded
    cost = _instance(kind, n, seed)
    row0, nrows, rps = sharded.shard_bounds(n, world, rank)
    sh = ModelShard(n, row0, nrows, cost[row0:row0 + nrows])
    r2c, total, dual = sharded.solve_sharded(sh, dist, rounds=8, want_dual=True)
    q.put((rank, row0, r2c.tolist(), total, dual))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, kind, n, seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort()
    r2c = [c for _, _, part, _, _ in outs for c in part]
    totals = {t for *_, t, _ in outs}
    duals = {d for *_, d in outs}
    assert len(totals) == 1 and len(duals) == 1
    return np.array(r2c), totals.pop(), duals.pop()


def test_shard_bounds():
    from taxidispatcher_amd.sharded import shard_bounds
    for n, w in [(10, 3), (65536, 8), (7, 8), (16384, 1)]:
        rows = [shard_bounds(n, w, r) for r in range(w)]
        assert sum(nr for _, nr, _ in rows) == n
        assert all(r0 == min(n, r * rps) for r, (r0, _, rps) in enumerate(rows))
        assert len({rps for *_, rps in rows}) == 1


@pytest.mark.parametrize("kind,n", [("g1", 96), ("g2", 61), ("g3", 80), ("padded", 70)])
def test_sharded_equals_single_and_oracle(kind, n):
    from oracle import oracle
    cost = _instance(kind, n, 5)
    ref = oracle.assign(cost)[0]
    r1, t1, d1 = _run(1, kind, n, 5)
    r2, t2, d2 = _run(2, kind, n, 5)
    assert t1 == t2 == ref and d1 == d2 == ref
    # MAX-reduced packed keys make the sharded run bit-identical to the single-rank run
    assert np.array_equal(r1, r2)
    assert sorted(r2.tolist()) == list(range(n))
    assert int(cost[np.arange(n), r2].sum()) == ref


def test_three_ranks_uneven_shards():
    from oracle import oracle
    n = 50  # rps = 17: shards of 17, 17, 16
    cost = _instance("g1", n, 9)
    r3, t3, d3 = _run(3, "g1", n, 9)
    assert t3 == oracle.assign(cost)[0] == d3
    assert sorted(r3.tolist()) == list(range(n))


@pytest.mark.parametrize("world,kind,n,path", [(2, "g2", 61, "line"), (3, "g2", 50, "line"), (2, "g1", 42, "auction"),
                                               (2, "g3", 64, "auction"), (3, "g2dup", 47, "line"),
                                               (2, "g1", 40, "blocks"), (4, "g1", 64, "blocks"), (2, "padded", 72, "blocks"),
                                               (8, "padded", 48, "blocks")])
def test_sharded_line_path_over_ranks(world, kind, n, path):
    """the sorted matching over row shards (four SUM all-reduces of O(n) words): taken for |a - b| matrices, refused
    (and the auction run) for the others; duplicates in the positions do not matter.  "blocks": the 1-byte attempt
    started on every rank's own diagonal blocks and the ranks met in ONE all-gather (csrc/td_blocks.h; n a multiple of
    8, 8 a multiple of the world size); g3 does not fit one byte: every rank learns it from the same all-gather and the
    ordinary sequence runs from the 2-byte width"""
    from oracle import oracle
    if kind == "g2dup":
        rng = np.random.default_rng(3)
        a, b = rng.integers(0, 12, n), rng.integers(0, 12, n)
        cost = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    else:
        cost = _instance(kind, n, 11)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_line_worker_inst, args=(r, world, port, cost, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r2c = np.array([c for _, part, _, _, _ in outs for c in part])
    ref = oracle.assign(cost)[0]
    assert {t for _, _, t, _, _ in outs} == {ref} and {d for _, _, _, d, _ in outs} == {ref}
    assert {pp for *_, pp in outs} == {path}
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(cost[np.arange(n), r2c].sum()) == ref


def _line_worker_inst(rank, world, port, cost, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shard_model import ModelShard
    from taxidispatcher_amd import sharded
    n = cost.shape[0]
    row0, nrows, rps = sharded.shard_bounds(n, world, rank)
    sh = ModelShard(n, row0, nrows, cost[row0:row0 + nrows])
    r2c, total, dual = sharded.solve_sharded(sh, dist, rounds=8, want_dual=True)
    q.put((rank, r2c.tolist(), total, dual, sharded.solve_sharded.last_path))
    dist.barrier()
    dist.destroy_process_group()


def _pool_worker(rank, world, port, name, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pool_fixtures as pf
    from oracle import oracle
    from taxidispatcher_amd import sharded
    d, _ = pf.load(name, k)
    n = len(d)

    def finder(kk, dd, child):      # host model of td_pool_n for one child slice
        a, b = pf.child_slice(n, child)
        return oracle.pool_n(kk, dd[:, 1], dd[:, 2], dd[:, 3], dd[:, 4], None, a, b)[0]

    out = sharded.pool_fanout(k, d, dist, finder=finder, merger=lambda kk, nn, lists: pf.merge_restatement(kk, lists))
    q.put((rank, out.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _pool_dead_rank_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import pool_fixtures as pf
    from oracle import oracle
    from taxidispatcher_amd import _ffi, sharded
    d, _ = pf.load("a40", 2)
    n = len(d)

    def finder(kk, dd, child):
        a, b = pf.child_slice(n, child)
        return oracle.pool_n(kk, dd[:, 1], dd[:, 2], dd[:, 3], dd[:, 4], None, a, b)[0]

    if rank == world - 1:          # the "dead" rank: never calls the fan-out (findpool.c: a child that never raises its flag)
        time.sleep(6)
        q.put((rank, "silent"))
    else:
        t0 = time.time()
        try:
            sharded.pool_fanout(2, d, dist, finder=finder, merger=lambda kk, nn, lists: pf.merge_restatement(kk, lists), timeout=2.0)
            q.put((rank, "returned"))
        except _ffi.TdError as e:
            q.put((rank, "TdError after %.1f s: %s" % (time.time() - t0, e)))
    # no barrier: the silent rank must not be waited for by a collective either


def test_pool_fanout_gives_up_on_a_dead_rank():
    """findpool.c:149-169: the parent waits 60 s for its children's flags, then 'ERROR: not all threads have returned
    results'.  pool_fanout's waits are bounded the same way: with one rank silent the others raise within the timeout
    (2 s here) and name it, instead of blocking in a collective."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pool_dead_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert outs[2] == "silent"
    for r in (0, 1):
        assert outs[r].startswith("TdError after"), outs[r]
        assert float(outs[r].split()[2]) < 12.0
    assert "[2]" in outs[0]


@pytest.mark.parametrize("world", [2, 3])
def test_pool_fanout_over_ranks(world):
    """f-4 fan-out (findpool.c's 8 children) over `world` ranks: every rank ends with the merge of
    the reference's own child outputs."""
    import pool_fixtures as pf
    name, k = "b120", 4
    _, exp = pf.load(name, k)
    ref = pf.merge_restatement(k, [exp[c] for c in range(8)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pool_worker, args=(r, world, port, name, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got in outs:
        assert got == ref


def _lcm_worker(rank, world, port, n, seed, variant, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shard_model import ModelLcmShard
    from taxidispatcher_amd import sharded
    cost = _lcm_instance(n, seed)
    row0, nrows, _ = sharded.shard_bounds(n, world, rank)
    kw = _LCM_VARIANTS[variant]
    sh = ModelLcmShard(n, row0, nrows, cost[row0:row0 + nrows], kw.get("stop_value_on", 0), kw.get("stop_value", 0))
    out = sharded.lcm_sharded([sh], dist, n, **kw)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _lcm_instance(n, seed):
    rng = np.random.default_rng(seed)
    a, b = rng.integers(0, 50, n), rng.integers(0, 50, max(1, int(0.7 * n)))
    c = np.full((n, n), 250000, np.int32)
    d = np.abs(a[:, None] - b[None, :])
    c[:, :b.size] = np.where(d < 10, d, 250000)
    return c


_LCM_VARIANTS = {
    "greedy_opt": dict(mask=250000, threshold=10, sum_below=250000),                                   # greedy_opt.py:61-82
    "simulator": dict(mask=250000, stop_value_on=1, stop_value=250000, stop_size=20, sum_below=250000),  # Simulator.java:523-549
    "heuristic": dict(mask=100),                                                                        # heuristic.py:24-33
}


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("variant", ["greedy_opt", "simulator", "heuristic"])
def test_lcm_sharded_equals_oracle(world, variant):
    """SURVEY 8e sharded LCM: the collective driver (product code) over numpy shard models gives the
    oracle's pair list, total and last_min."""
    from oracle import oracle
    n, seed = 57, 11
    cost = _lcm_instance(n, seed)
    if variant == "heuristic":
        cost = np.random.default_rng(seed).integers(1, 40, (n, n)).astype(np.int32)
        globals()["_lcm_instance"] = lambda nn, ss: np.random.default_rng(ss).integers(1, 40, (nn, nn)).astype(np.int32)
    kw = _LCM_VARIANTS[variant]
    tot_o, rows_o, cols_o, lm_o = oracle.lcm(cost, mask=kw["mask"], threshold=kw.get("threshold", -1),
                                             stop_value_on=kw.get("stop_value_on", 0), stop_value=kw.get("stop_value", 0),
                                             stop_size=kw.get("stop_size", -1), sum_below=kw.get("sum_below", 2**62),
                                             java_scan=kw.get("stop_value_on", 0))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lcm_worker_v, args=(r, world, port, n, seed, variant, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, (tot, rows, cols, lm) in outs:
        assert tot == tot_o and rows == rows_o.tolist() and cols == cols_o.tolist()
        assert lm == lm_o


def _lcm_worker_v(rank, world, port, n, seed, variant, q):
    if variant == "heuristic":
        globals()["_lcm_instance"] = lambda nn, ss: np.random.default_rng(ss).integers(1, 40, (nn, nn)).astype(np.int32)
    _lcm_worker(rank, world, port, n, seed, variant, q)
