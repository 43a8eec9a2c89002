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


@pytest.mark.parametrize("kind,n", [("g1", 96), ("g2", 61), ("g3", 80)])
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
