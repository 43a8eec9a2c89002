"""GPU test of the row-sharded path with the REAL device shards (HipShard through the C ABI):
two processes share GPU 0, host collectives over gloo (RCCL needs one GPU per rank; the driver
exercises that at round end with bench.py --gpus N).  Covers the hipIpc peer mapping between
processes and the gather fallback."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(q, procs, count, limit=240):
    """results from the workers; fails fast when a worker died instead of waiting for the queue"""
    import queue
    import time
    outs, t0 = [], time.time()
    while len(outs) < count:
        try:
            outs.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > limit:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError("sharded worker failed: exit codes %s" % [p.exitcode for p in procs])
    return sorted(outs)


def _padded_rows(n, seed):
    """rows of big_cost (dummy cabs, greedy_opt.py:88-90) in a uniform matrix: a third of the rows, anywhere"""
    return np.random.default_rng(seed).permutation(n)[:n // 3]


def _worker(rank, world, port, n, seed, use_ipc, q, kind="g1"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import taxidispatcher_amd as td
    from taxidispatcher_amd import sharded
    from oracle import oracle
    td.init(0)
    row0, nrows, rps = sharded.shard_bounds(n, world, rank)
    rows = oracle.gen_uniform(n, seed, 10, 40, row0, nrows)
    if kind == "padded":
        dummy = _padded_rows(n, seed)
        rows[dummy[(dummy >= row0) & (dummy < row0 + nrows)] - row0] = 250000
    rows = torch.from_numpy(rows).cuda()
    sh = sharded.HipShard(n, row0, nrows, rows)
    try:
        # three solves on one shard object, as a tick loop or bench.py does: the peer mapping of rank 1's shard is
        # opened once and reused, the results are the same every time
        first = None
        for rep in range(3):
            r2c, total, dual = sharded.solve_sharded(sh, dist, want_dual=True, use_ipc=use_ipc)
            if first is None:
                first = (r2c.tolist(), total, dual)
            assert (r2c.tolist(), total, dual) == first
        if use_ipc and rank == 0:
            # (the peer's first compress pass may still grow its workspace once: at most two handles per peer)
            assert world - 1 <= sh.ipc_opens <= 2 * (world - 1), sh.ipc_opens
    finally:
        sh.close()
    q.put((rank, r2c.tolist(), total, dual))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,use_ipc,kind", [(512, True, "g1"), (2050, True, "g1"), (1000, False, "g1"), (1500, True, "padded"),
                                            (300, False, "padded")])
def test_two_ranks_one_gpu(n, use_ipc, kind):
    """kind "padded": a third of the rows are dummy cabs (constant rows): they sit out the rounds and the searches on
    both ranks (td_shard_const_rows) and take the left-over columns on the finisher's rank"""
    from oracle import oracle
    seed = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, seed, use_ipc, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    outs = _collect(q, procs, 2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cost = oracle.gen_uniform(n, seed, 10, 40)
    if kind == "padded":
        cost[_padded_rows(n, seed)] = 250000
    ref = oracle.assign(cost)[0]
    r2c = np.array(outs[0][1] + outs[1][1])
    assert outs[0][2] == outs[1][2] == ref
    assert outs[0][3] == outs[1][3] == ref      # duality certificate summed over shards
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(cost[np.arange(n), r2c].sum()) == ref


def test_single_rank_shard_api_matches_td_assign(td):
    """world = 1 through the shard API == td_assign (same kernels, same keys)."""
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from taxidispatcher_amd import sharded
    if not dist.is_initialized():
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
    n = 777
    cost = oracle.gen_uniform(n, 11, 10, 40)
    sh = sharded.HipShard(n, 0, n, torch.from_numpy(cost).cuda())
    try:
        r2c, total, dual = sharded.solve_sharded(sh, dist, want_dual=True)
    finally:
        sh.close()
    r_ref, t_ref = td.assign(cost)
    assert total == t_ref == dual == oracle.assign(cost)[0]
    assert np.array_equal(r2c, r_ref)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_lcm_sharded_bit_identical_to_td_lcm(td):
    """SURVEY 8e sharded LCM on the GPU: 1, 3 and 8 row shards (td_lcm_shard_*) driven in one
    process give td_lcm's pair list, total and last_min for the reference variants."""
    import numpy as np
    from taxidispatcher_amd import dispatch, sharded
    rng = np.random.default_rng(21)
    BIG = 250000
    # Simulator.java tick shape: 1300 cabs x 900 requests, stop at 600 left / on big_cost
    a, b = rng.integers(0, 50, 1300), rng.integers(0, 50, 900)
    n, cost = td.cost_build(a, b, None, fill=BIG, threshold=10)
    variants = [
        (cost, dict(mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=600, sum_below=BIG)),    # Simulator.java:523-549
        (cost, dict(mask=BIG, threshold=10, sum_below=BIG)),                                      # greedy_opt.py:61-82
        (rng.integers(1, 40, (100, 100)).astype(np.int32), dict(mask=100)),                       # heuristic.py:24-33
        (cost[:333, :333].copy(), dict(mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=0, sum_below=BIG)),
    ]
    for c, kw in variants:
        n = c.shape[0]
        ref = dispatch._lcm(n, c, kw["mask"], kw.get("threshold", -1), kw.get("stop_value_on", 0), kw.get("stop_value", 0),
                            kw.get("stop_size", -1), kw.get("sum_below", 2**62))
        for world in (1, 3, 8):
            shards = []
            try:
                for r in range(world):
                    row0, nrows, _ = sharded.shard_bounds(n, world, r)
                    shards.append(sharded.HipLcmShard(n, row0, nrows, np.ascontiguousarray(c[row0:row0 + nrows]),
                                                      kw.get("stop_value_on", 0), kw.get("stop_value", 0)))
                tot, rows, cols, lm = sharded.lcm_sharded(shards, None, n, **kw)
                rounds = sharded.lcm_sharded.last_rounds
            finally:
                for s in shards:
                    s.close()
            assert tot == ref[0] and rows == ref[1].tolist() and cols == ref[2].tolist() and lm == ref[3], (n, world)
            # rounds of locally dominant cells: tens of exchanges where the per-pick driver needs one per pair
            assert rounds <= 64 and (len(rows) < 100 or rounds < len(rows) // 4), (rounds, len(rows))
        # the one-exchange-per-pick driver (the round driver's comparator) on 3 shards
        shards = []
        try:
            for r in range(3):
                row0, nrows, _ = sharded.shard_bounds(n, 3, r)
                shards.append(sharded.HipLcmShard(n, row0, nrows, np.ascontiguousarray(c[row0:row0 + nrows]),
                                                  kw.get("stop_value_on", 0), kw.get("stop_value", 0)))
            tot, rows, cols, lm = sharded.lcm_sharded(shards, None, n, by_pick=True, **kw)
        finally:
            for s in shards:
                s.close()
        assert tot == ref[0] and rows == ref[1].tolist() and cols == ref[2].tolist() and lm == ref[3], (n, "by pick")


def _lcm_rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        import taxidispatcher_amd as td
        from taxidispatcher_amd import dispatch, sharded
        td.init(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        rng = np.random.default_rng(3)
        a, b = rng.integers(0, 50, 700), rng.integers(0, 50, 500)
        n, cost = td.cost_build(a, b, None, fill=250000, threshold=10)
        kw = dict(mask=250000, stop_value_on=1, stop_value=250000, stop_size=300, sum_below=250000)
        ref = dispatch._lcm(n, cost, kw["mask"], -1, 1, kw["stop_value"], kw["stop_size"], kw["sum_below"])
        res = []
        for by_pick in (False, True):
            sh = sharded.HipLcmShard(n, 0, n, cost, 1, kw["stop_value"])
            try:
                tot, rows, cols, lm = sharded.lcm_sharded([sh], dist, n, by_pick=by_pick, force_collectives=True, **kw)
            finally:
                sh.close()
            res.append(bool(tot == ref[0] and rows == ref[1].tolist() and cols == ref[2].tolist() and lm == ref[3]))
        dist.destroy_process_group()
        q.put(("ok", res))
    except Exception as e:   # noqa: BLE001 — the parent asserts on the message
        q.put(("error", repr(e)))


@pytest.mark.gpu
def test_lcm_sharded_through_rccl_one_rank(td):
    """the sharded LCM's exchanges on an RCCL-only process group (backend nccl, one rank, collectives
    forced): the key vectors are device tensors, so both drivers run where a CPU tensor would raise 'No backend type
    associated with device type cpu'.  Own process: this one may already hold a gloo group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_lcm_rccl_worker, args=(_free_port(), q))
    p.start()
    status, res = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", res
    assert res == [True, True]


def _rounds_worker(port, native, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TD_SHARD_FORCE_AR"] = "1"          # issue ncclAllReduce with one rank too (read once per process)
    os.environ["TD_SHARD_NATIVE"] = "1" if native else "0"
    try:
        import torch
        import torch.distributed as dist
        import taxidispatcher_amd as td
        from oracle import oracle
        from taxidispatcher_amd import sharded
        td.init(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        n = 3000
        cost = torch.from_numpy(oracle.gen_uniform(n, 5, 10, 40)).cuda()
        sh = sharded.HipShard(n, 0, n, cost)
        try:
            r2c, total, dual = sharded.solve_sharded(sh, dist, want_dual=True)
        finally:
            sh.close()
        sharded.HipShard.destroy_comm()
        dist.destroy_process_group()
        q.put(("ok", (np.asarray(r2c).tolist(), int(total), int(dual))))
    except Exception as e:   # noqa: BLE001
        q.put(("error", repr(e)))


@pytest.mark.gpu
def test_native_rccl_rounds_equal_python_rounds(td):
    """td_shard_rounds calls ncclAllReduce through dlsym with hand-copied enum values (uint64, max).  With
    TD_SHARD_FORCE_AR=1 the collective is issued with one rank too: the native loop (bid -> ncclAllReduce -> apply on the
    library's stream) must give the same row_to_col and total as the Python loop over torch.distributed."""
    ctx = mp.get_context("spawn")
    res = []
    for native in (True, False):
        q = ctx.Queue()
        p = ctx.Process(target=_rounds_worker, args=(_free_port(), native, q))
        p.start()
        status, out = q.get(timeout=300)
        p.join(timeout=60)
        assert status == "ok", out
        res.append(out)
    assert res[0][1] == res[1][1] == 10 * 3000 == res[0][2] == res[1][2]
    assert res[0][0] == res[1][0], "native RCCL rounds and the Python rounds must be bit-identical"


@pytest.mark.gpu
def test_bench_sharded_leg_through_rccl(td):
    """The multi-GPU leg of bench.py with one rank: process group on the nccl backend, the library's
    own RCCL communicator (td_comm_init), td_shard_rounds, finisher, totals — and the JSON contract."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-sharded", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras", "--sharded-n", "8192", "--n", "4096"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["total_cost"] == 10 * 4096
    sh = line["sharded_single_instance"]
    assert "error" not in sh, sh
    assert sh["optimal"] and sh["total_cost"] == 10 * 8192 and sh["speedup_vs_single_gpu"] > 0


def _blocks_rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TD_SHARD_FORCE_AR"] = "1"          # issue the collectives with one rank too
    try:
        import torch
        import torch.distributed as dist
        import taxidispatcher_amd as td
        from taxidispatcher_amd import _ffi, sharded
        td.init(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        n = 12288
        rows = torch.empty((n, n), dtype=torch.int32, device="cuda")
        _ffi.check(_ffi.lib().td_gen_uniform(n, 3, 10, 40, 0, n, rows.data_ptr()))
        sh = sharded.HipShard(n, 0, n, rows)
        try:
            outs = [sharded.solve_sharded(sh, dist, want_dual=True) for _ in range(2)]   # twice: the workspace and the stream are reused
            path, left = sharded.solve_sharded.last_path, sharded.solve_sharded.last_left
        finally:
            sh.close()
        ref, ref_total = td.assign(rows)
        sharded.HipShard.destroy_comm()
        dist.destroy_process_group()
        q.put(("ok", (path, left, [int(o[1]) for o in outs], [int(o[2]) for o in outs], bool(np.array_equal(outs[0][0], outs[1][0])),
                      bool(np.array_equal(outs[0][0], ref)), int(ref_total))))
    except Exception as e:   # noqa: BLE001
        q.put(("error", repr(e)))


@pytest.mark.gpu
def test_block_local_start_through_rccl_one_rank(td):
    """solve_sharded's block-local sequence on an RCCL process group (one rank, TD_SHARD_FORCE_AR=1: the all-gather of the
    state segments and the scalar all-reduce are issued by RCCL on the stream the library shares with torch): optimal,
    repeatable, and td_assign's row_to_col when phase A leaves nothing"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_blocks_rccl_worker, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", out
    path, left, totals, duals, same, same_ref, ref_total = out
    assert path == "blocks"
    assert totals == duals == [10 * 12288] * 2 and ref_total == 10 * 12288
    assert same
    if left == 0:
        assert same_ref


@pytest.mark.gpu
def test_bench_two_gpus_end_to_end(td):
    """the first box that shows more than one GPU runs `bench.py --gpus 2` end to end — two ranks, RCCL with
    world 2 (the all-gather of the block-local start, the scalar all-reduces, the MAX all-reduces of the bid keys should rows
    be left), hipIpc peer mappings between two different devices if the finisher is needed.  Skipped on one-GPU boxes."""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--sharded-n", "16384", "--n", "8192"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    sh = line["sharded_single_instance"]
    assert "error" not in sh, sh
    assert sh["optimal"] and sh["total_cost"] == 10 * 16384 and sh["speedup_vs_single_gpu"] > 0
    assert sh["sequence"] in ("blocks", "auction")
    # the plain sequence too (a MAX all-reduce of the keys per round, the finisher on rank 0 over hipIpc-mapped shards)
    env["TD_SHARD_BLOCKS"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--sharded-n", "16384", "--n", "8192"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    sh = json.loads(r.stdout.strip().splitlines()[-1])["sharded_single_instance"]
    assert "error" not in sh and sh["optimal"] and sh["sequence"] == "auction", sh


def _line_instance(n, seed, spread=10):
    rng = np.random.default_rng(seed)
    a, b = rng.integers(0, spread * n, n), rng.integers(0, spread * n, n)
    return np.abs(a[:, None] - b[None, :]).astype(np.int32), a, b


@pytest.mark.gpu
@pytest.mark.parametrize("n,world,spread", [(2048, 4, 10), (1001, 3, 10), (513, 8, 1), (4096, 1, 1000), (96, 5, 10)])
def test_line_sharded_in_process_equals_sorted_matching(td, n, world, spread):
    """the line-metric path over row shards (td_line_shard_*), `world` shards driven in one process:
    the total is the sorted matching's (== the oracle's optimum), every shard certifies its rows, the local
    row_to_col pieces form a permutation; ragged shards, n % 4 != 0 (scalar certificate loads), duplicate positions."""
    import torch
    from oracle import oracle
    from taxidispatcher_amd import sharded
    cost, a, b = _line_instance(n, 21, spread)
    ref = int(np.abs(np.sort(a) - np.sort(b)).sum())     # an optimal matching of points on a line does not cross
    if n <= 1001:
        assert ref == oracle.assign(cost)[0]
    full = torch.from_numpy(cost).cuda()
    shards = []
    try:
        for r in range(world):
            row0, nrows, _ = sharded.shard_bounds(n, world, r)
            shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
        got = sharded.line_sharded(shards, None)
    finally:
        for s in shards:
            s.close()
    assert got is not None
    total, parts = got
    r2c = np.concatenate(parts)
    assert total == ref
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(cost[np.arange(n), r2c].sum()) == ref


@pytest.mark.gpu
def test_line_sharded_refuses_other_matrices(td):
    """a matrix that is no line metric (uniform, thresholded |a - b|, one perturbed cell) is refused by the shards'
    certificates, never answered wrongly"""
    import torch
    from oracle import oracle
    from taxidispatcher_amd import sharded
    n, world = 1024, 4
    cost, _, _ = _line_instance(n, 4)
    thr = cost.copy()
    thr[thr >= 900] = 250000
    bumped = cost.copy()
    bumped[n // 2 + 3, 5] = max(0, int(bumped[n // 2 + 3, 5]) - 400)    # one cheaper cell in another shard's row
    for m in (oracle.gen_uniform(n, 3, 10, 40), thr, bumped):
        full = torch.from_numpy(m).cuda()
        shards = []
        try:
            for r in range(world):
                row0, nrows, _ = sharded.shard_bounds(n, world, r)
                shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
            got = sharded.line_sharded(shards, None)
        finally:
            for s in shards:
                s.close()
        if got is not None:      # (a perturbation that leaves the sorted matching optimal may still certify)
            assert got[0] == oracle.assign(m)[0]
        else:
            assert m is not cost


def _line_rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import torch
        import torch.distributed as dist
        import taxidispatcher_amd as td
        from taxidispatcher_amd import sharded
        td.init(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        n = 3000
        cost, a, b = _line_instance(n, 8)
        sh = sharded.HipShard(n, 0, n, torch.from_numpy(cost).cuda())
        try:
            r2c, total, dual = sharded.solve_sharded(sh, dist, want_dual=True)
            path = sharded.solve_sharded.last_path
            os.environ["TD_LINE"] = "0"
            r2, t2, d2 = sharded.solve_sharded(sh, dist, want_dual=True)
            path2 = sharded.solve_sharded.last_path if hasattr(sharded.solve_sharded, "last_path") else None
        finally:
            sh.close()
        sharded.HipShard.destroy_comm()
        dist.destroy_process_group()
        ref = int(np.abs(np.sort(a) - np.sort(b)).sum())
        ok = total == dual == t2 == d2 == ref and sorted(np.asarray(r2c).tolist()) == list(range(n))
        q.put(("ok" if ok else "mismatch", (path, path2, int(total), int(t2), ref)))
    except Exception as e:   # noqa: BLE001
        q.put(("error", repr(e)))


@pytest.mark.gpu
def test_solve_sharded_takes_the_line_path_under_rccl(td):
    """solve_sharded on the nccl backend (one rank): |a - b| rows go through the line phases on the shard's shared
    stream (device segments all-reduced by RCCL's ordering), TD_LINE=0 forces the auction: same optimum"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_line_rccl_worker, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", out
    assert out[0] == "line"


def _drive_shards_in_process(torch, sharded, full, n, world, fused, blocks=False):
    """the steps of solve_sharded over `world` shards in one process (sharded.solve_shards_in_process)"""
    shards = []
    try:
        for r in range(world):
            row0, nrows, rps = sharded.shard_bounds(n, world, r)
            shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
        return sharded.solve_shards_in_process(shards, blocks=blocks, fused_round0=fused)   # fused False: round 0 as its own k_bid launch
    finally:
        for s in shards:
            s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["g1", "padded", "uniq"])
def test_fused_round0_in_shards_is_bit_identical(td, kind):
    """td_shard_options: a wide shard's 1-byte compress pass writes round 0's bids (k_compress_reg<.., BID0>) and defers
    its constant rows; the run must equal the run with round 0 as its own k_bid launch AND td_assign, bit for bit"""
    import torch
    from taxidispatcher_amd import sharded
    n, world = 12288, 3
    g = torch.Generator(device="cuda").manual_seed(11)
    if kind == "uniq":
        full = torch.randint(3, 200, (n, n), dtype=torch.int32, device="cuda", generator=g)
        full[torch.arange(n, device="cuda"), torch.randint(0, n, (n,), device="cuda", generator=g)] = 0
    else:
        full = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
        if kind == "padded":
            full[torch.randperm(n, device="cuda", generator=g)[: n // 4]] = 250
    a = _drive_shards_in_process(torch, sharded, full, n, world, True)
    b = _drive_shards_in_process(torch, sharded, full, n, world, False)
    assert a[1] == a[2] == b[1] == b[2]
    assert np.array_equal(a[0], b[0])
    assert a[3]["path"] == b[3]["path"] == "auction"     # three ranks do not own whole diagonal blocks
    was = _ffi_lib().td_set_blocks(0)                       # td_assign without the block-local start: the same sequence
    try:
        ref, ref_total, ref_dual = td.assign(full, want_dual=True)
    finally:
        _ffi_lib().td_set_blocks(was)
    assert ref_total == a[1] == ref_dual
    if kind != "padded":    # (td_assign may solve a padded model transposed: another optimum among the ties)
        assert np.array_equal(ref, a[0])
    assert sorted(a[0].tolist()) == list(range(n))


def _ffi_lib():
    from taxidispatcher_amd import _ffi
    return _ffi.lib()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("g1", 8), ("g1", 2), ("padded", 4), ("uniq", 8), ("wide", 2), ("ties", 1)])
def test_block_local_start_in_shards(td, kind, world):
    """csrc/td_blocks.h through the shard API (n = 12 288: the smallest size whose compress pass writes the zero-slice
    bids): phase A on every shard's own diagonal blocks, ONE exchange, the ordinary rounds and the finisher for what is
    left.  Optimal (total == dual bound), a permutation, and — when phase A left nothing, which is when both sequences
    coincide — td_assign's row_to_col with the same 8 blocks, bit for bit.  "wide" does not fit one byte: every shard
    learns it from the exchanged segments and the plain sequence runs."""
    import torch
    from taxidispatcher_amd import sharded
    n = 12288
    g = torch.Generator(device="cuda").manual_seed(17)
    if kind == "uniq":      # one cell at the minimum of every row: phase A places the rows whose minimum lies in their own block
        full = torch.randint(3, 200, (n, n), dtype=torch.int32, device="cuda", generator=g)
        full[torch.arange(n, device="cuda"), torch.randint(0, n, (n,), device="cuda", generator=g)] = 0
    elif kind == "wide":
        full = torch.randint(0, 1000, (n, n), dtype=torch.int32, device="cuda", generator=g)
    elif kind == "ties":
        full = torch.randint(0, 3, (n, n), dtype=torch.int32, device="cuda", generator=g)
    else:
        full = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
        if kind == "padded":
            full[torch.randperm(n, device="cuda", generator=g)[: n // 4]] = 250
    r2c, tot, dual, info = _drive_shards_in_process(torch, sharded, full, n, world, True, blocks=True)
    assert info["path"] == ("auction" if kind == "wide" else "blocks")
    assert tot == dual
    assert sorted(r2c.tolist()) == list(range(n))
    assert int(full[torch.arange(n, device="cuda"), torch.from_numpy(r2c).cuda().long()].long().sum().item()) == tot
    was = _ffi_lib().td_set_blocks(8)
    try:
        ref, ref_total, ref_dual = td.assign(full, want_dual=True)
    finally:
        _ffi_lib().td_set_blocks(was)
    assert ref_total == tot == ref_dual
    if kind == "g1":
        assert tot == 10 * n
    if info["left"] == 0 and kind not in ("padded", "wide"):
        assert np.array_equal(ref, r2c)


def _fused_worker(rank, world, port, n, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import taxidispatcher_amd as td
    from taxidispatcher_amd import _ffi, sharded
    td.init(0)
    row0, nrows, rps = sharded.shard_bounds(n, world, rank)
    rows = torch.empty((nrows, n), dtype=torch.int32, device="cuda")
    _ffi.check(_ffi.lib().td_gen_uniform(n, seed, 10, 40, row0, nrows, rows.data_ptr()))
    sh = sharded.HipShard(n, row0, nrows, rows)
    try:
        r2c, total, dual = sharded.solve_sharded(sh, dist, want_dual=True)
        path = sharded.solve_sharded.last_path
    finally:
        sh.close()
    q.put((rank, np.asarray(r2c).tolist(), int(total), int(dual), path, sharded.solve_sharded.last_left))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_wide_shards_fused_round0(td):
    """two processes, shards wide enough for round 0 out of the compress pass (n = 12 288: k_compress_reg<.., BID0> through
    solve_sharded's own sequence, the refused line attempt in front of it): the result is td_assign's, bit for bit"""
    import torch
    from taxidispatcher_amd import _ffi
    n, seed = 12288, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, 2, port, n, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = _collect(q, procs, 2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(_ffi.lib().td_gen_uniform(n, seed, 10, 40, 0, n, full.data_ptr()))
    ref, ref_total, ref_dual = td.assign(full, want_dual=True)
    assert outs[0][2] == outs[1][2] == ref_total == ref_dual == outs[0][3] == outs[1][3]
    assert outs[0][4] == outs[1][4] == "blocks"     # two ranks own four diagonal blocks each: the block-local start
    got = np.array(outs[0][1] + outs[1][1])
    assert sorted(got.tolist()) == list(range(n))
    if outs[0][5] == 0:   # phase A left nothing: the same sequence as td_assign's (8 blocks by default at this size)
        assert np.array_equal(got, ref)
