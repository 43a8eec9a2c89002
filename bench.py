#!/usr/bin/env python3
"""bench.py — headline benchmark of the assignment hot path on MI355X.

metric   : N x N assignments/s = N / wall time of one full solve (BASELINE.json / SURVEY 8d)
workload : g1  perf.jl:5 instance family, cost ~ U{10..40}, N = 16384 (BASELINE configs[2], the
               configuration the north-star target is quoted on; fits one GPU).  A "step" is one
               full pass of the hot path over one synthetic instance that is generated ON the
               device: the 4*N^2-byte int32 cost write (synthetic stand-in for the cost-matrix
               build: perf.jl draws the matrix directly) + td_assign (compress, auction bidding
               rounds, augmenting-path finisher, total).  The optimum is known (10*N) and is
               checked every step.
           g2  greedy_opt.py geometry: positions -> td_cost_build (|a-b|) -> td_assign.
           g3  Simulator.java tick shape: S=50, DROP_TIME threshold, dummy columns.
           tick BASELINE configs[4]: one Simulator.java tick (n = 1300 cabs x 900 requests):
               td_cost_build -> td_lcm down to 600 -> shrink -> td_cost_build -> td_assign;
               value = cabs dispatched per second (n / step time), --n is ignored.
--gpus N : one process per GPU (torch.distributed, RCCL).  Started without WORLD_SIZE, bench.py
           launches the N ranks itself (torch.distributed.run as a child process, before anything
           touches the GPU).  For N > 1 the headline is BASELINE configs[3]: ONE 65 536 x 65 536
           instance, rows sharded over the N GPUs, every rank builds its row block in place and
           starts on its own diagonal blocks (csrc/td_blocks.h), ONE RCCL all-gather, then one
           RCCL MAX all-reduce of the packed bid keys per bidding round for the rows still free
           ("scaling": "strong"; the same instance solved by td_assign on one GPU is timed in the
           same run).  The replicas
           figure (an independent N = 16 384 instance per GPU, no collective, weak scaling) is a
           side field; --multi-mode replicas makes it the headline instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--workload", default="g1", choices=["g1", "g2", "g3", "tick", "pool"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded leg even with one rank (exercises RCCL + the shard API on 1 GPU)")
    ap.add_argument("--sharded-n", type=int, default=65536,
                    help="with --gpus > 1: also solve ONE instance of this size row-sharded over all ranks "
                         "(BASELINE configs[3]); 0 disables")
    ap.add_argument("--shard-builder", default="gen", choices=["gen", "cost", "both"],
                    help="sharded leg: how a rank fills its rows — td_gen_uniform's row window, td_cost_build_rows, or both (two timings)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--multi-mode", default="sharded", choices=["sharded", "replicas"],
                    help="headline of a multi-GPU run: the row-sharded single instance (configs[3]) or replicas")
    ap.add_argument("--line-metric", default="on", choices=["on", "off"],
                    help="off: td_set_line_metric(0), the general solver alone (matters for g2, whose |a-b| matrix the "
                         "default path answers with the sorted matching + certificate pass)")
    ap.add_argument("--no-extras", action="store_true", help="skip the g2 / g3 / tick side measurements (1 GPU)")
    return ap.parse_args()


class TickWorkload:
    """BASELINE configs[4]: the per-tick re-solve of Simulator.java:163-208 with LCM pre-reduce.
    One C-ABI call per tick (td_tick): both cost matrices are library buffers in HBM, the shrink runs on the
    device; what crosses PCIe per tick is the position arrays (<= 5 KiB), the LCM pair list, the kept indices
    and row_to_col."""

    def __init__(self, seed, td):
        import torch
        self.td, self.torch = td, torch
        rng = np.random.default_rng(seed)
        self.n = 1300
        self.cab_to = rng.integers(0, 50, 1300).astype(np.int32)
        self.dem_from = rng.integers(0, 50, 900).astype(np.int32)
        self.expected = None
        self.kind = "tick"

    def step(self):
        # ONE C-ABI call: cost build -> LCM -> shrink on the device -> cost build -> assign (td_tick)
        t = self.td.tick(self.cab_to, self.dem_from, None, big_cost=250000, drop_time=10, max_non_lcm=600)
        rows, n2, total = t["lcm_rows"], t["n_rest"], t["total"]
        self.last = (len(rows), n2, total)
        return total


class Workload:
    """Device-resident synthetic instance + the step that solves it through the C ABI."""

    def __init__(self, kind, n, seed, torch, td, ffi):
        self.kind, self.n, self.seed = kind, n, seed
        self.td, self.ffi, self.lib = td, ffi, ffi.lib()
        self.cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
        self.r2c = torch.empty(n, dtype=torch.int32, device="cuda")
        self.total = ctypes.c_int64(0)
        rng = np.random.default_rng(seed)
        if kind == "g2":
            S = 10 * n
            self.cab_to = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
            self.dem_from = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
            self.expected = int(np.abs(np.sort(self.cab_to.cpu().numpy().astype(np.int64)) -
                                       np.sort(self.dem_from.cpu().numpy().astype(np.int64))).sum())
            self.args = (-1, 250000)
        elif kind == "g2u":   # greedy_opt.py's rand_list drops from == to draws: two cabs short, big_cost rows
            S = 10 * n
            self.cab_to = torch.from_numpy(rng.integers(0, S, n - 2).astype(np.int32)).cuda()
            self.dem_from = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
            self.expected = None
            self.args = (-1, 250000)
        elif kind in ("g3", "g3f"):   # g3f: the same model through td_build_assign (no int32 matrix on the device)
            S = 50
            nd = max(1, int(0.363 * n))
            self.cab_to = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
            self.dem_from = torch.from_numpy(rng.integers(0, S, nd).astype(np.int32)).cuda()
            self.expected = None
            self.args = (10, 250000)
        elif kind == "wide":   # uniform 0..10^6: tie-free 4-byte rows (VERDICT r2 item 1's second family); the matrix is made once
            self.cost.copy_(torch.from_numpy(rng.integers(0, 10**6, (n, n)).astype(np.int32)))
            self.expected = None
        elif kind == "geo2":   # 2-D city grid, Manhattan distance (not in the reference, whose table is a line)
            ax, ay, bx, by = (torch.from_numpy(rng.integers(0, 4000, n).astype(np.int32)).cuda() for _ in range(4))
            self.cost.copy_((ax[:, None] - bx[None, :]).abs() + (ay[:, None] - by[None, :]).abs())
            self.expected = None
        else:
            self.expected = 10 * n if n >= 1000 else None

    def build(self):
        if self.kind == "g1":
            self.ffi.check(self.lib.td_gen_uniform(self.n, self.seed, 10, 40, 0, self.n, self.cost.data_ptr()))
        elif self.kind in ("wide", "geo2", "g3f"):
            pass   # resident matrix: the step is the solve alone
        else:
            thr, fill = self.args
            self.ffi.check(self.lib.td_cost_build(self.cab_to.data_ptr(), None, int(self.cab_to.numel()),
                                                  self.dem_from.data_ptr(), None, int(self.dem_from.numel()),
                                                  None, 0, fill, thr, 0, self.cost.data_ptr()))

    def solve(self):
        if self.kind == "g3f":
            thr, fill = self.args
            self.ffi.check(self.lib.td_build_assign(self.cab_to.data_ptr(), int(self.cab_to.numel()), self.dem_from.data_ptr(),
                                                    int(self.dem_from.numel()), None, 0, fill, thr, self.r2c.data_ptr(),
                                                    ctypes.byref(self.total), None))
            return self.total.value
        self.ffi.check(self.lib.td_assign(self.n, self.cost.data_ptr(), self.r2c.data_ptr(),
                                          ctypes.byref(self.total), None))
        return self.total.value

    def step(self):
        self.build()
        t = self.solve()
        if self.expected is not None and t != self.expected:
            raise RuntimeError("wrong optimum: got %d expected %d" % (t, self.expected))
        return t


def kernel_profile(wl, ffi, reps):
    """Per-kernel-class device time with HIP events recorded on the library's stream."""
    lib = ffi.lib()
    ffi.check(lib.td_profile_enable(1))
    ffi.check(lib.td_profile_reset())
    for _ in range(reps):
        wl.step()
    out = {}
    for name, k in ffi.TD_K.items():
        ms = ctypes.c_double(0)
        cnt = ctypes.c_int64(0)
        ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            out[name] = {"total_ms": ms.value / reps, "launches": cnt.value // reps,
                         "avg_us": 1e3 * ms.value / cnt.value}
    ffi.check(lib.td_profile_enable(0))
    return out


def sharded_leg(n, world, rank, torch, dist, ffi, steps, warmup, builder="gen"):
    """BASELINE configs[3]: ONE n x n perf.jl instance row-sharded over all ranks.  A step = every
    rank writes its row block in place (td_gen_uniform with its row window: the synthetic stand-in
    for the shard-local cost build, td_cost_build_rows) + the sharded solve: one RCCL MAX all-reduce
    of the packed bid keys per bidding round (taxidispatcher_amd/sharded.py), finisher on rank 0
    over hipIpc-mapped shards.  W untimed steps, then K steps between barrier + synchronize."""
    from taxidispatcher_amd import sharded
    row0, nrows, _ = sharded.shard_bounds(n, world, rank)
    rows = torch.empty((max(nrows, 1), n), dtype=torch.int32, device="cuda")
    sh = sharded.HipShard(n, row0, nrows, rows)   # workspace allocated once, reused by every solve
    total = None

    if builder == "cost":
        # the shard-local COST BUILD itself (td_cost_build_rows, greedy_opt.py:86-99 over a general S x S table with
        # perf.jl's value range 10..40): positions and table are replicated, every rank builds its own rows
        rng = np.random.default_rng(5)
        S = 1000
        table = torch.from_numpy(rng.integers(10, 41, (S, S)).astype(np.int32)).cuda()
        cab = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
        dem = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()

    seq = {}

    def step():
        if nrows and builder == "cost":
            ffi.check(ffi.lib().td_cost_build_rows(cab.data_ptr(), None, n, dem.data_ptr(), None, n, table.data_ptr(), S, 250000, -1, 0,
                                                   row0, nrows, rows.data_ptr()))
        elif nrows:
            ffi.check(ffi.lib().td_gen_uniform(n, 7, 10, 40, row0, nrows, rows.data_ptr()))
        tot = sharded.solve_sharded(sh, dist)[1]
        seq["path"], seq["left"] = sharded.solve_sharded.last_path, getattr(sharded.solve_sharded, "last_left", None)
        return tot

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(warmup):
            total = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            total = step()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finally:
        sh.close()
    del rows
    torch.cuda.empty_cache()
    how = ("block-local start on every rank's diagonal blocks (csrc/td_blocks.h), ONE all-gather of %d KiB, %s"
           % ((16 + 2 * nrows) * 4 * world // 1024, "nothing left for the rounds" if seq.get("left") == 0 else
              "then RCCL MAX all-reduces of %d KiB keys per bidding round + finisher on rank 0 for the %s rows left" % (n * 8 // 1024, seq.get("left")))
           if seq.get("path") == "blocks" else
           "RCCL MAX all-reduce of %d KiB keys per bidding round, finisher on rank 0 over hipIpc-mapped shards" % (n * 8 // 1024))
    return {"workload": "g1 N=%d, ONE instance row-sharded over %d GPUs (%d rows each), shard-local %s + %s"
                        % (n, world, nrows, "cost build (td_cost_build_rows, 1000-stand table)" if builder == "cost" else "cost write", how),
            "sequence": seq.get("path"), "rows_left_after_phase_a": seq.get("left"),
            "n": n, "ms_per_step": 1e3 * dt / steps, "assignments_per_s": n * steps / dt,
            "total_cost": int(total), "optimal": bool(total == 10 * n), "scaling": "strong", "seconds": dt}


def single_gpu_reference(n, torch, ffi, reps=3):
    """the same n x n instance through td_assign on ONE GPU (rank 0), for the strong-scaling factor"""
    lib = ffi.lib()
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    r2c = torch.empty(n, dtype=torch.int32, device="cuda")
    tot = ctypes.c_int64(0)
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ffi.check(lib.td_gen_uniform(n, 7, 10, 40, 0, n, cost.data_ptr()))
        ffi.check(lib.td_assign(n, cost.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), None))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    del cost
    torch.cuda.empty_cache()
    return {"ms_per_step": 1e3 * min(ts[1:]), "total_cost": int(tot.value)}


# name, seed, n, max_wait, loss choices (the generator of tests/golden/make_pool_fixtures.py, gendemand.py:10 trips), pool sizes.
# pool_n.c keeps at most MAX_ARR = 10 000 happy plans per child in a static array WITHOUT a bounds check: n = 600, k = 3 has
# 17 154 for one child (k = 4: 310 944) and the reference binary dies with SIGSEGV on it.  So the reference binary is timed on
# the committed fixture instances (tests/golden/pool_n/<name>_demand.csv, the largest it can run), the bench sizes run
# on the GPU and on the oracle's port only.
POOL_CASES = [
    ("n600", 11, 600, 2, [1, 10, 25], (2, 3, 4)),
    ("n2000", 12, 2000, 1, [1, 5], (2, 3)),
]
POOL_FIXTURE_CASES = [("c300", (2, 3)), ("d200", (4,)), ("f400", (3,)), ("b120", (2, 3, 4))]


def pool_demand(seed, n, max_wait, losses):
    rng = np.random.default_rng(seed)
    frm = rng.integers(0, 50, n)
    diff = rng.integers(1, 9, n) * rng.choice([-1, 1], n)
    to = np.clip(frm + diff, 0, 49)
    to = np.where(to == frm, np.where(frm > 0, frm - 1, 1), to)
    wait = rng.integers(0, max_wait + 1, n)
    loss = rng.choice(losses, n)
    return np.stack([np.arange(n), frm, to, wait, loss], 1)


def pool_workload(td, reps, cpu_seconds):
    """f-4 (pool_n.c:101-242 / findpool.c:122-176): pools of k = 2, 3, 4 passengers over findpool.c's 8 first-pick-up
    slices (8 td_pool_n calls) + the merge (td_pool_merge), timed per (n, k); beside it the oracle's single-thread C
    restatement of the same 8 children + merge on this box's host core (bounded), and — where the file exists — the
    wall time of the REFERENCE binary oracle/_ref/pool_n measured in the build container
    (tests/golden/pool_n/reference_timing.json, made by tests/golden/time_pool_reference.py)."""
    import torch
    from oracle import oracle
    ref_path = os.path.join(ROOT, "tests", "golden", "pool_n", "reference_timing.json")
    ref = json.load(open(ref_path)) if os.path.exists(ref_path) else {}
    out = {}
    cases = [(name, pool_demand(seed, n, mw, losses), ks) for name, seed, n, mw, losses, ks in POOL_CASES]
    for name, ks in POOL_FIXTURE_CASES:
        fp = os.path.join(ROOT, "tests", "golden", "pool_n", name + "_demand.csv")
        if os.path.exists(fp):
            cases.append((name, np.loadtxt(fp, delimiter=",", dtype=np.int64), ks))
    for name, d, ks in cases:
        n = int(d.shape[0])
        frm, to, wait, loss = (np.ascontiguousarray(d[:, c]).astype(np.int32) for c in (1, 2, 3, 4))
        for k in ks:
            def gpu_once():
                lists, happy = [], 0
                for child in range(8):
                    lst, nh = td.find_pool_n(k, d, child=child, children=8)
                    lists.append(lst)
                    happy += nh
                return td.merge_pools(k, n, lists), happy
            merged, happy = gpu_once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                merged, happy = gpu_once()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / reps
            # CPU port: the same 8 slices + the same merge rule, one thread, bounded
            t_budget = time.perf_counter() + cpu_seconds
            creps, spent, cm = 0, 0.0, None
            step = n // 8 + 1
            while True:
                t1 = time.perf_counter()
                lists = []
                for child in range(8):
                    a = step * child
                    lst, _ = oracle.pool_n(k, frm, to, wait, loss, None, a, max(a, min(n, a + step)), cap=4000000)
                    lists.append(lst.tolist())
                allp = [r for lst in lists for r in lst]
                if k == 4:
                    allp = sorted(allp, key=lambda r: r[8])
                used, kept = set(), []
                for r in allp:
                    if any(x in used for x in r[:k]):
                        continue
                    used.update(r[:k])
                    kept.append(r)
                cm = kept
                spent += time.perf_counter() - t1
                creps += 1
                if creps >= 8 or time.perf_counter() + spent / creps > t_budget:
                    break
            key = "%s_k%d" % (name, k)
            out[key] = {"n": n, "k": k, "happy_plans": int(happy), "pools": int(len(merged)), "gpu_ms": ms,
                        "requests_per_s": n / (ms * 1e-3), "cpu_port_ms": 1e3 * spent / creps, "cpu_port_reps": creps,
                        "pools_match_cpu_port": bool(np.array_equal(np.asarray(merged), np.asarray(cm, np.int32).reshape(-1, 2 * k + 1)))}
            if key in ref:
                out[key]["reference_binary_ms_build_container"] = ref[key]["wall_ms"]
                out[key]["reference_binary_plans"] = ref[key].get("plans")
    return out


def cpu_baseline_tick(seconds):
    from oracle import oracle
    rng = np.random.default_rng(1)
    cab_to = rng.integers(0, 50, 1300)
    dem_from = rng.integers(0, 50, 900)
    reps, spent = 0, 0.0
    t_budget = time.perf_counter() + seconds
    while True:
        t0 = time.perf_counter()
        _, cost = oracle.cost_build(cab_to, dem_from, None, 250000, 10)
        _, rows, cols, _ = oracle.lcm(cost, mask=250000, stop_value_on=1, stop_value=250000, stop_size=600,
                                      sum_below=250000, java_scan=1)
        keep_c = np.setdiff1d(np.arange(1300), rows)
        keep_d = np.setdiff1d(np.arange(900), cols)
        _, cost2 = oracle.cost_build(cab_to[keep_c], dem_from[keep_d], None, 250000, 10)
        tot = oracle.assign(cost2)[0]
        spent += time.perf_counter() - t0
        reps += 1
        if time.perf_counter() + spent / reps > t_budget or reps >= 64:
            break
    return {"value": 1300 * reps / spent, "unit": "assignments/s", "cores": 1, "kind": "port",
            "sample": "stand-in, not GLPK: %d x (tick: cost build 1300x900 + Java-variant LCM (k full n^2 scans) + cost build + exact "
                      "solve n=600, oracle/td_oracle.c, one thread), %.1f s" % (reps, spent),
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "last_total": int(tot)}


def cpu_baseline(kind, n_gpu, seconds):
    """The oracle's exact solver (a single-thread C port of the path: cost build / generate +
    exact assignment) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle
    if kind == "tick":
        return cpu_baseline_tick(seconds)
    t_budget = time.perf_counter() + seconds
    n = n_gpu
    if kind != "g1":
        n = min(n_gpu, 2048)
    else:
        n = min(n_gpu, 16384)
    rng = np.random.default_rng(1)
    reps, spent, tot = 0, 0.0, None
    while True:
        t0 = time.perf_counter()
        if kind == "g1":
            c = oracle.gen_uniform(n, 1 + reps, 10, 40)
        elif kind in ("g2", "g2u"):
            _, c = oracle.cost_build(rng.integers(0, 10 * n, n - (2 if kind == "g2u" else 0)), rng.integers(0, 10 * n, n), None, 250000, -1)
        elif kind == "wide":
            c = rng.integers(0, 10**6, (n, n)).astype(np.int32)
            t0 = time.perf_counter()   # (resident matrix on the GPU side too: the solve alone is timed)
        elif kind == "geo2":
            ax, ay, bx, by = (rng.integers(0, 4000, n) for _ in range(4))
            c = (np.abs(ax[:, None] - bx[None, :]) + np.abs(ay[:, None] - by[None, :])).astype(np.int32)
            t0 = time.perf_counter()
        else:
            _, c = oracle.cost_build(rng.integers(0, 50, n), rng.integers(0, 50, max(1, int(0.363 * n))), None,
                                     250000, 10)
        tot = oracle.assign(c)[0]
        spent += time.perf_counter() - t0
        reps += 1
        if time.perf_counter() + spent / reps > t_budget or reps >= 64:
            break
    return {"value": n * reps / spent, "unit": "assignments/s", "cores": 1, "kind": "port",
            "sample": "stand-in, not GLPK: %d x (%s instance N=%d: build + exact shortest-augmenting-path solve, "
                      "oracle/td_oracle.c, gcc -O3, one thread), %.1f s" % (reps, kind, n, spent),
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "last_total": int(tot)}


def glpk_baseline(sizes=(20, 100, 400)):
    """the reference's own Python / GLPK path on this box's host — iff `cvxopt` imports here.  The build's
    restatement of solver.py:13-26 (n x n objective as a cvxopt matrix, dense 2n x n^2 equality matrix A, one dummy
    inequality, every variable binary, cvxopt.glpk.ilp) on perf.jl-style instances, one thread; N <= 400 because A is
    O(N^3) bytes (1 GB at 400).  Returns a dict per N, or a string saying why it was not run."""
    try:
        from cvxopt import matrix
        from cvxopt.glpk import ilp
    except Exception as e:   # noqa: BLE001  (not installed in this image: no network, no wheel)
        return "cvxopt not importable on this box (%s: %s); the stand-in C port is the CPU leg" % (type(e).__name__, str(e)[:80])
    out = {}
    rng = np.random.default_rng(1)
    for n in sizes:
        cost = rng.integers(10, 41, (n, n)).tolist()
        t0 = time.perf_counter()
        c = matrix(cost, tc="d")                     # solver.py:13 (inner lists are columns: index n*cab + cust)
        a = np.zeros((2 * n, n * n))                 # solver.py:15-19
        for i in range(n):
            for j in range(n):
                a[i][n * i + j] = 1.0
                a[n + i][n * j + i] = 1.0
        A = matrix(a)                                # solver.py:20
        g = matrix([[0.0] * (n * n)])                # solver.py:21-23: one dummy inequality 0 x <= 0
        h = matrix([0.0])
        b = matrix([1.0] * (2 * n))
        idx = set(range(n * n))                      # solver.py:24-25: all binary
        status, x = ilp(matrix(list(c), (n * n, 1)), g.T, h, A, b, idx, idx)   # solver.py:26
        dt = time.perf_counter() - t0
        tot = None if x is None else int(round(sum(cost[k // n][k % n] * x[k] for k in range(n * n))))
        out["n%d" % n] = {"seconds": dt, "assignments_per_s": n / dt, "status": status, "total": tot}
    return out


TRAFFIC_PROFILE = os.path.join("profiles", "r4", "rocprof_summary_r4_g1.json")   # the committed PMC summary the traffic figure is read from
LIB_PATH = os.path.join(ROOT, "taxidispatcher_amd", "libtaxidispatcher_amd.so")


def sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def traffic_binary():
    """sha-16 of the library the committed PMC summary was taken with (written by tools/summarize_profile.py)"""
    try:
        return json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE))).get("library_sha16")
    except Exception:   # noqa: BLE001
        return None



def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary named in
    TRAFFIC_PROFILE (made by tools/profile_round.sh: separate FETCH_SIZE and WRITE_SIZE passes,
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  Not measured in this run: the line carries the
    file name as `traffic_source`.  None if the summary is missing."""
    prefix = {"compress": "k_compress", "gen": "k_gen_uniform", "cost_build": "k_cost_build", "bid": "k_bid",
              "sap": "k_sap", "assign": "k_assign", "final": "k_final", "lcm": "k_lcm", "cert": "k_line_cert"}.get(kernel_class)
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    if not os.path.exists(path) or not prefix:
        return None
    try:
        t = json.load(open(path)).get("traffic_per_dispatch", {})
        best = None
        for k, v in t.items():
            if prefix in k:
                best = max(best or 0.0, float(v["hbm_bytes_corrected"]))
        return best
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process
    (which has not touched the GPU and never will) and relay rank 0's JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1 or args.force_sharded:
        import datetime
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                timeout=datetime.timedelta(seconds=240))
    import taxidispatcher_amd as td
    from taxidispatcher_amd import _ffi as ffi
    td.init(local)
    td.set_line_metric(args.line_metric == "on")
    # the library runs on its own stream; the timed region is bracketed by device-wide synchronisation

    if args.workload == "pool":
        res = pool_workload(td, max(1, min(args.steps, 20)), 0.0 if args.no_cpu_baseline else min(args.cpu_seconds, 4.0))
        head = res["n600_k4"]
        line = {"metric": "pool requests/sec (pools of 4, 8 first-pick-up slices + merge)", "value": head["requests_per_s"],
                "unit": "requests/s", "n_gpus": 1, "steps": max(1, min(args.steps, 20)), "warmup": 1, "ms_per_step": head["gpu_ms"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 / f64 happiness test", "data": "synthetic",
                "config": {"workload": "pool_n.c:101-242 fan-out of findpool.c:122-176: n = 600 requests, pools of 4 (all sizes in `pool`)"},
                "pool": res,
                "cpu_baseline": {"value": 600.0 / (head["cpu_port_ms"] * 1e-3), "unit": "requests/s", "cores": 1, "kind": "port",
                                 "sample": "oracle_pool_n (oracle/td_oracle.c) over the same 8 slices + the merge rule, n = 600, k = 4, %d repetitions"
                                           % head["cpu_port_reps"], "cpu_model": cpu_model()}}
        print(json.dumps(line), flush=True)
        return
    if args.workload == "tick":
        wl = TickWorkload(1 + rank, td)
        args.n = wl.n
    else:
        wl = Workload(args.workload, args.n, 1 + rank, torch, td, ffi)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = wl.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = td.last_stats()

    prof = kernel_profile(wl, ffi, reps=3) if rank == 0 else {}

    def build_line(shard_res):
        n = args.n
        ms_per_step = 1e3 * dt / args.steps
        value = world * n * args.steps / dt
        # roofline: the dominant KERNEL of a step by device time.  Classes that are one launch of one
        # streaming kernel (cost write, compress, LCM) are compared directly; "bid" / "assign" / "sap"
        # are many launches of several different kernels (the largest single one, bidding round 0, is
        # ~1/3 of compress), so they are reported as a class but never priced as one kernel.
        alg_bytes = {"gen": 4.0 * n * n, "cost_build": 4.0 * n * n, "compress": 4.0 * n * n, "lcm": 4.0 * n * n,
                     "cert": 4.0 * n * n}   # cert = the certificate pass of the line-metric path (td_line.hip): one read of the matrix
        streaming = [k for k in prof if k in alg_bytes and prof[k]["launches"] <= 2]
        dom = max(streaming, key=lambda k: prof[k]["total_ms"]) if streaming else None
        roof = None
        if dom:
            p = prof[dom]
            b = alg_bytes[dom]
            achieved = b / (p["avg_us"] * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom), "traffic_source": TRAFFIC_PROFILE,
                    # the traffic figure comes from a committed rocprofv3 PMC pass, not from this run: the summary names the
                    # library it was taken with, this run names the one it loaded
                    "traffic_binary_sha16": traffic_binary(), "binary_sha16": sha16(LIB_PATH),
                    "traffic_binary_matches": traffic_binary() is not None and traffic_binary() == sha16(LIB_PATH),
                    "algorithmic_bytes_per_launch": b, "avg_launch_us": p["avg_us"], "launches_per_step": p["launches"],
                    "largest_class_by_time": max(prof, key=lambda k: prof[k]["total_ms"])}
            if args.workload == "tick":
                # a 1300 x 1300 model is ~45 launches of a few microseconds each: launch / memory LATENCY bounds the step,
                # an HBM fraction of it says nothing
                roof.update({"bound": "latency", "note": "latency-bound: ~25 dependent launches of a few microseconds on a 1300 x 900 model whose "
                             "matrices are never built (the LCM works on the stand positions, the remainder's cells are made inside the compress "
                             "pass); `achieved` / `frac` divide the 4 n^2 bytes a matrix WOULD have by the LCM class time, for the contract's sake only"})
        line = {
            "metric": "NxN assignments/sec", "value": value, "unit": "assignments/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s N=%d: device cost write (4N^2 B) + exact assignment, total checked == optimum"
                                   % (args.workload, n), "n": n, "instances_per_step": world,
                       "parallelism": "independent instance per GPU, no collective" if world > 1 else "single GPU"},
            "pairs_per_s": world * float(n) * n * args.steps / dt,
            "whole_step_algorithmic_GBps": 8.0 * n * n * world * args.steps / dt / 1e9,
            "total_cost": int(total), "solver_stats": stats, "kernels": prof, "roofline": roof,
        }
        if args.line_metric == "off":
            line["config"]["line_metric_attempt"] = "off"
        if shard_res is not None:
            line["sharded_single_instance"] = shard_res
            if world > 1 and args.multi_mode == "sharded" and "error" not in shard_res:
                # multi-GPU headline = BASELINE configs[3]; the replicas figure measured above moves to a side field
                line["replicas"] = {"value": value, "ms_per_step": ms_per_step, "scaling": "weak",
                                    "workload": line["config"]["workload"] + " (one independent instance per GPU, no collective)"}
                sn = shard_res["n"]
                line.update({"value": shard_res["assignments_per_s"], "ms_per_step": shard_res["ms_per_step"],
                             "scaling": "strong",
                             "config": {"workload": shard_res["workload"], "n": sn, "instances_per_step": 1,
                                        "parallelism": "rows sharded over %d GPUs; block-local start + one RCCL all-gather, RCCL MAX all-reduce per "
                                                       "bidding round for what is left" % world},
                             "pairs_per_s": float(sn) * sn * 1e3 / shard_res["ms_per_step"],
                             "whole_step_algorithmic_GBps": 8.0 * sn * sn / (shard_res["ms_per_step"] * 1e-3) / 1e9,
                             "total_cost": shard_res["total_cost"]})
        return line

    shard_res, single_ref = None, None
    watchdog = None
    if world > 1 or args.force_sharded:
        # a collective that never returns on some rank (RCCL / IPC trouble on a node this path has not seen) must
        # not take the whole run with it: after TD_BENCH_SHARD_TIMEOUT seconds every rank leaves, rank 0 having
        # printed the replicas line with the sharded leg marked as timed out
        import threading
        limit = float(os.environ.get("TD_BENCH_SHARD_TIMEOUT", "240"))

        def bail():
            if rank == 0:
                out = build_line({"error": "sharded leg did not finish within %.0f s (watchdog); replicas figure reported" % limit})
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
            os._exit(3)   # non-zero on every rank: a hung collective must not read as success (the JSON line above carries the error)

        watchdog = threading.Timer(limit, bail)
        watchdog.daemon = True
        watchdog.start()
        dist.barrier()
        if args.sharded_n > 0:
            del wl.cost
            torch.cuda.empty_cache()
            try:
                shard_res = sharded_leg(args.sharded_n, world, rank, torch, dist, ffi, max(1, args.steps), args.warmup,
                                        "cost" if args.shard_builder == "cost" else "gen")
                if args.shard_builder == "both":
                    extra = sharded_leg(args.sharded_n, world, rank, torch, dist, ffi, max(1, args.steps), args.warmup, "cost")
                    shard_res["with_td_cost_build_rows"] = {k: extra[k] for k in ("ms_per_step", "assignments_per_s", "total_cost", "workload")}
                if rank == 0:
                    single_ref = single_gpu_reference(args.sharded_n, torch, ffi)
                    shard_res["single_gpu_ms_per_step"] = single_ref["ms_per_step"]
                    shard_res["speedup_vs_single_gpu"] = single_ref["ms_per_step"] / shard_res["ms_per_step"]
            except Exception as e:  # keep the replicas line even if this leg fails
                shard_res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dist.barrier()
    if watchdog is not None:
        watchdog.cancel()
    if rank != 0:
        dist.destroy_process_group()
        return
    line = build_line(shard_res)
    n = args.n
    if world == 1 and not args.no_extras and args.workload == "g1":
        # the other workloads of SURVEY 8d / BASELINE configs, measured in the same run (side fields)
        extras = {}
        del wl.cost
        torch.cuda.empty_cache()
        # g2 twice: the default path (its |a-b| matrix is recognised as a line metric: sorted matching + certificate
        # pass, td_line.hip) and the general solver alone on the same instance (td_set_line_metric(0))
        for name, kind, en, reps in (("tick_1300x900", "tick", 0, 10), ("g3_n16384", "g3", 16384, 5),
                                     ("g3_n16384_td_build_assign_no_int32_matrix", "g3f", 16384, 5), ("g2_n16384", "g2", 16384, 10),
                                     ("g2_two_cabs_short_n16384", "g2u", 16384, 10), ("g2_n16384_general_solver", "g2", 16384, 2),
                                     ("uniform_0_1e6_n16384_solve_only", "wide", 16384, 3), ("manhattan_2d_n16384_solve_only", "geo2", 16384, 2)):
            try:
                td.set_line_metric(not name.endswith("general_solver"))
                w2 = TickWorkload(1, td) if kind == "tick" else Workload(kind, en, 1, torch, td, ffi)
                w2.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    w2.step()
                torch.cuda.synchronize()
                e_ms = 1e3 * (time.perf_counter() - t1) / reps
                extras[name] = {"ms_per_step": e_ms, "assignments_per_s": w2.n / (e_ms * 1e-3), "solver_stats": td.last_stats()}
                del w2
                torch.cuda.empty_cache()
                if kind == "g3f":
                    extras[name]["bytes_model"] = ("the fused pass writes N^2 bytes of 1-byte cells (0.27 GB) and reads position arrays only; the "
                                                   "8 N^2-byte figure of the int32 boundary does not apply to this entry")
                if not args.no_cpu_baseline and not name.endswith("general_solver") and kind != "g3f":
                    # the stand-in C port on a bounded sample of the same family (N <= 2048, ~2 s), same run, one host core
                    extras[name]["cpu_baseline"] = cpu_baseline(kind, en, min(2.0, args.cpu_seconds))
            except Exception as e:
                extras[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        td.set_line_metric(True)
        try:   # f-4: pools of 2 - 4 passengers, fan-out + merge (details: --workload pool)
            pr = pool_workload(td, 3, 0.5)
            extras["pool_fanout_merge"] = {k: {f: v[f] for f in ("gpu_ms", "cpu_port_ms", "happy_plans", "pools", "pools_match_cpu_port",
                                                                "reference_binary_ms_build_container") if f in v} for k, v in pr.items()}
        except Exception as e:
            extras["pool_fanout_merge"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        line["other_workloads"] = extras
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.workload, n, args.cpu_seconds)
        line["cpu_baseline"]["glpk"] = glpk_baseline()   # solver.py's own path, iff cvxopt imports here
        if args.workload == "tick":   # same instance (seed 1) through the C port: the totals must agree
            line["total_matches_cpu_port"] = bool(int(total) == int(line["cpu_baseline"]["last_total"]))
    if dist.is_initialized():
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio: drain it so the JSON line is the LAST line
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
