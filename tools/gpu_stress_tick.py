"""Stress of the one-call tick (td_tick) and of the sharded paths against the oracle's pipeline (dev tool):
random numbers of cabs / requests, stand counts, |a-b| or a general table, drop times, LCM stop sizes.
usage: python tools/gpu_stress_tick.py [seed] [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import taxidispatcher_amd as td
from taxidispatcher_amd import sharded
from oracle import oracle

td.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
BIG = 250000


def reference(cab_to, dem_from, dist, drop, stop):
    n_o, cost_o = oracle.cost_build(cab_to, dem_from, dist, BIG, drop)
    if 0 <= stop < n_o:
        _, rows, cols, lm = oracle.lcm(cost_o, mask=BIG, stop_value_on=1, stop_value=BIG, stop_size=stop, sum_below=BIG, java_scan=1)
    else:
        rows, cols, lm = np.zeros(0, np.int64), np.zeros(0, np.int64), BIG
    keep_c = np.setdiff1d(np.arange(len(cab_to)), rows)
    keep_d = np.setdiff1d(np.arange(len(dem_from)), cols)
    n2, cost2 = oracle.cost_build(np.asarray(cab_to)[keep_c], np.asarray(dem_from)[keep_d], dist, BIG, drop)
    tot = oracle.assign(cost2)[0] if n2 else 0
    return rows, cols, lm, keep_c, keep_d, n2, cost2, tot


cnt = bad = 0
kinds = {}
max_cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9   # optional: stop after this many instances (tests/test_gpu_stress.py)
while time.time() < t_end and cnt < max_cnt:
    what = ["tick", "tick", "tick", "line_sh", "padded_sh"][int(rng.integers(0, 5))]
    kinds[what] = kinds.get(what, 0) + 1
    if what == "tick":
        S = int(rng.choice([3, 10, 50, 200, 2000]))
        ns, nd = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        cab_to, dem_from = rng.integers(0, S, ns), rng.integers(0, S, nd)
        dist = None
        if S <= 200 and rng.random() < 0.5:
            dist = rng.integers(0, int(rng.choice([5, 25, 400])), (S, S)).astype(np.int32)
        drop = int(rng.choice([1, 3, 10, 40, 10**6]))
        stop = int(rng.choice([-1, 0, 1, 50, 220, 600, 5000]))
        rows, cols, lm, keep_c, keep_d, n2, cost2, tot = reference(cab_to, dem_from, dist, drop, stop)
        t = td.tick(cab_to, dem_from, dist, big_cost=BIG, drop_time=drop, max_non_lcm=stop)
        ok = t["lcm_rows"].tolist() == rows.tolist() and t["lcm_cols"].tolist() == cols.tolist()
        ok = ok and (not len(rows) or t["lcm_min_val"] == lm)
        ok = ok and t["kept_cabs"].tolist() == keep_c.tolist() and t["kept_dems"].tolist() == keep_d.tolist()
        ok = ok and t["n_rest"] == n2
        if 0 <= stop < max(ns, nd) and lm == BIG:   # Simulator.java:188-189: the LCM ran and ended on big_cost (possibly at its first look), the tick has nothing for the solver
            ok = ok and not t["solved"] and t["total"] == 0 and len(t["row_to_col"]) == 0
            kinds["tick without a solve"] = kinds.get("tick without a solve", 0) + 1
        else:
            ok = ok and t["total"] == tot and (t["solved"] or n2 == 0)
            r2c = t["row_to_col"]
            ok = ok and sorted(r2c.tolist()) == list(range(n2))
            ok = ok and (n2 == 0 or int(cost2[np.arange(n2), r2c].astype(np.int64).sum()) == tot)
        desc = (what, ns, nd, S, dist is not None, drop, stop)
        if not ok and os.environ.get("STRESS_VERBOSE"):
            print("  detail: lcm rows ok %s cols ok %s (k %d vs %d) lm %s vs %s kept ok %s / %s n_rest %s vs %s total %s vs %s solved %s" % (
                t["lcm_rows"].tolist() == rows.tolist(), t["lcm_cols"].tolist() == cols.tolist(), len(t["lcm_rows"]), len(rows), t["lcm_min_val"], lm,
                t["kept_cabs"].tolist() == keep_c.tolist(), t["kept_dems"].tolist() == keep_d.tolist(), t["n_rest"], n2, t["total"], tot, t["solved"]), flush=True)
    else:
        n = int(rng.integers(2, 1200))
        world = int(rng.integers(1, 9))
        if what == "line_sh":
            Sp = int(rng.choice([2, 7, 50, 10 * n, 10**6]))
            a, b = rng.integers(0, Sp, n), rng.integers(0, Sp, n)
            c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
            if rng.random() < 0.3:   # one perturbed cell: may or may not stay certifiable
                c[int(rng.integers(0, n)), int(rng.integers(0, n))] = int(rng.integers(0, 3 * Sp + 5))
            if rng.random() < 0.3:
                c = np.ascontiguousarray(c[rng.permutation(n)][:, rng.permutation(n)])
        else:
            c = rng.integers(0, int(rng.choice([10, 50, 100000])), (n, n)).astype(np.int32)
            c[rng.permutation(n)[:int(rng.integers(0, n))]] = BIG
        ref = oracle.assign(c)[0]
        full = torch.from_numpy(c).cuda()
        shards = []
        try:
            for r in range(world):
                row0, nrows, rps = sharded.shard_bounds(n, world, r)
                shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
            if what == "line_sh":
                got = sharded.line_sharded(shards, None)
                ok = True
                if got is not None:
                    r2c = np.concatenate(got[1])
                    ok = got[0] == ref and sorted(r2c.tolist()) == list(range(n)) and int(c[np.arange(n), r2c].astype(np.int64).sum()) == ref
                    kinds["line_sh accepted"] = kinds.get("line_sh accepted", 0) + 1
            else:   # the auction over in-process shards with the constant rows deferred (sharded.solve_shards_in_process)
                r2c, tot, dual, _ = sharded.solve_shards_in_process(shards)
                ok = tot == ref == dual and sorted(r2c.tolist()) == list(range(n)) and int(c[np.arange(n), r2c].astype(np.int64).sum()) == ref
        finally:
            for s in shards:
                s.close()
        desc = (what, n, world)
    cnt += 1
    if not ok:
        bad += 1
        print("FAIL", desc, flush=True)
print("stress tick/sharded: %d instances %s, %d failures" % (cnt, kinds, bad))
