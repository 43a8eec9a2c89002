"""BASELINE configs[3] on ONE GPU, timed per shard: 8 row shards of 8192 x 65536 driven in one process
(the geometry of tests/test_gpu_configs.py::test_config3_shard_geometry_in_process), every C call bracketed by
the library's HIP-event profiler.  What a RANK of an 8-GPU run would spend = the largest shard's time per phase;
what does not shard = the apply kernels (replicated), the finisher on rank 0, the exchanges.
usage: python tools/r3_shard_time.py [n] [world] [builder]   builder: gen (td_gen_uniform rows) | cost (td_cost_build_rows, |a-b| positions)"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi, sharded

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
builder = sys.argv[3] if len(sys.argv) > 3 else "gen"
td.init(0)
lib = _ffi.lib()
_ffi.check(lib.td_profile_enable(1))


def timed(fn):
    """(result, {class: ms}) of one library call, by the library's own HIP events"""
    _ffi.check(lib.td_profile_reset())
    t0 = time.perf_counter()
    out = fn()
    _ffi.check(lib.td_synchronize())
    wall = 1e3 * (time.perf_counter() - t0)
    res = {}
    for name, k in _ffi.TD_K.items():
        ms = ctypes.c_double(0)
        cnt = ctypes.c_int64(0)
        _ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            res[name] = ms.value
    res["_wall"] = wall
    return out, res


full = torch.empty((n, n), dtype=torch.int32, device="cuda")
rng = np.random.default_rng(1)
a = rng.integers(0, 10 * n, n).astype(np.int32)
b = rng.integers(0, 10 * n, n).astype(np.int32)
S = 1000
table = torch.from_numpy(rng.integers(10, 41, (S, S)).astype(np.int32)).cuda()
def run():
    global phases, round_bid, round_apply, fin, st, tot, dual, tt, width
    phases = {}


    def add(name, r, per_shard=True):
        ms = sum(v for k, v in r.items() if not k.startswith("_"))
        phases.setdefault(name, []).append(ms)


    shards = []
    try:
        for r in range(world):
            row0, nrows, rps = sharded.shard_bounds(n, world, r)
            rows = full[row0:row0 + nrows]
            if builder in ("gen", "padded"):
                _, t = timed(lambda: _ffi.check(lib.td_gen_uniform(n, 7, 10, 40, row0, nrows, rows.data_ptr())))
            else:   # a general S x S table with perf.jl's value range: the shard builds its rows from the replicated positions
                da, db = torch.from_numpy(a % S).cuda(), torch.from_numpy(b % S).cuda()
                _, t = timed(lambda: _ffi.check(lib.td_cost_build_rows(da.data_ptr(), None, n, db.data_ptr(), None, n, table.data_ptr(), S, 250000, -1, 0,
                                                                       row0, nrows, rows.data_ptr())))
            if builder == "padded":   # a third of the rows are dummy cabs (constant rows of big_cost)
                dummy = torch.from_numpy(np.random.default_rng(5).permutation(n)[:n // 3]).cuda()
                mine = dummy[(dummy >= row0) & (dummy < row0 + nrows)] - row0
                rows[mine] = 250000
                torch.cuda.synchronize()
            add("build", t)
            shards.append(sharded.HipShard(n, row0, nrows, rows, share_torch_stream=False))
        for s in shards:
            s.fused_round0(os.environ.get("TD_SHARD_FUSED0", "1") != "0")
        width = None
        for w in (1, 2, 4):
            oks = []
            ts = []
            for s in shards:
                ok, t = timed(lambda: s.compress(w))
                oks.append(ok)
                ts.append(t)
            if all(oks):
                for t in ts:
                    add("compress", t)
                width = w
                break
        if os.environ.get("TD_DEFER_CONST", "1") != "0":
            masks = [s.const_mask() for s in shards]
            tot_mask = masks[0]
            for m in masks[1:]:
                tot_mask += m
            torch.cuda.synchronize()
            for s in shards:
                s.set_const_mask(tot_mask)
        grange = max(s.range() for s in shards)
        for s in shards:
            _, t = timed(lambda: s.begin(grange))
            add("begin", t)
        keys = [s.new_keys() for s in shards]
        round_bid, round_apply = [], []
        for rnd in range(sharded.DEFAULT_ROUNDS):
            bt = []
            for s, k in zip(shards, keys):
                _, t = timed(lambda: s.bid(rnd, k))
                bt.append(sum(v for kk, v in t.items() if not kk.startswith("_")))
            red = keys[0].clone()
            for k in keys[1:]:
                red = torch.maximum(red, k)
            torch.cuda.synchronize()
            at = []
            for s, k in zip(shards, keys):
                k.copy_(red)
                torch.cuda.synchronize()
                _, t = timed(lambda: s.apply(rnd, k))
                at.append(sum(v for kk, v in t.items() if not kk.startswith("_")))
            round_bid.append(bt)
            round_apply.append(at)
        _, t = timed(lambda: shards[0].finish([s.cc_ref() for s in shards], rps))
        fin = t
        st = td.last_stats()
        owner = shards[0].get_owner()
        price = shards[0].get_price()
        torch.cuda.synchronize()
        tot = dual = 0
        for s in shards[1:]:
            s.set_owner(owner)
            s.set_price(price)
        tt = []
        for s in shards:
            (tv, dv), t = timed(lambda: s.totals(True))
            tot += tv
            dual += dv
            tt.append(sum(v for kk, v in t.items() if not kk.startswith("_")))
    finally:
        for s in shards:
            s.close()



run()          # warm-up: first-touch allocations, code object loads, the cooperative kernel's first launch
run()
rank_time = lambda xs: max(xs)
res = {
    "n": n, "world": world, "builder": builder, "bytes_per_cell": width, "total": int(tot), "dual": int(dual),
    "per_rank_ms": {
        "build (max over shards)": rank_time(phases["build"]),
        "compress (max over shards)": rank_time(phases["compress"]),
        "begin": rank_time(phases["begin"]),
        "bid per round (max over shards)": [round(max(b), 4) for b in round_bid],
        "apply per round (replicated on every rank)": [round(max(x), 4) for x in round_apply],
        "finisher on rank 0": {k: round(v, 4) for k, v in fin.items()},
        "totals (max over shards)": rank_time(tt),
    },
    "sum_over_shards_ms": {"build": sum(phases["build"]), "compress": sum(phases["compress"]),
                           "bid": sum(sum(b) for b in round_bid), "apply": sum(sum(x) for x in round_apply)},
    "finisher_stats": st,
}
pr = res["per_rank_ms"]
stream = pr["build (max over shards)"] + pr["compress (max over shards)"] + sum(pr["bid per round (max over shards)"]) + pr["totals (max over shards)"]
serial = sum(pr["apply per round (replicated on every rank)"]) + sum(v for k, v in fin.items() if not k.startswith("_")) + pr["begin"]
res["projection"] = {"sharded part per rank ms": stream, "replicated / rank-0 part ms": serial,
                     "exchanges": "%d MAX all-reduces of %d KiB + owner broadcast + 1 scalar SUM" % (sharded.DEFAULT_ROUNDS, n * 8 // 1024),
                     "kernel ms per step on %d GPUs (no exchange cost)" % world: stream + serial}
print(json.dumps(res, indent=1))
