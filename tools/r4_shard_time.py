"""BASELINE configs[3] on ONE GPU, timed per shard (round 4): 8 row shards of 8192 x 65536 driven in one process through
sharded.solve_shards_in_process — the steps a rank of an 8-GPU run makes — every library call bracketed by the library's
HIP-event profiler.  What a RANK spends = the largest shard's time per phase; the exchanges are listed, not timed.
usage: python tools/r4_shard_time.py [n] [world] [builder] [blocks]
   builder: gen (td_gen_uniform rows) | cost (td_cost_build_rows over a 1000-stand table) | padded   blocks: 1 | 0"""
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
blocks = (sys.argv[4] if len(sys.argv) > 4 else "1") != "0"
td.init(0)
lib = _ffi.lib()
_ffi.check(lib.td_profile_enable(1))


def timed(fn):
    """(result, kernel ms by the library's own HIP events, host wall ms) of one library call"""
    _ffi.check(lib.td_profile_reset())
    t0 = time.perf_counter()
    out = fn()
    _ffi.check(lib.td_synchronize())
    wall = 1e3 * (time.perf_counter() - t0)
    ms_all = 0.0
    for name, k in _ffi.TD_K.items():
        ms = ctypes.c_double(0)
        cnt = ctypes.c_int64(0)
        _ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            ms_all += ms.value
    return out, ms_all, wall


full = torch.empty((n, n), dtype=torch.int32, device="cuda")
rng = np.random.default_rng(1)
S = 1000
table = torch.from_numpy(rng.integers(10, 41, (S, S)).astype(np.int32)).cuda()
a = rng.integers(0, S, n).astype(np.int32)
b = rng.integers(0, S, n).astype(np.int32)


def run(want_dual):
    phases = {}

    def call(name, k, fn):
        out, ms, wall = timed(fn)
        phases.setdefault(name, {})[k] = (ms, wall)
        return out

    shards = []
    try:
        for r in range(world):
            row0, nrows, rps = sharded.shard_bounds(n, world, r)
            rows = full[row0:row0 + nrows]
            if builder in ("gen", "padded"):
                call("build", r, lambda: _ffi.check(lib.td_gen_uniform(n, 7, 10, 40, row0, nrows, rows.data_ptr())))
            else:
                da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
                call("build", r, lambda: _ffi.check(lib.td_cost_build_rows(da.data_ptr(), None, n, db.data_ptr(), None, n, table.data_ptr(), S,
                                                                           250000, -1, 0, row0, nrows, rows.data_ptr())))
            if builder == "padded":   # a third of the rows are dummy cabs (constant rows of big_cost)
                dummy = torch.from_numpy(np.random.default_rng(5).permutation(n)[:n // 3]).cuda()
                mine = dummy[(dummy >= row0) & (dummy < row0 + nrows)] - row0
                rows[mine] = 250
                torch.cuda.synchronize()
            shards.append(sharded.HipShard(n, row0, nrows, rows, share_torch_stream=False))
        r2c, tot, dual, info = sharded.solve_shards_in_process(shards, blocks=blocks, call=call, want_dual=want_dual)
    finally:
        for s in shards:
            s.close()
    return phases, r2c, tot, dual, info


_, r2c, tot, dual, _ = run(True)   # warm-up (first-touch allocations, code object loads) and the certificate: total == dual bound
assert sorted(r2c.tolist()) == list(range(n))
assert tot == dual, (tot, dual)
phases, r2c2, tot2, _, info = run(False)   # timed: what bench.py's sharded leg asks for (no certificate pass)
assert tot2 == tot and np.array_equal(r2c, r2c2)
per_rank, rank0_only, replicated = {}, {}, {}
for name, d in phases.items():
    ms = [v[0] for v in d.values()]
    if name == "finish":
        rank0_only[name] = round(max(ms), 4)
    elif name.startswith("apply") or name in ("import", "place_const"):
        replicated[name] = round(max(ms), 4)     # every rank does the same work on replicated state
    else:
        per_rank[name] = round(max(ms), 4)
sharded_ms = sum(per_rank.values())
serial_ms = sum(replicated.values()) + sum(rank0_only.values())
if info["path"] == "blocks":
    exch = ["line attempt: 1 SUM all-reduce of 2n+16 words (refused after it)",
            "1 all-gather of %d KiB (width flag + owners of the column slices + constant-row flags)" % ((16 + 2 * (n // world)) * 4 * world // 1024)]
    if info["left"]:
        exch += ["%d MAX all-reduces of %d KiB bid keys" % (sharded.DEFAULT_ROUNDS, n * 8 // 1024), "hipIpc handles all-gather + 1 flag broadcast",
                 "owner broadcast (%d KiB)" % (n * 4 // 1024)]
    exch += ["1 SUM all-reduce of 2 words (totals)"]
else:
    exch = ["line attempt: 1 SUM all-reduce", "width flag MIN all-reduce", "constant-row mask SUM all-reduce (%d KiB)" % (n * 4 // 1024),
            "range MAX all-reduce", "%d MAX all-reduces of %d KiB bid keys" % (sharded.DEFAULT_ROUNDS, n * 8 // 1024),
            "hipIpc handles all-gather + 1 flag broadcast", "owner broadcast (%d KiB)" % (n * 4 // 1024), "1 SUM all-reduce of 2 words (totals)"]
res = {"n": n, "world": world, "builder": builder, "sequence": info["path"], "rows_left_after_phase_a": info["left"],
       "finisher_ran": info["finisher_ran"], "total": int(tot), "dual": int(dual),
       "per_rank_kernel_ms (max over shards)": per_rank, "replicated_on_every_rank_ms": replicated, "rank0_only_ms": rank0_only,
       "kernel_ms_per_step_on_%d_gpus" % world: round(sharded_ms + serial_ms, 4),
       "exchanges": exch, "n_exchanges": len(exch) if info["path"] == "blocks" and not info["left"] else None,
       "host_wall_ms_per_call (max over shards)": {name: round(max(v[1] for v in d.values()), 3) for name, d in phases.items()}}
print(json.dumps(res, indent=1))
