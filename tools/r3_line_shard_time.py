"""G2 (|a - b| rows from td_cost_build_rows, positions 0..10n) over `world` in-process row shards on ONE GPU:
the sharded line-metric path (td_line_shard_*), per phase and per shard, by the library's HIP-event profiler.
What a rank of a `world`-GPU run spends per phase = the largest shard's time; the exchanges are four SUM
all-reduces of O(n) words.  usage: python tools/r3_line_shard_time.py [n] [world]"""
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
td.init(0)
lib = _ffi.lib()
_ffi.check(lib.td_profile_enable(1))


def timed(fn):
    _ffi.check(lib.td_profile_reset())
    out = fn()
    _ffi.check(lib.td_synchronize())
    tot = 0.0
    for name, k in _ffi.TD_K.items():
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        _ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        tot += ms.value
    return out, tot


rng = np.random.default_rng(1)
a = rng.integers(0, 10 * n, n).astype(np.int32)
b = rng.integers(0, 10 * n, n).astype(np.int32)
ref = int(np.abs(np.sort(a).astype(np.int64) - np.sort(b)).sum())
da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
full = torch.empty((n, n), dtype=torch.int32, device="cuda")


def run():
    shards, build = [], []
    try:
        for r in range(world):
            row0, nrows, _ = sharded.shard_bounds(n, world, r)
            rows = full[row0:row0 + nrows]
            _, t = timed(lambda: _ffi.check(lib.td_cost_build_rows(da.data_ptr(), None, n, db.data_ptr(), None, n, None, 0, 250000, -1, 0,
                                                                   row0, nrows, rows.data_ptr())))
            build.append(t)
            shards.append(sharded.HipShard(n, row0, nrows, rows, share_torch_stream=False))
        wss = [s.line_ws() for s in shards]
        phase_ms, seg_words = [], []
        t0 = time.perf_counter()
        for ph in range(sharded.LINE_PHASES):
            segs, ts = [], []
            for s, ws in zip(shards, wss):
                (off, ln), t = timed(lambda: s.line_phase(ph, ws))
                segs.append(ws[off:off + ln])
                ts.append(t)
            red = segs[0]
            for sg in segs[1:]:
                red += sg
            for sg in segs[1:]:
                sg.copy_(red)
            torch.cuda.synchronize()
            phase_ms.append(ts)
            seg_words.append(int(red.numel()))
        out = [s.line_result(ws) for s, ws in zip(shards, wss)]
        wall = 1e3 * (time.perf_counter() - t0)
    finally:
        for s in shards:
            s.close()
    return build, phase_ms, seg_words, out, wall


run()
build, phase_ms, seg_words, out, wall = run()
r2c = np.concatenate([r for _, _, r in out])
res = {
    "n": n, "world": world, "accepted": all(acc for acc, _, _ in out), "total": out[0][1], "sorted_matching_total": ref,
    "is_permutation": bool((np.sort(r2c) == np.arange(n)).all()),
    "per_rank_ms": {"build (max over shards)": max(build),
                    "phase 0..3 (max over shards)": [round(max(t), 4) for t in phase_ms]},
    "sum_over_shards_ms": {"build": sum(build), "phases": [round(sum(t), 4) for t in phase_ms]},
    "exchange_words_per_phase": seg_words,
    "in_process_wall_ms (all shards serial, host syncs included)": round(wall, 3),
}
res["projection"] = {"kernel ms per step on %d GPUs (no exchange cost)" % world:
                     round(max(build) + sum(max(t) for t in phase_ms), 4)}
print(json.dumps(res, indent=1))
