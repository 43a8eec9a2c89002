"""Round 4: the block-local start (csrc/td_blocks.h) on one GPU.
  1. td_assign on the perf.jl instance with TD_BLOCKS = 0 / 1 / 8: optimum, certificate, permutation, time
  2. 8 in-process row shards through the new sequence (phase A -> one exchange -> rounds / finisher for what is left):
     bit-identical to td_assign with 8 blocks
usage: python tools/r4_blocks.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi, sharded

td.init(0)
lib = _ffi.lib()
sizes = [int(x) for x in sys.argv[1:]] or [16384]


def timed_assign(full, reps=5):
    out = None
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = td.assign(full, want_dual=False)
        ts.append(1e3 * (time.perf_counter() - t0))
    return out, min(ts), sorted(ts)[len(ts) // 2]


def drive_blocks(full, n, world):
    shards = []
    try:
        for r in range(world):
            row0, nrows, rps = sharded.shard_bounds(n, world, r)
            shards.append(sharded.HipShard(n, row0, nrows, full[row0:row0 + nrows], share_torch_stream=False))
        for s in shards:
            s.blocks_start(True)
        fits = [s.compress(1) for s in shards]
        for s, f in zip(shards, fits):
            if f and s.blocks_pending():
                s.phase_a()
        segs = [s.state_segment(rps, f) for s, f in zip(shards, fits)]
        allseg = torch.cat(segs)
        torch.cuda.synchronize()
        summs = [s.state_import(world, r, rps, allseg) for r, s in enumerate(shards)]
        assert all(x == summs[0] for x in summs), summs
        summ = summs[0]
        assert summ["fit"] and summ["ran"], summ
        for s in shards:
            s.begin(summ["range"])
        if summ["left"] > 0:
            keys = [s.new_keys() for s in shards]
            for rnd in range(sharded.DEFAULT_ROUNDS):
                for s, k in zip(shards, keys):
                    s.bid(rnd, k)
                red = keys[0].clone()
                for k in keys[1:]:
                    red = torch.maximum(red, k)
                torch.cuda.synchronize()
                for s, k in zip(shards, keys):
                    k.copy_(red)
                    torch.cuda.synchronize()
                    s.apply(rnd, k)
            shards[0].finish([s.cc_ref() for s in shards], rps)
            owner, price = shards[0].get_owner(), shards[0].get_price()
            torch.cuda.synchronize()
            for s in shards[1:]:
                s.set_owner(owner)
                s.set_price(price)
        tot = dual = 0
        parts = []
        for s in shards:
            t, d = s.totals(True)
            tot += t
            dual += d
            parts.append(s.row_to_col())
        return np.concatenate(parts), tot, dual, summ
    finally:
        for s in shards:
            s.close()


for n in sizes:
    full = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(lib.td_gen_uniform(n, 7, 10, 40, 0, n, full.data_ptr()))
    _ffi.check(lib.td_synchronize())
    res = {}
    for blocks in (0, 1, 8):
        lib.td_set_blocks(blocks)
        r2c, tot, dual = td.assign(full, want_dual=True)
        assert tot == dual == 10 * n, (blocks, tot, dual)
        assert sorted(r2c.tolist()) == list(range(n))
        (_, _), best, med = timed_assign(full)
        st = td.last_stats()
        print("n=%d td_assign blocks=%d: total %d = dual, %.3f ms best / %.3f median, rounds %d, free rows to the finisher %d" %
              (n, blocks, tot, best, med, st["bid_rounds"], st["sap_free_rows"]), flush=True)
        res[blocks] = r2c
    for world in (8, 2):
        r2c, tot, dual, summ = drive_blocks(full, n, world)
        assert tot == dual == 10 * n, (tot, dual)
        print("n=%d, %d in-process shards, block-local start: total %d = dual, left after phase A %d, identical to td_assign(8 blocks): %s" %
              (n, world, tot, summ["left"], bool(np.array_equal(r2c, res[8]))), flush=True)
    lib.td_set_blocks(-1)
    del full
    torch.cuda.empty_cache()
