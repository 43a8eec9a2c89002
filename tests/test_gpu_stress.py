"""Seeded, time-boxed slices of the stress tools under tools/ (the deep evidence belongs in what
`pytest -m gpu` runs).  Every tool checks its instances against the oracle or against the device's LP certificate
(total == dual bound) + the permutation + the matrix itself, and prints "... N failures"; the tests run a fixed seed
and a fixed instance count in a child process and require 0 failures."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_tool(name, seed, count, env=None, limit=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name), str(seed), str(limit), str(count)], env=e,
                       capture_output=True, text=True, timeout=limit + 120)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    m = re.search(r"(\d+) instances.*?(\d+) failures", r.stdout)
    assert m, r.stdout[-2000:]
    assert "FAIL" not in r.stdout, r.stdout[-3000:]
    return int(m.group(1)), int(m.group(2)), r.stdout


@pytest.mark.gpu
def test_stress_small_families_against_the_oracle(td):
    """tools/gpu_stress.py: every family (perf.jl, heuristic.py, |a-b|, thresholded, wide, negative, constant, padded
    rectangular models, line metrics with missing cabs / requests and perturbed cells), 2 <= n < 1400, vs oracle.assign"""
    cnt, bad, _ = _run_tool("gpu_stress.py", 20261, 260)
    assert cnt == 260 and bad == 0


@pytest.mark.gpu
def test_stress_large_instances_certificate(td):
    """tools/gpu_stress_large.py: 4096 <= n <= 12 288 (forest finisher, cooperative finisher, multi-chunk kernels, the
    general solver alone on |a-b| and 2-D grids, line models with missing cabs): LP certificate + permutation"""
    cnt, bad, _ = _run_tool("gpu_stress_large.py", 20262, 24)
    assert cnt == 24 and bad == 0


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", ["0", "8"])
def test_stress_wide_one_byte_rows(td, blocks):
    """tools/gpu_stress_bid0.py: 1-byte instances of n >= 12 288 (round 0 out of the compress pass; with 8 blocks and sizes
    aligned to 128 the block-local start of csrc/td_blocks.h): ties, single minima, padded rows / columns, thresholds"""
    cnt, bad, _ = _run_tool("gpu_stress_bid0.py", 20263, 45, env={"TD_BLOCKS": blocks, "STRESS_ALIGN": "128"})
    assert cnt == 45 and bad == 0


@pytest.mark.gpu
def test_stress_ticks_and_in_process_shards(td):
    """tools/gpu_stress_tick.py: td_tick against the oracle's pipeline (random cabs / requests / stands / tables / drop
    times / stop sizes, incl. ticks whose LCM ends on big_cost: no solve), the line path and the padded auction over
    1 - 8 in-process shards"""
    cnt, bad, out = _run_tool("gpu_stress_tick.py", 20264, 120)
    assert cnt == 120 and bad == 0


_BID0_CHILD = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
out = {}
g = torch.Generator(device="cuda").manual_seed(5)
for name, n in [("g1", 12288), ("g1", 16384), ("uniq", 12288), ("uniq", 16384), ("padrows", 12288), ("padboth", 16384), ("g4", 12288)]:
    if name == "g4":
        c = torch.randint(0, 4, (n, n), dtype=torch.int32, device="cuda", generator=g)
    elif name == "uniq":   # every row has ONE cell at its minimum: round 0 raises the price by second - first
        c = torch.randint(3, 200, (n, n), dtype=torch.int32, device="cuda", generator=g)
        c[torch.arange(n, device="cuda"), torch.randint(0, n, (n,), device="cuda", generator=g)] = 0
        c[::7] += 1000
    else:
        c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
        if name in ("padrows", "padboth"):
            c[torch.randperm(n, device="cuda", generator=g)[:n // 5]] = 250
        if name == "padboth":
            c[:, torch.randperm(n, device="cuda", generator=g)[:n // 3]] = 250
    r2c, tot, dual = td.assign(c, want_dual=True)
    assert tot == dual and sorted(r2c.tolist()) == list(range(n))
    out["%%s_%%d" %% (name, n)] = [int(tot), hashlib.sha1(np.asarray(r2c).tobytes()).hexdigest()]
print(json.dumps(out))
''' % ROOT


@pytest.mark.gpu
def test_round0_out_of_the_compress_pass_is_bit_identical(td):
    """tools/r3_bid0_check.py as a test: TD_BID0=1 (k_compress_reg<.., BID0> writes round 0's bids) against TD_BID0=0
    (round 0 as its own k_bid launch) at n = 12 288 and 16 384, incl. the single-minimum-per-row family — the same
    row_to_col, bit for bit.  (TD_BLOCKS=0: the block-local start replaces round 0 by its own zero-cell rule.)"""
    import json
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, TD_BID0=mode, TD_BLOCKS="0")
        r = subprocess.run([sys.executable, "-c", _BID0_CHILD], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["1"] == res["0"]
    assert len(res["1"]) == 7


_LAZY_CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
out = {}
g = torch.Generator(device="cuda").manual_seed(11)
for name, n in [("g1", 12288), ("g1", 16384), ("uniq", 12288), ("wide8", 12288), ("padboth", 16384), ("g4", 12288), ("sparse0", 16384)]:
    if name == "g4":
        c = torch.randint(0, 4, (n, n), dtype=torch.int32, device="cuda", generator=g)
    elif name == "uniq":      # ONE cell at the minimum of every row: the block-local start places few rows
        c = torch.randint(3, 200, (n, n), dtype=torch.int32, device="cuda", generator=g)
        c[torch.arange(n, device="cuda"), torch.randint(0, n, (n,), device="cuda", generator=g)] = 0
        c[::7] += 1000
    elif name == "wide8":     # ~ n / 250 zero cells per row
        c = torch.randint(-125, 125, (n, n), dtype=torch.int32, device="cuda", generator=g)
    elif name == "sparse0":   # zero cells mostly OUTSIDE the diagonal blocks: the two-hop pass over the whole matrix has work
        c = torch.randint(1, 60, (n, n), dtype=torch.int32, device="cuda", generator=g)
        cols = (torch.arange(n, device="cuda") * 7919 + 4321) %% n
        for k in range(6):
            c[torch.arange(n, device="cuda"), (cols + k * 2731) %% n] = 0
    else:
        c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
        if name == "padboth":
            c[torch.randperm(n, device="cuda", generator=g)[:n // 5]] = 250
            c[:, torch.randperm(n, device="cuda", generator=g)[:n // 3]] = 250
    for want_dual in (False, True):
        got = td.assign(c, want_dual=want_dual)
        r2c, tot = got[0], got[1]
        assert sorted(np.asarray(r2c).tolist()) == list(range(n))
        assert int(c[torch.arange(n, device="cuda"), torch.as_tensor(np.asarray(r2c), device="cuda").long()].sum().item()) == tot
        if want_dual:
            assert got[2] == tot
        out["%%s_%%d_%%d" %% (name, n, int(want_dual))] = [int(tot), hashlib.sha1(np.asarray(r2c).tobytes()).hexdigest()]
print(json.dumps(out))
""" % ROOT


@pytest.mark.gpu
def test_lazy_narrow_copy_is_bit_identical(td):
    """TD_LAZY_CC=1 (the compress pass of the block-local start stores the diagonal slices of the 1-byte copy only, the
    two-hop pass over the whole matrix reads the int32 matrix, k_compress_rest writes the other cells when rows are left
    or the dual bound is asked for) against TD_LAZY_CC=0 (the whole copy at once): the same row_to_col bit for bit, with
    and without the dual bound, on families where phase A places everything, nearly nothing, and where the zero cells lie
    outside the diagonal blocks."""
    import json
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, TD_LAZY_CC=mode)
        r = subprocess.run([sys.executable, "-c", _LAZY_CHILD], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["1"] == res["0"]
    assert len(res["1"]) == 14
