"""1-byte instances wide enough for round 0 out of the compress pass (n >= 12 288; k_compress_reg<.., BID0>), checked
by the LP certificate (total == dual bound), the permutation and the matrix itself (dev tool).
python tools/gpu_stress_bid0.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cnt = bad = 0
kinds = {}
max_cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9   # optional: stop after this many instances (tests/test_gpu_stress.py)
while time.time() < t_end and cnt < max_cnt:
    kind = ["g1", "ties", "r250", "uniq", "padrows", "padcols", "padboth", "thresh", "rowshift"][int(rng.integers(0, 9))]
    n = 4 * int(rng.integers(3072, 5200)) if rng.random() < 0.8 else 4 * int(rng.integers(5200, 8200))
    if os.environ.get("STRESS_ALIGN"):   # STRESS_ALIGN=128: sizes the block-local start (td_blocks.h, 8 blocks) applies to
        al = int(os.environ["STRESS_ALIGN"])
        n = (n + al - 1) // al * al
    kinds[kind] = kinds.get(kind, 0) + 1
    if kind == "g1":
        c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda")
    elif kind == "ties":
        c = torch.randint(0, int(rng.integers(2, 6)), (n, n), dtype=torch.int32, device="cuda")
    elif kind == "r250":
        c = torch.randint(0, 250, (n, n), dtype=torch.int32, device="cuda")
    elif kind == "uniq":   # one cell at the minimum of every row
        c = torch.randint(3, 200, (n, n), dtype=torch.int32, device="cuda")
        c[torch.arange(n, device="cuda"), torch.randint(0, n, (n,), device="cuda")] = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
    elif kind == "rowshift":   # rows with their own offsets (negative too): the range per row still fits one byte
        c = torch.randint(0, 200, (n, n), dtype=torch.int32, device="cuda") + torch.randint(-10**6, 10**6, (n, 1), dtype=torch.int32, device="cuda")
    elif kind == "thresh":     # |a - b| cut at a threshold: few distinct values, huge tie classes
        a = torch.randint(0, 60, (n,), device="cuda"); b = torch.randint(0, 60, (n,), device="cuda")
        c = (a[:, None] - b[None, :]).abs().to(torch.int32)
        c[c >= 10] = 250
        c = c.contiguous()
    else:
        c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda")
        if kind in ("padrows", "padboth"):
            c[torch.randperm(n, device="cuda")[: int(rng.integers(1, n // 2))]] = 250
        if kind in ("padcols", "padboth"):
            c[:, torch.randperm(n, device="cuda")[: int(rng.integers(1, n // 2))]] = 250
    first = td.assign(c, n) if os.environ.get("STRESS_BOTH") else None   # without the dual bound first: a lazy narrow copy (TD_LAZY_CC) then stays partial when phase A places every row
    r2c, tot, dual = td.assign(c, n, want_dual=True)
    if first is not None and not (first[1] == tot and np.array_equal(np.asarray(first[0]), np.asarray(r2c))):
        bad += 1
        print("FAIL (with / without the dual bound differ)", kind, n, first[1], tot, flush=True)
    r = torch.from_numpy(np.asarray(r2c)).cuda().long()
    ok = tot == dual and bool((torch.sort(r).values == torch.arange(n, device="cuda")).all()) and \
        int(c[torch.arange(n, device="cuda"), r].long().sum().item()) == tot
    cnt += 1
    if not ok:
        bad += 1
        print("FAIL", kind, n, tot, dual, td.last_stats(), flush=True)
    del c
print("stress bid0: %d instances %s, %d failures" % (cnt, kinds, bad))
