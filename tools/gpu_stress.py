"""One-off stress: many random instances of all families against the oracle (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import taxidispatcher_amd as td
from oracle import oracle
from test_gpu_parity import make_instance
td.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
cnt = bad = 0
slow = []
max_cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9   # optional: stop after this many instances (tests/test_gpu_stress.py)
while time.time() < t_end and cnt < max_cnt:
    kind = ["g1", "g4", "g2", "g3", "wide", "neg", "const", "rect", "line", "lineu", "linep"][int(rng.integers(0, 11))]
    n = int(rng.integers(2, 1400)) if kind not in ("g2", "wide") else int(rng.integers(2, 700))
    if kind == "rect":   # padded rectangular model: constant rows / columns, sometimes permuted
        rr, rc = int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))
        c = np.full((n, n), 250000, np.int32)
        blk = rng.integers(0, int(rng.choice([10, 50, 100000])), (rr, rc)).astype(np.int32)
        if rng.random() < 0.5:
            blk[rng.random(blk.shape) < 0.7] = 250000
        c[:rr, :rc] = blk
        if rng.random() < 0.3:
            c = np.ascontiguousarray(c[rng.permutation(n)][:, rng.permutation(n)])
    elif kind in ("line", "lineu", "linep"):
        # line metric: balanced / padded with k missing cabs or requests / one cell perturbed
        S = int(rng.choice([2, 3, 7, 50, 10 * n, 10**6]))
        k = 0 if kind != "lineu" else int(rng.integers(1, min(40, n - 1) + 1)) if n > 2 else 0
        nc, nr = (n - k, n) if rng.random() < 0.5 else (n, n - k)
        a, b = rng.integers(0, S, nc), rng.integers(0, S, nr)
        c = np.full((n, n), 250000, np.int32)
        c[:nc, :nr] = np.abs(a[:, None] - b[None, :])
        if kind == "linep":
            c[int(rng.integers(0, n)), int(rng.integers(0, n))] = int(rng.integers(0, 3 * S + 5))
        if rng.random() < 0.2:
            c = np.ascontiguousarray(c[rng.permutation(n)][:, rng.permutation(n)])
    else:
        c = make_instance(kind, n, rng)
    t0 = time.time()
    r2c, tot, dual = td.assign(c, want_dual=True)
    dt = time.time() - t0
    slow.append((dt / max(n, 1) ** 2, dt, kind, n, dict(td.last_stats())))
    ref = oracle.assign(c)[0]
    ok = tot == ref == dual and sorted(r2c.tolist()) == list(range(n)) and int(c[np.arange(n), r2c].astype(np.int64).sum()) == tot
    cnt += 1
    if not ok:
        bad += 1
        print("FAIL", kind, n, tot, ref, dual, td.last_stats(), flush=True)
print("stress: %d instances, %d failures" % (cnt, bad))
slow.sort(key=lambda x: -x[1])
for x in slow[:12]:   # slowest per cell: where the solver spends unusually long
    print("slow: %.2f ms %s n=%d %s" % (1e3 * x[1], x[2], x[3], x[4]))
