"""LCM timing at several sizes (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
rng = np.random.default_rng(0)
for kind, n in [("tick", 1300), ("g2thr", 1000), ("g2thr", 4096), ("g2thr", 16384), ("g1", 4096), ("g4", 1000)]:
    if kind == "tick":
        a = rng.integers(0, 50, n); b = rng.integers(0, 50, 900)
        _, c = td.cost_build(a, b, None, fill=250000, threshold=10)
        f = lambda: td.LCM_simulator(ct, max_non_lcm=600)
    elif kind == "g2thr":
        a = rng.integers(0, 10 * n, n); b = rng.integers(0, 10 * n, n)
        ct = torch.empty((n, n), dtype=torch.int32, device="cuda"); td.cost_build(a, b, None, fill=250000, threshold=-1, out=ct); c = None
        f = lambda: td.LCM(n, ct, threshold=10)
    elif kind == "g1":
        c = rng.integers(10, 41, (n, n)).astype(np.int32); f = lambda: td.LCM(n, ct, threshold=10)
    else:
        c = rng.integers(1, 40, (n, n)).astype(np.int32); f = lambda: td.LCM_heuristic(n, ct)
    if c is not None: ct = torch.from_numpy(np.ascontiguousarray(c)).cuda()
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    npairs = len(r[0]) if kind == "tick" else len(r[1])
    print(f"{kind:6s} n={n:6d}: {1e3*dt:9.2f} ms  pairs={npairs}", flush=True)
