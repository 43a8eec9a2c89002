"""solve-only time of perf.jl instances over several seeds for the phase-A tunables given in the environment"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ts = []
for sd in range(seeds):
    g = torch.Generator(device="cuda").manual_seed(1000 + sd)
    c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
    td.assign(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r2c, tot = td.assign(c)[:2]
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
    assert tot == 10 * n
print("n=%d %s: mean %.4f ms  min %.4f  max %.4f  (%s)" % (n, {k: os.environ[k] for k in os.environ if k.startswith("TD_")}, np.mean(ts), min(ts), max(ts), " ".join("%.3f" % t for t in ts)))
