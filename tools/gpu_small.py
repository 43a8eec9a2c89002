import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import taxidispatcher_amd as td
from oracle import oracle
td.init(0)
rng = np.random.default_rng(3)
for kind, n in [("g2", 400), ("g2", 600), ("g2", 1000), ("wide", 400), ("wide", 1000), ("g3", 600)]:
    if kind == "g2":
        a = rng.integers(0, 10 * n, n); b = rng.integers(0, 10 * n, n); c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    elif kind == "wide":
        c = rng.integers(0, 1000000, (n, n)).astype(np.int32)
    else:
        a = rng.integers(0, 50, n); b = rng.integers(0, 50, n)
        c = np.abs(a[:, None] - b[None, :]).astype(np.int32); c[c >= 10] = 250000; c[:, int(.363 * n):] = 250000
    td.assign(c)
    t0 = time.time(); r2c, tot = td.assign(c); dt = time.time() - t0
    t1 = time.time(); ref = oracle.assign(c)[0]; dc = time.time() - t1
    print(kind, n, "OK" if tot == ref else "FAIL", "gpu %.2f ms  cpu-oracle %.2f ms" % (1e3 * dt, 1e3 * dc), td.last_stats())
