import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import taxidispatcher_amd as td
td.init(0)
g = json.load(open("tests/golden/tick49_instance.json"))
n, cost = td.cost_build(g["cab_to"], g["dem_from"], None, fill=250000, threshold=10)
for rep in range(2):
    t0 = time.time(); r2c, tot = td.assign(cost); print("tick49", tot, "%.2f ms" % (1e3 * (time.time() - t0)), td.last_stats())
rng = np.random.default_rng(1)
for n in (600, 1300):
    a = rng.integers(0, 50, n); b = rng.integers(0, 50, n)
    c = np.abs(a[:, None] - b[None, :]).astype(np.int32); c[c >= 10] = 250000; c[:, int(.363 * n):] = 250000
    t0 = time.time(); r2c, tot = td.assign(c); print("g3", n, tot, "%.2f ms" % (1e3 * (time.time() - t0)), td.last_stats())
    c2 = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    t0 = time.time(); r2c, tot = td.assign(c2); print("S50 nothr", n, tot, "%.2f ms" % (1e3 * (time.time() - t0)), td.last_stats())
from oracle import oracle
for kind, n in [("g2", 1000), ("g2", 2048), ("wide", 1000), ("wide", 2048), ("g1", 4096)]:
    if kind == "g2":
        a = rng.integers(0, 10 * n, n); b = rng.integers(0, 10 * n, n); c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    elif kind == "wide":
        c = rng.integers(0, 1000000, (n, n)).astype(np.int32)
    else:
        c = rng.integers(10, 41, (n, n)).astype(np.int32)
    t0 = time.time(); r2c, tot, dual = td.assign(c, want_dual=True); dt = time.time() - t0
    ref = oracle.assign(c)[0]
    print(kind, n, tot, ref, dual, "OK" if tot == ref == dual else "FAIL", "%.2f ms" % (1e3 * dt), td.last_stats())
