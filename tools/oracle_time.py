import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import oracle
rng = np.random.default_rng(1)
for kind, n in [("wide", 4096), ("g2", 4096), ("wide", 8192)]:
    if kind == "g2":
        a = rng.integers(0, 10 * n, n); b = rng.integers(0, 10 * n, n)
        c = np.abs(a[:, None] - b[None, :]).astype(np.int32)
    else:
        c = rng.integers(0, 10**6, (n, n)).astype(np.int32)
    t = time.time(); tot = oracle.assign(c)[0]; print("oracle", kind, n, "%.1f ms" % (1e3 * (time.time() - t)), tot, flush=True)
