import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
n = 10269
for rr, rc in ((5000, 3000), (3000, 5000), (10000, 5700), (6000, 5000), (5700, 10000), (2000, 1000), (8000, 6100), (9500, 7600), (4000, 2100)):
    c = torch.full((n, n), 250000, dtype=torch.int32, device="cuda")
    c[:rr, :rc] = torch.randint(0, 50, (rr, rc), dtype=torch.int32, device="cuda")
    for rep in range(2):
        t0 = time.time(); r2c, tot, dual = td.assign(c, n, want_dual=True); dt = time.time() - t0
    print(rr, rc, "%.1f ms" % (1e3 * dt), tot == dual, td.last_stats(), flush=True)
