"""Large random instances checked by the LP certificate (total == dual bound) and the permutation:
exercises the cooperative finisher and the multi-chunk kernels (dev tool).
python tools/gpu_stress_large.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cnt = bad = 0
max_cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9   # optional: stop after this many instances (tests/test_gpu_stress.py)
while time.time() < t_end and cnt < max_cnt:
    kind = ["wide", "g2", "mid", "rect", "g1", "g2gen", "geo2", "g2u"][int(rng.integers(0, 8))]
    n = int(rng.integers(4096, 12289))
    td.set_line_metric(kind != "g2gen")   # g2gen: the general solver alone on the |a-b| geometry
    if kind == "wide":
        c = torch.randint(0, 10**6, (n, n), dtype=torch.int32, device="cuda")
    elif kind == "mid":     # u16 rows
        c = torch.randint(0, 40000, (n, n), dtype=torch.int32, device="cuda")
    elif kind == "g1":
        c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda")
    elif kind == "geo2":    # 2-D Manhattan grid: 2-byte rows, wide and tie-free (redone as 4-byte cells for k_sapx)
        ax, ay, bx, by = (torch.randint(0, 4000, (n,), device="cuda") for _ in range(4))
        c = ((ax[:, None] - bx[None, :]).abs() + (ay[:, None] - by[None, :]).abs()).to(torch.int32).contiguous()
    elif kind == "g2u":     # line metric with k missing cabs or requests
        k = int(rng.integers(1, 40))
        nc, nr = (n - k, n) if rng.random() < 0.5 else (n, n - k)
        a = torch.randint(0, 10 * n, (nc,), device="cuda"); b = torch.randint(0, 10 * n, (nr,), device="cuda")
        c = torch.full((n, n), 250000, dtype=torch.int32, device="cuda")
        c[:nc, :nr] = (a[:, None] - b[None, :]).abs().to(torch.int32)
    elif kind in ("g2", "g2gen"):
        a = torch.randint(0, 10 * n, (n,), device="cuda"); b = torch.randint(0, 10 * n, (n,), device="cuda")
        c = (a[:, None] - b[None, :]).abs().to(torch.int32).contiguous()
    else:
        c = torch.full((n, n), 250000, dtype=torch.int32, device="cuda")
        rr, rc = int(rng.integers(1, n)), int(rng.integers(1, n))
        c[:rr, :rc] = torch.randint(0, 50, (rr, rc), dtype=torch.int32, device="cuda")
    t0 = time.time()
    r2c, tot, dual = td.assign(c, n, want_dual=True)
    dt = time.time() - t0
    r2c_t = torch.from_numpy(np.asarray(r2c)).cuda().long()
    perm_ok = bool((torch.sort(r2c_t).values == torch.arange(n, device="cuda")).all())
    tot2 = int(c[torch.arange(n, device="cuda"), r2c_t].long().sum())
    ok = perm_ok and tot == dual == tot2
    cnt += 1
    print(("ok  " if ok else "FAIL"), kind, n, "%.1f ms" % (1e3 * dt), tot, dual, td.last_stats(), flush=True)
    bad += 0 if ok else 1
print("large stress: %d instances, %d failures" % (cnt, bad))
