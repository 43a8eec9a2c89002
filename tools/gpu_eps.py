"""Measures the literal eps-scaling auction (TD_SOLVER=eps) against the default hybrid on the GPU."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import taxidispatcher_amd as td
    from taxidispatcher_amd import _ffi
    from oracle import oracle
    td.init(0)
    kind, n = sys.argv[2], int(sys.argv[3])
    rng = np.random.default_rng(1)
    if kind == "g1": c = rng.integers(10, 41, (n, n))
    elif kind == "g4": c = rng.integers(1, 40, (n, n))
    elif kind == "g2":
        a = rng.integers(0, 10 * n, n); b = rng.integers(0, 10 * n, n); c = np.abs(a[:, None] - b[None, :])
    elif kind == "wide": c = rng.integers(0, 1000000, (n, n))
    else:
        a = rng.integers(0, 50, n); b = rng.integers(0, 50, n)
        c = np.abs(a[:, None] - b[None, :]); c[c >= 10] = 250000; c[:, int(.363 * n):] = 250000
    c = c.astype(np.int32)
    ct = torch.from_numpy(c).cuda()
    td.assign(ct, n)
    t0 = time.perf_counter(); r2c, tot = td.assign(ct, n); dt = time.perf_counter() - t0
    ref = oracle.assign(c)[0] if n <= 4096 else 10 * n
    st = td.last_stats()
    print(json.dumps({"kind": kind, "n": n, "ms": round(1e3 * dt, 3), "ok": bool(tot == ref), "rounds": int(st["bid_rounds"])}))
    sys.exit(0)
cases = [("g1", 1000), ("g1", 4096), ("g1", 16384), ("g4", 100), ("g3", 600), ("g2", 400), ("wide", 1000)]
for kind, n in cases:
    row = []
    for env in [{}, {"TD_SOLVER": "eps", "TD_EPS0_MULT": "0"}, {"TD_SOLVER": "eps", "TD_EPS0_MULT": "4"}, {"TD_SOLVER": "eps", "TD_EPS0_MULT": "32"}]:
        e = dict(os.environ); e.update(env)
        try:
            out = subprocess.run([sys.executable, __file__, "child", kind, str(n)], env=e, capture_output=True, text=True, timeout=240)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            row.append((env.get("TD_EPS0_MULT", "hybrid"), json.loads(line[-1]) if line else out.stderr[-200:]))
        except subprocess.TimeoutExpired:
            row.append((env.get("TD_EPS0_MULT", "hybrid"), "timeout 240 s"))
    print(kind, n, row, flush=True)
