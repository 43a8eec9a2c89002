"""Quick on-GPU sanity + timing script (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
from oracle import oracle

td.init(0)
lib = _ffi.lib()
rng = np.random.default_rng(0)
ok = True
for n in [64, 600, 1000, 1300, 2048]:
    for kind in ["g1", "g4", "g2", "g3", "wide"]:
        if kind == "g1": c = rng.integers(10, 41, (n, n))
        elif kind == "g4": c = rng.integers(1, 40, (n, n))
        elif kind == "wide": c = rng.integers(0, 1000000, (n, n))
        else:
            S = 50 if kind == "g3" else 10 * n
            a = rng.integers(0, S, n); b = rng.integers(0, S, n)
            c = np.abs(a[:, None] - b[None, :])
            if kind == "g3":
                nd = max(1, int(n * 0.363))
                c[c >= 10] = 250000; c[:, nd:] = 250000
        c = c.astype(np.int32)
        t0 = time.time()
        r2c, tot, dual = td.assign(c, want_dual=True)
        dt = time.time() - t0
        ref = oracle.assign(c)[0]
        st = td.last_stats()
        good = tot == ref and dual == tot and sorted(r2c.tolist()) == list(range(n)) and int(c[np.arange(n), r2c].sum()) == tot
        ok &= good
        print(f"{kind:5s} n={n:5d} total={tot} ref={ref} dual={dual} {'OK' if good else 'FAIL'} {dt*1e3:.2f}ms {st}", flush=True)
print("ALL OK" if ok else "SOME FAILED")

# timing at scale, device resident
for n in [1000, 4096, 16384]:
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(lib.td_gen_uniform(n, 1, 10, 40, 0, n, cost.data_ptr()))
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.time()
        _ffi.check(lib.td_gen_uniform(n, 1, 10, 40, 0, n, cost.data_ptr()))
        t1 = time.time()
        r2c, tot, dual = td.assign(cost, n, want_dual=True)
        t2 = time.time()
        print(f"G1 n={n}: gen {1e3*(t1-t0):.3f}ms assign {1e3*(t2-t1):.3f}ms total={tot} dual={dual} {td.last_stats()}", flush=True)
    _ffi.check(lib.td_profile_enable(1)); _ffi.check(lib.td_profile_reset())
    r2c, tot = td.assign(cost, n)
    import ctypes
    for name, k in _ffi.TD_K.items():
        ms = ctypes.c_double(0); cnt = ctypes.c_int64(0)
        lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: print(f"   {name:10s} {ms.value:8.3f} ms / {cnt.value} launches")
    _ffi.check(lib.td_profile_enable(0))
    if n <= 4096:
        ref = oracle.assign(cost.cpu().numpy())[0]
        print("   oracle total", ref, "OK" if ref == tot else "FAIL")
