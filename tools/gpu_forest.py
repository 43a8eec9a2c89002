"""Dev tool: the sparse-core + forest path against closed forms / the oracle.
python tools/gpu_forest.py [sizes...]   (|a-b| geometry and wide uniform costs)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
sizes = [int(x) for x in sys.argv[1:]] or [300, 1000, 2048, 4096, 8192, 16384]
td.init(0)
lib = _ffi.lib()
bad = 0
for kind in ("g2", "wide"):
    for n in sizes:
        for seed in (1, 2):
            rng = np.random.default_rng(seed)
            ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
            expect = None
            if kind == "g2":
                a = rng.integers(0, 10 * n, n).astype(np.int32); b = rng.integers(0, 10 * n, n).astype(np.int32)
                td.cost_build(a, b, None, fill=250000, threshold=-1, out=ct)
                expect = int(np.abs(np.sort(a).astype(np.int64) - np.sort(b).astype(np.int64)).sum())
            else:
                if n > 8192: continue
                ct.copy_(torch.from_numpy(rng.integers(0, 10**6, (n, n)).astype(np.int32)))
            r2c = torch.empty(n, dtype=torch.int32, device="cuda")
            tot = ctypes.c_int64(0); dual = ctypes.c_int64(0)
            ts = []
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), ctypes.byref(dual)))
                ts.append(time.perf_counter() - t0)
            st = td.last_stats()
            perm = sorted(r2c.cpu().tolist()) == list(range(n))
            chk = int(ct[torch.arange(n, device="cuda"), r2c.long()].sum().item())
            ok = perm and chk == tot.value == dual.value and (expect is None or expect == tot.value)
            bad += 0 if ok else 1
            print(f"{kind} n={n} seed={seed}: {1e3*min(ts):.2f} ms total={tot.value} dual={dual.value} expect={expect} perm={perm} {'OK' if ok else 'FAIL'} stats={st}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
