"""Dev tool: one |a-b| instance with numpy seed: python tools/gpu_one_g2.py n seed [reps]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
n, seed = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
td.init(0); lib = _ffi.lib()
rng = np.random.default_rng(seed)
a = rng.integers(0, 10 * n, n).astype(np.int32); b = rng.integers(0, 10 * n, n).astype(np.int32)
ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
td.cost_build(a, b, None, fill=250000, threshold=-1, out=ct)
expect = int(np.abs(np.sort(a).astype(np.int64) - np.sort(b).astype(np.int64)).sum())
r2c = torch.empty(n, dtype=torch.int32, device="cuda"); tot = ctypes.c_int64(0); dual = ctypes.c_int64(0)
ts = []
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), ctypes.byref(dual)))
    ts.append(time.perf_counter() - t0)
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TD_") and k != "TD_DEBUG")
print(f"g2 n={n} seed={seed} [{env}]: {1e3*min(ts):.2f} ms {'OK' if tot.value == dual.value == expect else 'FAIL'} {td.last_stats()}", flush=True)
