"""Dev tool: unbalanced |a-b| model: python tools/gpu_one_unbal.py n_cabs n_requests [seed]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
nc, nr = int(sys.argv[1]), int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = max(nc, nr)
td.init(0); lib = _ffi.lib()
rng = np.random.default_rng(seed)
a = rng.integers(0, 10 * n, nc).astype(np.int32); b = rng.integers(0, 10 * n, nr).astype(np.int32)
ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
td.cost_build(a, b, None, fill=250000, threshold=-1, out=ct)
r2c = torch.empty(n, dtype=torch.int32, device="cuda"); tot = ctypes.c_int64(0); dual = ctypes.c_int64(0)
for line in (1, 0):
    td.set_line_metric(bool(line))
    ts = []
    for rep in range(3 if line else 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), ctypes.byref(dual)))
        ts.append(time.perf_counter() - t0)
    st = td.last_stats()
    print(f"unbal {nc}x{nr} line={line}: {1e3*min(ts):.3f} ms total={tot.value} cert={'ok' if tot.value == dual.value else 'FAIL'} line_metric={st['line_metric']} dummies={st['line_dummies']} transposed={st['transposed']}", flush=True)
    if os.environ.get("SKIP_GENERAL"): break
