"""Rectangular (dummy-column) instances: td_assign time by size; run with TD_SHAPE=0 / 1 (dev tool)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
td.init(0)
lib = _ffi.lib()
rng = np.random.default_rng(1)
for n in [600, 1300, 2048, 4096, 8192, 16384]:
    for frac in (0.33, 0.8):
        nd = max(1, int(frac * n))
        a = rng.integers(0, 50, n).astype(np.int32)
        b = rng.integers(0, 50, nd).astype(np.int32)
        ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
        td.cost_build(a, b, None, fill=250000, threshold=10, out=ct)
        r2c = torch.empty(n, dtype=torch.int32, device="cuda")
        tot = ctypes.c_int64(0)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), None))
            ts.append(time.perf_counter() - t0)
        st = td.last_stats()
        print(f"n={n:6d} real_cols={nd:6d}: {1e3*min(ts[1:]):8.3f} ms total={tot.value} T={st['transposed']} rounds={st['bid_rounds']} "
              f"sap_rows={st['sap_free_rows']} steps={st['sap_steps']} par={st['parallel_sap_rows']} bpc={st['bytes_per_cell']}", flush=True)
