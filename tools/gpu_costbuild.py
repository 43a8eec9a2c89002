"""Times td_cost_build (a-2) and td_lcm row scan at N=16384 on the device (dev tool)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
td.init(0); lib = _ffi.lib()
n = 16384
rng = np.random.default_rng(0)
cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
for name, S, dist, thr in [("analytic |a-b|, S=10n", 10 * n, None, -1), ("analytic, S=50, threshold 10", 50, None, 10),
                           ("LDS table S=50", 50, rng.integers(0, 60, (50, 50)).astype(np.int32), 10),
                           ("global table S=1000", 1000, rng.integers(0, 60, (1000, 1000)).astype(np.int32), -1)]:
    a = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, S, n).astype(np.int32)).cuda()
    dt = None if dist is None else torch.from_numpy(dist).cuda()
    args = (a.data_ptr(), None, n, b.data_ptr(), None, n, None if dt is None else dt.data_ptr(), 0 if dt is None else S, 250000, thr, 0, cost.data_ptr())
    for _ in range(3): _ffi.check(lib.td_cost_build(*args))
    _ffi.check(lib.td_profile_enable(1)); _ffi.check(lib.td_profile_reset())
    for _ in range(10): _ffi.check(lib.td_cost_build(*args))
    ms = ctypes.c_double(0); cnt = ctypes.c_int64(0)
    lib.td_profile_get(_ffi.TD_K["cost_build"], ctypes.byref(ms), ctypes.byref(cnt))
    _ffi.check(lib.td_profile_enable(0))
    us = 1e3 * ms.value / cnt.value
    print(f"cost_build {name:32s}: {us:8.1f} us  -> {4.0*n*n/us/1e3:7.1f} GB/s written ({4.0*n*n/us/1e3/80:.1f}% of 8 TB/s)")
