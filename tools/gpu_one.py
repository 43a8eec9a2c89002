"""One instance, one line: python tools/gpu_one.py <kind> <n> [reps]  (kind: g1 g2 g3 wide; dev tool).
Env tunables (TD_*) apply; prints best-of-reps solve time and the solver statistics."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi
kind, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
td.init(0)
lib = _ffi.lib()
rng = np.random.default_rng(1)
ct = torch.empty((n, n), dtype=torch.int32, device="cuda")
if kind == "g1":
    _ffi.check(lib.td_gen_uniform(n, 1, 10, 40, 0, n, ct.data_ptr()))
elif kind == "g2":
    a = rng.integers(0, 10 * n, n).astype(np.int32); b = rng.integers(0, 10 * n, n).astype(np.int32)
    td.cost_build(a, b, None, fill=250000, threshold=-1, out=ct)
elif kind == "g3":
    a = rng.integers(0, 50, n).astype(np.int32); b = rng.integers(0, 50, max(1, int(0.363 * n))).astype(np.int32)
    td.cost_build(a, b, None, fill=250000, threshold=10, out=ct)
elif kind == "geo2":   # 2-D city grid, Manhattan distance (not in the reference: its table is a line)
    side = int(os.environ.get("GEO_SIDE", "4000"))
    ax, ay, bx, by = (torch.from_numpy(rng.integers(0, side, n).astype(np.int32)).cuda() for _ in range(4))
    ct.copy_((ax[:, None] - bx[None, :]).abs() + (ay[:, None] - by[None, :]).abs())
elif kind == "neg":     # uniform -5000..4999: 2-byte rows
    ct.copy_(torch.from_numpy(rng.integers(-5000, 5000, (n, n)).astype(np.int32)))
elif kind == "mid":     # uniform 0..40000: 2-byte rows, moderately tied
    ct.copy_(torch.randint(0, 40000, (n, n), dtype=torch.int32, device="cuda"))
else:
    ct.copy_(torch.from_numpy(rng.integers(0, 10**6, (n, n)).astype(np.int32)))
r2c = torch.empty(n, dtype=torch.int32, device="cuda")
tot = ctypes.c_int64(0); dual = ctypes.c_int64(0)
ts = []
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), ctypes.byref(dual)))
    ts.append(time.perf_counter() - t0)
st = td.last_stats()
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TD_") and k != "TD_DEBUG")
if os.environ.get("PROF"):
    _ffi.check(lib.td_profile_enable(1)); _ffi.check(lib.td_profile_reset())
    _ffi.check(lib.td_assign(n, ct.data_ptr(), r2c.data_ptr(), ctypes.byref(tot), ctypes.byref(dual)))
    for name, k in _ffi.TD_K.items():
        ms = ctypes.c_double(0); cnt = ctypes.c_int64(0)
        _ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value: print(f"   {name:10s} {ms.value:9.3f} ms {cnt.value:6d} launches", flush=True)
print(f"{kind} n={n} [{env}]: {1e3*min(ts):.3f} ms total={tot.value} cert={'ok' if tot.value == dual.value else 'FAIL'} {st}", flush=True)
