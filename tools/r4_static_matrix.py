"""the compress pass with and without a matrix write in front of it (is the write-back of the generated matrix part of its time?)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
from taxidispatcher_amd import _ffi as ffi
td.init(0)
lib = ffi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cuda").manual_seed(1)
c = torch.randint(10, 41, (n, n), dtype=torch.int32, device="cuda", generator=g)
other = torch.empty((n, n), dtype=torch.int32, device="cuda")


def prof(step, reps=10):
    ffi.check(lib.td_profile_enable(1))
    ffi.check(lib.td_profile_reset())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out = {}
    for name, k in ffi.TD_K.items():
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        ffi.check(lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            out[name] = round(1e3 * ms.value / cnt.value, 1)
    ffi.check(lib.td_profile_enable(0))
    return round(dt * 1e3, 4), out


for _ in range(3):
    td.assign(c)
print("static matrix, nothing written between solves :", prof(lambda: td.assign(c)))
print("1 GiB written to ANOTHER buffer before each    :", prof(lambda: (other.fill_(7), td.assign(c))))
print("the matrix itself rewritten (copy) before each :", prof(lambda: (c.copy_(c.clone()) if False else c.add_(0), td.assign(c))))
