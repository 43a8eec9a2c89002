"""Dev tool: times td_assign on G1 for a few tunable settings (each in a fresh process via env)."""
import os, sys, time, ctypes, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import taxidispatcher_amd as td
    from taxidispatcher_amd import _ffi
    td.init(0); lib = _ffi.lib()
    n = int(sys.argv[2])
    cost = torch.empty((n, n), dtype=torch.int32, device="cuda")
    _ffi.check(lib.td_gen_uniform(n, 1, 10, 40, 0, n, cost.data_ptr()))
    for _ in range(3): r2c, tot = td.assign(cost, n)
    t0 = time.perf_counter()
    for _ in range(10): r2c, tot = td.assign(cost, n)
    dt = (time.perf_counter() - t0) / 10
    _ffi.check(lib.td_profile_enable(1)); _ffi.check(lib.td_profile_reset())
    r2c, tot, dual = td.assign(cost, n, want_dual=True)
    prof = {}
    for name, k in _ffi.TD_K.items():
        ms = ctypes.c_double(0); cnt = ctypes.c_int64(0)
        lib.td_profile_get(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value: prof[name] = round(ms.value, 3)
    print(json.dumps({"n": n, "ms": round(dt * 1e3, 3), "total": tot, "ok": tot == 10 * n == dual, "stats": td.last_stats(), "prof": prof}))
    sys.exit(0)
for n in [16384]:
    for env in [{}, {"TD_LDS_GRID": "2"}, {"TD_LDS_GRID": "4"}, {"TD_LDS_ROUNDS": "0"}, {"TD_LDS_ROUNDS": "2"}, {"TD_LDS_ROUNDS": "2", "TD_LDS_GRID": "2"}, {"TD_ROW_ROUNDS": "1"}]:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child", str(n)], env=e, capture_output=True, text=True)
        print(env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
