"""dev: small thresholded models that take milliseconds (tools/r4_outliers.py: g3 n = 60: 7.75 ms, repeatably)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import taxidispatcher_amd as td
from test_gpu_parity import make_instance
td.init(0)
rng = np.random.default_rng(3)
rows = []
kind = sys.argv[1] if len(sys.argv) > 1 else "g3"
for n in list(range(40, 100, 5)) + [128, 200, 300, 450, 600, 800, 1000, 1300]:
    for rep in range(3):
        c = make_instance(kind, n, rng)
        td.assign(c)
        t0 = time.perf_counter(); r2c, tot, dual = td.assign(c, want_dual=True); dt = 1e3 * (time.perf_counter() - t0)
        st = dict(td.last_stats())
        rows.append((dt, n, st))
        if dt > 2.0 and os.environ.get("TD_DEBUG"):
            print("slow: n=%d %.2f ms %s" % (n, dt, st), flush=True)
rows.sort(key=lambda r: -r[0])
for dt, n, st in rows[:10]:
    print("%.2f ms n=%d rounds %d warm %d free %d steps %d bpc %d transposed %d" % (dt, n, st["bid_rounds"], st["warm_rounds"], st["sap_free_rows"], st["sap_steps"], st["bytes_per_cell"], st["transposed"]))
