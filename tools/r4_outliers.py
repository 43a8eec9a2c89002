"""33 - 52 ms calls among ~0.3 ms ones in tools/gpu_stress.py's log.  The same seeded sequence of small
instances twice in one process, every call timed: are the outliers a property of the instance or of the first time the
process meets a size / a kernel variant (grow-only workspace reallocation, code-object load, hipFuncSetAttribute)?
usage: python tools/r4_outliers.py [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import taxidispatcher_amd as td
from test_gpu_parity import make_instance
td.init(0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
def sequence():
    rng = np.random.default_rng(7)
    out = []
    for _ in range(count):
        kind = ["g1", "g4", "g2", "g3", "wide", "neg", "const"][int(rng.integers(0, 7))]
        n = int(rng.integers(2, 1400)) if kind not in ("g2", "wide") else int(rng.integers(2, 700))
        out.append((kind, n, make_instance(kind, n, rng)))
    return out
seq = sequence()
res = []
for p in range(3):
    ts = []
    for kind, n, c in seq:
        t0 = time.perf_counter()
        td.assign(c)
        ts.append(1e3 * (time.perf_counter() - t0))
    res.append(ts)
for p, ts in enumerate(res):
    order = np.argsort(ts)[::-1][:6]
    print("pass %d: median %.3f ms, p99 %.3f, max %.3f; slowest: %s" % (p, float(np.median(ts)), float(np.percentile(ts, 99)), max(ts),
          ", ".join("%s n=%d %.2f ms" % (seq[i][0], seq[i][1], ts[i]) for i in order)))
slow0 = [i for i in range(count) if res[0][i] > 5.0]
print("calls above 5 ms in pass 0: %d; the same instances in pass 2: %s" % (len(slow0), ", ".join("%.2f" % res[2][i] for i in slow0[:12])))
