"""how long the fallback of the lazy narrow copy (k_compress_rest) takes: a 1-byte family whose phase A leaves rows"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import taxidispatcher_amd as td
td.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cuda").manual_seed(3)
c = torch.randint(0, 250, (n, n), dtype=torch.int32, device="cuda", generator=g)
for _ in range(2):
    td.assign(c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    r2c, tot = td.assign(c)[:2]
torch.cuda.synchronize()
print("r250 n=%d TD_LAZY_CC=%s: %.3f ms per solve, total %d, %s" % (n, os.environ.get("TD_LAZY_CC", "1"), (time.perf_counter() - t0) / 5 * 1e3, tot, dict(td.last_stats())))
