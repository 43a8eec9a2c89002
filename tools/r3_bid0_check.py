"""round 0 out of the compress pass (TD_BID0) against round 0 as its own k_bid launch: same row_to_col, bit for bit"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import taxidispatcher_amd as td
td.init(0)
out = {}
rng = np.random.default_rng(5)
for name, n in [("g1", 8200), ("g1", 16384), ("g1", 20000), ("g4", 12288), ("padrows", 9000), ("padcols", 9000), ("padboth", 10240), ("wide8", 8400), ("uniq", 8400), ("uniq", 16640)]:
    if name == "g1":
        c = rng.integers(10, 41, (n, n)).astype(np.int32)
    elif name == "g4":
        c = rng.integers(0, 4, (n, n)).astype(np.int32)
    elif name == "wide8":
        c = rng.integers(0, 250, (n, n)).astype(np.int32)
    elif name == "uniq":   # every row has ONE cell at its minimum: round 0 raises the price by second - first
        c = rng.integers(3, 200, (n, n)).astype(np.int32)
        c[np.arange(n), rng.integers(0, n, n)] = rng.integers(0, 3, n)
        c[::7] += 1000
    else:
        c = rng.integers(10, 41, (n, n)).astype(np.int32)
        if name in ("padrows", "padboth"):
            c[rng.permutation(n)[:n // 5]] = 250
        if name in ("padcols", "padboth"):
            c[:, rng.permutation(n)[:n // 3]] = 250
    r2c, tot, dual = td.assign(c, want_dual=True)
    assert tot == dual and sorted(r2c.tolist()) == list(range(n))
    out["%%s_%%d" %% (name, n)] = [int(tot), hashlib.sha1(r2c.tobytes()).hexdigest(), dict(td.last_stats())["bid_rounds"], dict(td.last_stats())["transposed"]]
print(json.dumps(out))
''' % ROOT
res = {}
for mode in ("1", "0"):
    env = dict(os.environ, TD_BID0=mode)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)
    res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
bad = [k for k in res["1"] if res["1"][k] != res["0"][k]]
for k in res["1"]:
    print(k, res["1"][k], "==" if res["1"][k] == res["0"][k] else "!= " + str(res["0"][k]))
print("bid0 check:", "IDENTICAL" if not bad else "DIFFERENT %s" % bad)
sys.exit(1 if bad else 0)
