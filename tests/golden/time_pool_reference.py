#!/usr/bin/env python3
"""Times the REFERENCE's pool finder (oracle/_ref/pool_n = /root/reference/pool_n.c compiled where it lies by
`make -C oracle ref`) on the demand files of bench.py --workload pool, the way findpool.c:138-169 runs it: 8 children,
`pool_n <pool-size> <child> <demand file> <records> <out file>`.  findpool.c starts them concurrently (`start /B`); here
they run one after the other and the SUM is recorded next to the longest child (the wall time of a perfectly parallel
8-way run).  Build container only: the binary never travels.  Writes tests/golden/pool_n/reference_timing.json (data)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "oracle", "_ref", "pool_n")


def main():
    import bench
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {}
    for name, ks in bench.POOL_FIXTURE_CASES:   # the bench sizes (n = 600 / 2000) overflow pool_n.c's MAX_ARR: see bench.POOL_CASES
        d = np.loadtxt(os.path.join(HERE, "pool_n", name + "_demand.csv"), delimiter=",", dtype=np.int64)
        n = int(d.shape[0])
        with tempfile.TemporaryDirectory() as tmp:
            dpath = os.path.join(tmp, "demand.csv")
            np.savetxt(dpath, d, fmt="%d", delimiter=",")
            for k in ks:
                best_sum, best_max, plans = None, None, 0
                for rep in range(3):
                    times, plans = [], 0
                    for child in range(8):
                        t0 = time.perf_counter()
                        subprocess.check_call([BIN, str(k), str(child), dpath, str(n), "out.csv"], cwd=tmp)
                        times.append(time.perf_counter() - t0)
                        rows = [r for r in open(os.path.join(tmp, "out.csv")).read().splitlines() if r.strip()]
                        if len(rows) >= 9000:
                            sys.exit("%s k=%d child %d: %d plans, too close to pool_n.c's MAX_ARR" % (name, k, child, len(rows)))
                        plans += len(rows)
                    if best_sum is None or sum(times) < best_sum:
                        best_sum, best_max = sum(times), max(times)
                out["%s_k%d" % (name, k)] = {"n": n, "k": k, "wall_ms": 1e3 * best_sum, "longest_child_ms": 1e3 * best_max,
                                            "plans": plans, "note": "8 children run sequentially, best of 3; includes process start and file I/O like findpool.c's children"}
                print(name, k, out["%s_k%d" % (name, k)])
    out["_host"] = {"cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
                    "binary": "gcc -O2 /root/reference/pool_n.c (oracle/Makefile: ref)"}
    json.dump(out, open(os.path.join(HERE, "pool_n", "reference_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
