#!/usr/bin/env python3
"""Makes tests/golden/pool_n/*: demand files and the outputs of the REFERENCE's pool finder for them.

Run in the build container only (needs /root/reference): `make -C oracle ref` compiles
/root/reference/pool_n.c where it lies into oracle/_ref/pool_n (git-ignored); this script runs that
binary the way findpool.c:138-141 does — once per child 0..7, `pool_n <pool-size> <child> <demand
file> <records> <out file>` — and stores demand + outputs as data fixtures.  Only data is
committed, no reference source.  Instances stay below pool_n.c's MAX_ARR = 10 000 happy plans per
child (it has no overflow check) and MAX_DEMAND = 2000 / MAX_STAND = 51."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "pool_n")
BIN = os.path.join(ROOT, "oracle", "_ref", "pool_n")


def demand(seed, n, max_wait, losses):
    rng = np.random.default_rng(seed)
    frm = rng.integers(0, 50, n)
    diff = rng.integers(1, 9, n) * rng.choice([-1, 1], n)      # short trips, like gendemand.py:10
    to = np.clip(frm + diff, 0, 49)
    to = np.where(to == frm, np.where(frm > 0, frm - 1, 1), to)
    wait = rng.integers(0, max_wait + 1, n)
    loss = rng.choice(losses, n)
    return np.stack([np.arange(n), frm, to, wait, loss], 1)


CASES = [  # name, seed, n, max_wait, loss choices, pool sizes
    ("a40", 1, 40, 6, [10, 30, 50, 90], (2, 3, 4)),
    ("b120", 2, 120, 4, [1, 20, 50, 70], (2, 3, 4)),
    ("c300", 3, 300, 3, [1, 10, 30], (2, 3)),
    ("d200", 4, 200, 2, [1, 10, 25], (4,)),
    ("f400", 6, 400, 1, [1, 5], (3,)),
]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(OUT, exist_ok=True)
    for name, seed, n, mw, losses, sizes in CASES:
        d = demand(seed, n, mw, losses)
        dpath = os.path.join(OUT, "%s_demand.csv" % name)
        np.savetxt(dpath, d, fmt="%d", delimiter=",")
        for k in sizes:
            lines = []
            with tempfile.TemporaryDirectory() as tmp:
                for child in range(8):
                    subprocess.check_call([BIN, str(k), str(child), dpath, str(n), "out.csv"], cwd=tmp)
                    txt = open(os.path.join(tmp, "out.csv")).read()
                    rows = [r for r in txt.splitlines() if r.strip()]
                    if len(rows) >= 9000:
                        sys.exit("case %s k=%d child %d: too close to MAX_ARR" % (name, k, child))
                    lines.append("# child %d: %d plans" % (child, len(rows)))
                    lines += rows
            with open(os.path.join(OUT, "%s_k%d.out" % (name, k)), "w") as f:
                f.write("\n".join(lines) + "\n")
            print(name, k, sum(1 for l in lines if not l.startswith("#")), "plans")


if __name__ == "__main__":
    main()
