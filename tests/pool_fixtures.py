"""Readers of tests/golden/pool_n (outputs of the reference's pool_n.c, see make_pool_fixtures.py)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pool_n")


def cases():
    out = []
    for f in sorted(os.listdir(GOLD)):
        if f.endswith(".out"):
            name, k = f[:-4].split("_k")
            out.append((name, int(k)))
    return out


def load(name, k):
    """-> demand int64 [n, 5], {child: list of records [2k+1]}"""
    d = np.loadtxt(os.path.join(GOLD, name + "_demand.csv"), delimiter=",", dtype=np.int64)
    exp, cur = {}, None
    for line in open(os.path.join(GOLD, "%s_k%d.out" % (name, k))):
        if line.startswith("# child"):
            cur = int(line.split()[2].rstrip(":"))
            exp[cur] = []
        elif line.strip():
            exp[cur].append([int(x) for x in line.strip().rstrip(",").split(",")])
    return d, exp


def child_slice(n, child, children=8):
    step = n // children + 1            # pool_n.c:243-246
    a = step * child
    return a, max(a, min(n, a + step))


def merge_restatement(k, lists):
    """findpool.c:73-98,166-172 restated on the host (the comparator of td_pool_merge): concatenate in
    child order; sort by the 9th field, which findpool.c's reader fills only for k == 4 (stable);
    keep a pool iff it shares no request with an earlier kept one."""
    allp = [r for lst in lists for r in lst]
    if k == 4:
        allp = sorted(allp, key=lambda r: r[8])     # Python's sort is stable
    used, kept = set(), []
    for r in allp:
        c = r[:k]
        if any(x in used for x in c):
            continue
        used.update(c)
        kept.append(list(r))
    return kept
