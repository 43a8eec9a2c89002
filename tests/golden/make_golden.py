"""Regenerates tests/golden/known_answers.json.

The reference cannot be imported here (cvxopt/GLPK are not installed: ModuleNotFoundError, an
ordinary error, nothing was denied) and it commits no solver output vector, so every vector
below is DATA transcribed from the reference's own hand-written examples plus the optimum of
that instance computed by two independent exact solvers (oracle/td_oracle.c and
scipy.optimize.linear_sum_assignment), which equals GLPK's optimum because the assignment
polytope is totally unimodular.

  pdf_table5      python.py:7 (c vector, row = cab), same numbers as glpk.mod:27-31 (transposed)
                  and julia.jl:5 + the dummy row of 100s
  procedure_py    procedure.py:32-51: S=10 stands, dist=|i-j|, 4 requests, 3 cabs; cost built by
                  id with fill n*n=16 (procedure.py:9-12)
  julia_3x4       julia.jl:5 rectangular 3x4 (rows ==1, cols <=1): optimum over real rows only
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402


def both(cost):
    c = np.asarray(cost, dtype=np.int64)
    ri, ci = linear_sum_assignment(c)
    s = int(c[ri, ci].sum())
    t, r2c, u, v = oracle.assign(c.astype(np.int32))
    assert s == t, (s, t)
    return t, bool(oracle.is_unique(c.astype(np.int32), r2c, u, v))


def main():
    out = {}
    table5 = [[5, 5, 0, 5], [1, 1, 3, 8], [9, 9, 5, 0], [100, 100, 100, 100]]
    t, uniq = both(table5)
    out["pdf_table5"] = {"source": "python.py:7 / glpk.mod:27-31 / taxi_dispatching.pdf p.3 Table 5",
                         "cost": table5, "total": t, "unique": uniq, "real_total": t - 100}
    # procedure.py:41-51
    demand = [[0, 0, 2], [1, 0, 5], [2, 3, 1], [3, 5, 1]]
    cabs = [[0, 3, 3], [1, 3, 1], [2, 0, 5]]
    n = 4
    cost = [[n * n] * n for _ in range(n)]
    for cid, cfrm, cto in cabs:
        for did, dfrm, dto in demand:
            cost[cid][did] = abs(cto - dfrm)
    t, uniq = both(cost)
    out["procedure_py"] = {"source": "procedure.py:32-51", "n_stands": 10, "demand": demand, "cabs": cabs,
                           "cost": cost, "total": t, "unique": uniq, "real_total": t - 16}
    j34 = [[5, 5, 0, 5], [1, 1, 3, 8], [9, 9, 5, 0]]
    ri, ci = linear_sum_assignment(np.asarray(j34))
    out["julia_3x4"] = {"source": "julia.jl:5", "cost": j34, "total": int(np.asarray(j34)[ri, ci].sum())}
    # counter-based generator pins (perf.jl:5 distribution): first cells + checksum
    g = oracle.gen_uniform(8, 1, 10, 40)
    out["gen_uniform_seed1_n8"] = {"source": "perf.jl:5 t = rand(10:40,n,n) as splitmix64 counter hash",
                                   "first_row": g[0].tolist(), "sum": int(g.sum())}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
