"""Builds the simulator golden fixtures (data only — no reference source is copied):

  taxi_demand.txt.gz          the reference's committed input simulations/taxi_demand.txt
                              (42 161 rows `(id,from,to,time,at)`), gzipped
  simulog_solv_t0_49.txt      lines t:0 .. t:49 of the reference's committed run log
                              simulations/simulog_solv.txt — the tie-invariant part: the first solver
                              call is at t=49 (`OPT count=32` is the same for every optimum), and WHICH
                              of the tied optima the solver returned steers the state from t=50 on
                              (SURVEY.md says 0..51 reproduced with scipy's tie choice; with another
                              exact solver t=50/51 differ by 2-4 cabs, so they are not pinned)
  tick49_instance.json        the 600 x 600 model the reference sent to its solver at t=49
                              (600 cabs x 218 requests), captured from the replay, with the optimum
                              computed by two exact solvers and the tie-invariant OPT count

Needs /root/reference (this container only).  Run: python tests/golden/make_sim_fixture.py
"""
import gzip
import json
import os
import shutil
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference/simulations"

from oracle import oracle  # noqa: E402
from sim_backend import OracleBackend  # noqa: E402
from taxidispatcher_amd import simulator  # noqa: E402


def main():
    with open(os.path.join(REF, "taxi_demand.txt"), "rb") as f, \
            gzip.GzipFile(os.path.join(HERE, "taxi_demand.txt.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    lines = [l for l in open(os.path.join(REF, "simulog_solv.txt")).read().split("\n") if l.startswith("t:")]
    golden = [l for l in lines if int(l[2:l.index(".")]) <= 49]
    open(os.path.join(HERE, "simulog_solv_t0_49.txt"), "w").write("\n".join(golden) + "\n")

    captured = {}

    def grab(t, supply, demand, cost):
        if t == 49:
            captured.update(t=t, cab_to=[s[2] for s in supply], dem_from=[d[1] for d in demand],
                            cost=np.asarray(cost).copy())

    sim = simulator.Simulator(simulator.read_demand(os.path.join(HERE, "taxi_demand.txt.gz")), OracleBackend(),
                              on_solver_instance=grab)
    log = sim.run(50)
    bad = [(a, b) for a, b in zip(log, golden) if a.strip() != b.strip()]
    print("replayed %d lines, %d differ" % (len(log), len(bad)))
    for a, b in bad[:5]:
        print(" ours:", a, "\n  ref:", b)
    assert not bad and len(log) == len(golden)
    cost = captured["cost"]
    t, r, u, v = oracle.assign(cost)
    ri, ci = linear_sum_assignment(cost.astype(np.int64))
    assert t == int(cost[ri, ci].sum())
    s, k = oracle.count_sum(cost, r)
    out = {"source": "Simulator.java replay on simulations/taxi_demand.txt; simulog_solv.txt:51", "t": 49,
           "n": int(cost.shape[0]), "cab_to": captured["cab_to"], "dem_from": captured["dem_from"],
           "fill": 250000, "threshold": 10, "total": t, "real_total": s, "opt_count": k,
           "unique": bool(oracle.is_unique(cost, r, u, v))}
    json.dump(out, open(os.path.join(HERE, "tick49_instance.json"), "w"))
    print({k: v for k, v in out.items() if k not in ("cab_to", "dem_from")})


if __name__ == "__main__":
    main()
