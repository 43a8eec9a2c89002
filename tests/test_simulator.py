"""Golden trace of the reference's committed simulation (simulations/simulog_solv.txt t=0..49,
input simulations/taxi_demand.txt) replayed through the tick-loop harness.

CPU tier: path operations by the oracle -> pins the oracle's cost build, Java-variant LCM and
exact solver against output the REFERENCE itself committed (incl. `OPT count=32` of its first
GLPK call).  GPU tier: the same replay with every path operation on the MI355X."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = os.path.join(HERE, "golden")


def golden_lines():
    return [l.strip() for l in open(os.path.join(GOLD, "simulog_solv_t0_49.txt")).read().split("\n") if l.strip()]


def tick49():
    return json.load(open(os.path.join(GOLD, "tick49_instance.json")))


def test_replay_matches_reference_log_oracle_backend():
    from sim_backend import OracleBackend
    from taxidispatcher_amd import simulator
    sim = simulator.Simulator(simulator.read_demand(os.path.join(GOLD, "taxi_demand.txt.gz")), OracleBackend())
    log = [l.strip() for l in sim.run(50)]
    gold = golden_lines()
    assert len(gold) == 50 and log == gold
    assert gold[49].endswith("Sent to solver: demand=218, supply=600. ; OPT count=32")
    assert sim.m["max_model_size"] <= 1300 and sim.m["max_solver_size"] == 600


def test_replay_one_call_per_tick_harness_oracle_backend():
    """the harness path that hands a whole tick to ONE backend call (what HipTickBackend / td_tick serve on the GPU),
    driven by the oracle's pipeline: the reference's log t = 0 ... 49 again, line by line"""
    from sim_backend import OracleTickBackend
    from taxidispatcher_amd import simulator
    sim = simulator.Simulator(simulator.read_demand(os.path.join(GOLD, "taxi_demand.txt.gz")), OracleTickBackend())
    assert [l.strip() for l in sim.run(50)] == golden_lines()


def test_tick49_instance_oracle():
    from oracle import oracle
    g = tick49()
    n, cost = oracle.cost_build(g["cab_to"], g["dem_from"], None, g["fill"], g["threshold"])
    assert n == 600 and len(g["cab_to"]) == 600 and len(g["dem_from"]) == 218
    t, r, u, v = oracle.assign(cost)
    assert t == g["total"] == 142000088 == 568 * 250000 + 88
    assert oracle.count_sum(cost, r) == (g["real_total"], g["opt_count"]) == (88, 32)


@pytest.mark.gpu
def test_replay_matches_reference_log_gpu(td):
    from taxidispatcher_amd import simulator
    sim = simulator.Simulator(simulator.read_demand(os.path.join(GOLD, "taxi_demand.txt.gz")))  # HipBackend
    log = [l.strip() for l in sim.run(50)]
    assert log == golden_lines()


@pytest.mark.gpu
def test_replay_matches_reference_log_gpu_one_call_per_tick(td):
    """The reference's committed log (simulations/simulog_solv.txt:2-51) replayed with ONE td_tick call per tick
    (Simulator.java:163-208 behind one C-ABI entry point: cost build -> LCM -> device shrink -> cost build -> optimal
    assignment); every line t = 0 ... 49 must come out as the reference wrote it, incl. the ticks whose LCM ends on
    big_cost and therefore hand nothing to the solver (:188-189) and the first solver call's `OPT count=32`."""
    from taxidispatcher_amd import simulator
    sim = simulator.Simulator(simulator.read_demand(os.path.join(GOLD, "taxi_demand.txt.gz")), simulator.HipTickBackend())
    log = [l.strip() for l in sim.run(50)]
    gold = golden_lines()
    assert log == gold
    assert gold[49].endswith("Sent to solver: demand=218, supply=600. ; OPT count=32")
    assert sim.m["max_solver_size"] == 600 and sim.m["total_LCM_used"] > 0


@pytest.mark.gpu
def test_tick49_instance_gpu(td):
    from oracle import oracle
    g = tick49()
    n, cost = td.cost_build(g["cab_to"], g["dem_from"], None, fill=g["fill"], threshold=g["threshold"])
    assert np.array_equal(cost, oracle.cost_build(g["cab_to"], g["dem_from"], None, g["fill"], g["threshold"])[1])
    r2c, total, dual = td.assign(cost, want_dual=True)
    assert total == g["total"] == dual
    assert td.count_sum(n, cost, r2c) == g["real_total"]
    x = td.expand_x(n, r2c)
    opt_count = int(((x.reshape(n, n) == 1) & (cost < 250000)).sum())   # Simulator.java:378-383
    assert opt_count == g["opt_count"] == 32


@pytest.mark.gpu
def test_config5_tick_pipeline_gpu(td):
    """BASELINE config 5: a 1300-cab tick re-solved with LCM pre-reduce — cost build (n=1300) ->
    LCM down to 600 -> shrink -> cost build -> optimal assignment, against the oracle."""
    from oracle import oracle
    from taxidispatcher_amd.simulator import BIG_COST, DROP_TIME, MAX_NON_LCM
    rng = np.random.default_rng(49)
    cab_to = rng.integers(0, 50, 1300)
    dem_from = rng.integers(0, 50, 900)
    n, cost = td.cost_build(cab_to, dem_from, None, fill=BIG_COST, threshold=DROP_TIME)
    n_o, cost_o = oracle.cost_build(cab_to, dem_from, None, BIG_COST, DROP_TIME)
    assert n == 1300 and np.array_equal(cost, cost_o)
    pairs, lm = td.LCM_simulator(cost, max_non_lcm=MAX_NON_LCM)
    _, rows, cols, lm_o = oracle.lcm(cost_o, mask=BIG_COST, stop_value_on=1, stop_value=BIG_COST,
                                     stop_size=MAX_NON_LCM, sum_below=BIG_COST, java_scan=1)
    assert pairs == list(zip(rows.tolist(), cols.tolist())) and lm == lm_o and len(pairs) == 700
    keep_c = np.setdiff1d(np.arange(1300), rows)
    keep_d = np.setdiff1d(np.arange(900), cols)
    n2, cost2 = td.cost_build(cab_to[keep_c], dem_from[keep_d], None, fill=BIG_COST, threshold=DROP_TIME)
    assert n2 == 600
    r2c, total, dual = td.assign(cost2, want_dual=True)
    t_o, r_o, _, _ = oracle.assign(cost2)
    assert total == t_o == dual
    assert td.count_sum(n2, cost2, r2c) == oracle.count_sum(cost2, r_o)[0]


@pytest.mark.gpu
def test_config5_tick_in_one_call_gpu(td):
    """The same tick behind ONE C-ABI call (td_tick: the shrink runs on the device, Simulator.java:163-208,613-674)
    against the oracle's pipeline: pair list, LCM_min_val, the kept cabs / requests, the remainder's optimum; plus
    the shapes a tick loop meets: fewer cabs than requests, no LCM needed, a general distance table, an empty model."""
    from oracle import oracle
    from taxidispatcher_amd.simulator import BIG_COST, DROP_TIME, MAX_NON_LCM

    def reference(cab_to, dem_from, dist, stop):
        n_o, cost_o = oracle.cost_build(cab_to, dem_from, dist, BIG_COST, DROP_TIME)
        if 0 <= stop < n_o:
            _, rows, cols, lm = oracle.lcm(cost_o, mask=BIG_COST, stop_value_on=1, stop_value=BIG_COST, stop_size=stop,
                                           sum_below=BIG_COST, java_scan=1)
        else:
            rows, cols, lm = np.zeros(0, np.int64), np.zeros(0, np.int64), BIG_COST
        keep_c = np.setdiff1d(np.arange(len(cab_to)), rows)
        keep_d = np.setdiff1d(np.arange(len(dem_from)), cols)
        n2, cost2 = oracle.cost_build(np.asarray(cab_to)[keep_c], np.asarray(dem_from)[keep_d], dist, BIG_COST, DROP_TIME)
        tot = oracle.assign(cost2)[0] if n2 else 0
        return rows, cols, lm, keep_c, keep_d, n2, cost2, tot

    rng = np.random.default_rng(49)
    S = 50
    table = rng.integers(0, 25, (S, S)).astype(np.int32)   # a general (asymmetric) table, PDF Table 2
    cases = [(rng.integers(0, 50, 1300), rng.integers(0, 50, 900), None, MAX_NON_LCM),
             (rng.integers(0, 50, 700), rng.integers(0, 50, 1100), None, MAX_NON_LCM),
             (rng.integers(0, 50, 500), rng.integers(0, 50, 218), None, MAX_NON_LCM),     # n <= MAX_NON_LCM: no LCM
             (rng.integers(0, S, 900), rng.integers(0, S, 900), table, 250)]
    for cab_to, dem_from, dist, stop in cases:
        rows, cols, lm, keep_c, keep_d, n2, cost2, tot = reference(cab_to, dem_from, dist, stop)
        t = td.tick(cab_to, dem_from, dist, big_cost=BIG_COST, drop_time=DROP_TIME, max_non_lcm=stop)
        assert t["lcm_rows"].tolist() == rows.tolist() and t["lcm_cols"].tolist() == cols.tolist()
        if len(rows):
            assert t["lcm_min_val"] == lm
        assert t["kept_cabs"].tolist() == keep_c.tolist() and t["kept_dems"].tolist() == keep_d.tolist()
        assert t["n_rest"] == n2
        if 0 <= stop < max(len(cab_to), len(dem_from)) and lm == BIG_COST:   # Simulator.java:188-189: the LCM ran and ended on big_cost, nothing goes to the solver
            assert not t["solved"] and t["total"] == 0 and len(t["row_to_col"]) == 0
            continue
        assert t["solved"] and t["total"] == tot
        r2c = t["row_to_col"]
        assert sorted(r2c.tolist()) == list(range(n2))
        assert int(cost2[np.arange(n2), r2c].astype(np.int64).sum()) == tot
    t = td.tick(np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert t["n_rest"] == 0 and len(t["lcm_rows"]) == 0 and t["total"] == 0


@pytest.mark.gpu
def test_pool_of_two_gpu_matches_host_restatement(td):
    """f-3: td_pool2 (symmetric lowest-cost method on the pair-cost matrix) against the numpy
    restatement of Simulator.java:681-758 used by the oracle-backed replay."""
    from sim_backend import OracleBackend
    from taxidispatcher_amd import simulator
    host = simulator.Simulator(np.zeros((0, 5), np.int64), OracleBackend())
    rng = np.random.default_rng(8)
    for n in (2, 3, 17, 128, 500, 1445):
        frm = rng.integers(0, 50, n)
        to = rng.integers(0, 50, n)
        demand = [[i, int(frm[i]), int(to[i]), -1, -1, 0] for i in range(n)]
        ref = host.find_pool(demand)
        got = td.find_pool(frm, to)
        assert got == ref, n
        assert len(got) == n // 2
    # general (asymmetric) distance table
    S = 30
    dist = rng.integers(0, 40, (S, S)).astype(np.int32)
    frm = rng.integers(0, S, 200)
    to = rng.integers(0, S, 200)
    got = td.find_pool(frm, to, dist)
    d = dist.astype(np.int64)
    c1 = d[frm[:, None], frm[None, :]] + d[frm[None, :], to[:, None]] + d[to[:, None], to[None, :]]
    c2 = d[frm[:, None], frm[None, :]] + d[frm[None, :], to[None, :]] + d[to[None, :], to[:, None]]
    cost = np.minimum(c1, c2)
    used, exp = set(), []
    order = sorted(((int(cost[a, b]), a, b) for a in range(200) for b in range(200) if a != b))
    for cst, a, b in order:
        if a in used or b in used:
            continue
        used.update((a, b))
        exp.append((a, b, 1 if c1[a, b] < c2[a, b] else 0, cst))
    assert got == exp
