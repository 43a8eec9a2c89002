"""Tick-loop harness around the assignment path — the caller side of BASELINE config 5.

A from-scratch host harness with the semantics of `Simulator.java` (file:line citations below)
so that the committed run log `simulations/simulog_solv.txt` can be replayed: per tick it builds
the cost matrix, cuts the model with LCM when it is larger than MAX_NON_LCM, applies the pairs,
re-builds the remainder and solves it to optimality.  The three path operations go through a
backend object; the product backend is `HipBackend` (the MI355X library), which also runs the
pool-of-two pre-reduce on the GPU (td_pool2, SURVEY f-3).  World bookkeeping (cab movement,
request intake/drop) is plain host code — it is not on the hot path and is
kept only as far as the golden trace needs it; the numpy pool finder below is the restatement
the oracle-backed replay uses.

Bug-compatible details that the trace depends on are listed in SURVEY.md Appendix A.
"""
import gzip

import numpy as np

HOURS = 2
N_STANDS = 50          # Simulator.java:110
DROP_TIME = 10         # :111
MAX_NON_LCM = 600      # :112
N_CABS = 1300          # :113
BIG_COST = 250000      # :114
CLNT_A_ENDS, CLNT_B_ENDS = 0, 1   # :97-98


class HipBackend:
    """cost build / LCM / optimal assignment on the GPU through the C ABI."""

    def __init__(self):
        from . import dispatch
        self.d = dispatch

    def calculate_cost(self, cab_to, dem_from):
        return self.d.cost_build(cab_to, dem_from, None, fill=BIG_COST, threshold=DROP_TIME)[1]

    def lcm(self, cost):
        return self.d.LCM_simulator(cost, max_non_lcm=MAX_NON_LCM, big_cost=BIG_COST)

    def solve(self, cost):
        return self.d.assign(cost)[0]

    def find_pool(self, frm, to):
        return self.d.find_pool(frm, to, None)


class HipTickBackend(HipBackend):
    """The same path with ONE C-ABI call per tick (td_tick = Simulator.java:163-208 behind one entry point: cost build ->
    LCM -> removal of the matched cabs / requests on the device -> cost build -> optimal assignment); the harness
    only applies the pairs and the solution to its world."""

    def tick(self, cab_to, dem_from):
        return self.d.tick(np.asarray(cab_to, np.int32), np.asarray(dem_from, np.int32), None, big_cost=BIG_COST,
                           drop_time=DROP_TIME, max_non_lcm=MAX_NON_LCM)


class _RealCells:
    """cost[s][c] of a model that stayed on the device, as far as analyzeSolution reads it (Simulator.java:508-511: a
    cell is dist[cab.to][request.from] when that is below DROP_TIME, else big_cost)"""

    def __init__(self, supply, demand):
        self.supply, self.demand = supply, demand

    def __getitem__(self, s):
        to = self.supply[s][2]
        return _RealRow(to, self.demand)


class _RealRow:
    def __init__(self, to, demand):
        self.to, self.demand = to, demand

    def __getitem__(self, c):
        d = abs(self.to - self.demand[c][1])
        return d if d < DROP_TIME else BIG_COST


def read_demand(path):
    """taxi_demand.txt rows `(id,from,to,time,at)` (Simulator.java:280-304, gendemand.py:19)."""
    op = gzip.open if str(path).endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([int(v) for v in line[1:-1].split(",")])
    return np.asarray(rows, dtype=np.int64).reshape(-1, 5)


def cheat_a_bit(frm, cost):
    """Simulator.java:469-474"""
    if frm + cost >= N_STANDS:
        return 0 if frm - cost < 0 else frm - cost
    return frm + cost


class Simulator:
    def __init__(self, demand_rows, backend=None, n_cabs=N_CABS, on_solver_instance=None):
        self.be = backend if backend is not None else HipBackend()
        self.on_solver_instance = on_solver_instance
        d = np.asarray(demand_rows, dtype=np.int64)
        self.d_id, self.d_from, self.d_to, self.d_time, self.d_at = (d[:, k].copy() for k in range(5))
        nd = d.shape[0]
        self.d_cab = np.full(nd, -1, np.int64)          # cab_assigned (-2 = dropped)
        self.d_pick = np.full(nd, -1, np.int64)
        self.d_pool_id = np.full(nd, -1, np.int64)
        self.d_pool_plan = np.full(nd, -1, np.int64)
        self.d_pool_cost = np.zeros(nd, np.int64)
        self.id2idx = {int(v): i for i, v in enumerate(self.d_id)}
        # initSupply, Simulator.java:565-573
        self.n_cabs = n_cabs
        self.c_from = np.arange(n_cabs, dtype=np.int64) % N_STANDS
        self.c_to = self.c_from.copy()
        self.c_clnt = np.full(n_cabs, -1, np.int64)
        self.c_onboard = np.zeros(n_cabs, np.int64)
        self.c_start = np.full(n_cabs, -1, np.int64)
        self.m = dict(total_dropped=0, total_pickup_time=0, total_pickup_numb=0, total_LCM_used=0,
                      max_model_size=0, max_solver_size=0, max_POOL_MEM_size=0, max_POOL_size=0,
                      total_second_passengers=0)
        self.log = []

    # ---- Simulator.java:220-254
    def check_if_cab_at_destination(self, t):
        moving = np.nonzero((self.c_from != self.c_to) &
                            (np.abs(self.c_from - self.c_to) == t - self.c_start))[0]
        for c in moving:
            if self.c_onboard[c] == 0:
                d = self.id2idx.get(int(self.c_clnt[c]))
                if d is None:
                    continue
                self.d_cab[d] = c
                self.d_pick[d] = t
                self.m["total_pickup_numb"] += 1
                self.c_from[c] = self.d_from[d]
                self.c_to[c] = self.d_to[d] if self.d_pool_id[d] == -1 else cheat_a_bit(int(self.d_from[d]),
                                                                                       int(self.d_pool_cost[d]))
                self.c_clnt[c] = self.d_id[d]
                self.c_onboard[c] = 1
                self.c_start[c] = t
            else:
                self.c_from[c] = self.c_to[c]
                self.c_clnt[c] = -1
                self.c_onboard[c] = 0
                self.c_start[c] = -1

    @staticmethod
    def _near(stand_flags):
        """near[s] = any flagged stand within distance < DROP_TIME of s"""
        cs = np.concatenate([[0], np.cumsum(stand_flags.astype(np.int64))])
        s = np.arange(N_STANDS)
        lo = np.maximum(0, s - (DROP_TIME - 1))
        hi = np.minimum(N_STANDS - 1, s + (DROP_TIME - 1))
        return (cs[hi + 1] - cs[lo]) > 0

    # ---- Simulator.java:329-355
    def create_temp_demand(self, t):
        cand = np.nonzero((self.d_cab == -1) & (t >= self.d_at))[0]
        drop = cand[t - self.d_at[cand] >= DROP_TIME]
        self.d_cab[drop] = -2
        self.m["total_dropped"] += int(drop.size)
        keep = cand[t - self.d_at[cand] < DROP_TIME]
        free_to = np.zeros(N_STANDS, bool)
        free_to[self.c_to[self.c_clnt == -1]] = True
        near = self._near(free_to)
        keep = keep[near[self.d_from[keep]]]
        # TempDemand: id, from, to, pool_clnt_id, pool_plan, pool_cost
        return [[int(self.d_id[d]), int(self.d_from[d]), int(self.d_to[d]), -1, -1, 0] for d in keep]

    # ---- Simulator.java:358-372
    def create_temp_supply(self):
        has_req = np.zeros(N_STANDS, bool)
        has_req[self.d_from[self.d_cab == -1]] = True     # ANY unassigned request, no time check
        near = self._near(has_req)
        cabs = np.nonzero((self.c_from == self.c_to) & (self.c_clnt == -1) & near[self.c_to])[0]
        return [[int(c), int(self.c_from[c]), int(self.c_to[c])] for c in cabs]   # Supply: id, from, to

    # ---- Simulator.java:681-758 (every ordered pair is admitted: plan1 = plan2 = true at :691)
    def find_pool(self, temp_demand):
        n = len(temp_demand)
        if n < 2:
            return []
        frm = np.array([r[1] for r in temp_demand], np.int64)
        to = np.array([r[2] for r in temp_demand], np.int64)
        self.m["max_POOL_MEM_size"] = max(self.m["max_POOL_MEM_size"], n * (n - 1))
        # pool of two on the backend (td_pool2 on the GPU; the tests' comparator restates it on the host)
        out = self.be.find_pool(frm, to)
        self.m["max_POOL_size"] = max(self.m["max_POOL_size"], len(out))
        return out

    # ---- Simulator.java:760-784
    @staticmethod
    def analyze_pool(pool, temp_demand):
        is_b = {p[1] for p in pool}
        a_info = {}
        for a, b, plan, cost in pool:
            a_info.setdefault(a, (b, plan, cost))
        out = []
        for d, r in enumerate(temp_demand):
            if d in is_b:
                continue
            cust = [r[0], r[1], r[2], -1, -1, 0]
            if d in a_info:
                b, plan, cost = a_info[d]
                cust[3], cust[4], cust[5] = temp_demand[b][0], plan, cost
            out.append(cust)
        return out

    def calculate_cost(self, temp_demand, temp_supply):
        n = max(len(temp_demand), len(temp_supply))
        if n == 0:
            return np.zeros((0, 0), np.int32)
        return np.asarray(self.be.calculate_cost(np.array([s[2] for s in temp_supply], np.int32),
                                                 np.array([d[1] for d in temp_demand], np.int32)))

    # ---- Simulator.java:424-490
    def _assign_pooled(self, customer, cab):
        d2 = self.id2idx.get(int(customer))
        if d2 is not None:
            self.d_cab[d2] = cab
            self.m["total_second_passengers"] += 1

    def _assign_to_cab_and_go(self, t, c, cust):
        self.c_from[c] = cust[1]
        self.c_to[c] = cust[2] if cust[3] == -1 else cheat_a_bit(int(self.c_from[c]), cust[5])
        self.c_clnt[c] = cust[0]
        self.c_onboard[c] = 1
        self.c_start[c] = t
        self.m["total_pickup_numb"] += 1

    def _go_to_pickup(self, t, c, cust):
        self.c_to[c] = cust[1]
        self.c_clnt[c] = cust[0]
        self.c_onboard[c] = 0
        self.c_start[c] = t
        self.m["total_pickup_time"] += abs(int(self.c_from[c]) - int(self.c_to[c]))

    def _dispatch(self, t, supply, cust):
        c = supply[0]   # cab id == cab index
        if supply[2] == cust[1]:
            self._assign_to_cab_and_go(t, c, cust)
        elif abs(supply[2] - cust[1]) < DROP_TIME:
            self._go_to_pickup(t, c, cust)

    # ---- Simulator.java:613-674
    def analyze_pairs(self, t, pairs, temp_demand, temp_supply):
        by_cab = {}
        by_clnt = {}
        for cab, clnt in pairs:
            by_cab.setdefault(cab, clnt)
            by_clnt.setdefault(clnt, cab)
        supply2, demand2 = [], []
        for s, sup in enumerate(temp_supply):
            if s in by_cab:
                self._dispatch(t, sup, temp_demand[by_cab[s]])
            else:
                supply2.append(list(sup))
        for d, cust in enumerate(temp_demand):
            if d in by_clnt:
                c2 = self.id2idx[cust[0]]
                cab_id = temp_supply[by_clnt[d]][0]
                self.d_cab[c2] = cab_id
                self.d_pick[c2] = t
                if cust[3] > -1:
                    self._assign_pooled(cust[3], cab_id)
                    self.m["total_pickup_numb"] += 1
                # NOTE: pool info is NOT copied into demand[] on the LCM path (only :391-396 does)
            else:
                demand2.append(list(cust))
        return supply2, demand2

    # ---- Simulator.java:375-421
    def analyze_solution(self, t, r2c, cost, temp_demand, temp_supply):
        total = 0
        for s, sup in enumerate(temp_supply):
            c = int(r2c[s]) if s < len(r2c) else -1
            if 0 <= c < len(temp_demand) and cost[s][c] < BIG_COST and sup[1] == sup[2]:
                total += 1
                cust = temp_demand[c]
                d = self.id2idx[cust[0]]
                self.d_cab[d] = sup[0]
                self.d_pick[d] = t
                if cust[3] > -1:
                    self._assign_pooled(cust[3], sup[0])
                    self.d_pool_id[d], self.d_pool_plan[d], self.d_pool_cost[d] = cust[3], cust[4], cust[5]
                    self.m["total_pickup_numb"] += 1
                self._dispatch(t, sup, cust)
        return total

    # ---- one tick of Simulator.java:151-211 ; returns the simulog_solv line (or None)
    def tick(self, t):
        self.check_if_cab_at_destination(t)
        temp_demand = self.create_temp_demand(t)
        if not temp_demand:
            return None
        temp_supply = self.create_temp_supply()
        line = "t:%d. Initial Count of demand=%d, supply=%d. " % (t, len(temp_demand), len(temp_supply))
        cost = np.zeros((0, 0), np.int32)
        r2c = []
        if temp_supply and hasattr(self.be, "tick"):
            return self._tick_one_call(t, line, temp_demand, temp_supply)
        if temp_supply:
            temp_demand = self.analyze_pool(self.find_pool(temp_demand), temp_demand)
            cost = self.calculate_cost(temp_demand, temp_supply)
            self.m["max_model_size"] = max(self.m["max_model_size"], cost.shape[0])
            if cost.shape[0] > MAX_NON_LCM:
                pairs, lcm_min_val = self.be.lcm(cost)
                self.m["total_LCM_used"] += 1
                line += "LCM n_pairs=%d" % len(pairs)
                temp_supply, temp_demand = self.analyze_pairs(t, pairs, temp_demand, temp_supply)
                if lcm_min_val == BIG_COST:      # :188 no input for the solver
                    return line
                cost = self.calculate_cost(temp_demand, temp_supply)
                line += ". Sent to solver: demand=%d, supply=%d. " % (len(temp_demand), len(temp_supply))
            self.m["max_solver_size"] = max(self.m["max_solver_size"], cost.shape[0])
            if self.on_solver_instance is not None:
                self.on_solver_instance(t, temp_supply, temp_demand, cost)
            r2c = self.be.solve(cost)
        count = self.analyze_solution(t, r2c, cost, temp_demand, temp_supply)
        return line + "; OPT count=%d" % count

    # ---- the same tick (Simulator.java:163-208) with the whole path behind ONE backend call (td_tick)
    def _tick_one_call(self, t, line, temp_demand, temp_supply):
        temp_demand = self.analyze_pool(self.find_pool(temp_demand), temp_demand)
        n = max(len(temp_demand), len(temp_supply))
        self.m["max_model_size"] = max(self.m["max_model_size"], n)
        res = self.be.tick([s[2] for s in temp_supply], [d[1] for d in temp_demand])
        if n > MAX_NON_LCM:
            self.m["total_LCM_used"] += 1
            pairs = list(zip(res["lcm_rows"].tolist(), res["lcm_cols"].tolist()))
            line += "LCM n_pairs=%d" % len(pairs)
            kept_supply, kept_demand = self.analyze_pairs(t, pairs, temp_demand, temp_supply)
            # the device's shrink (k_tick_shrink) and analyzePairs' removal must agree
            assert [s[0] for s in kept_supply] == [temp_supply[k][0] for k in res["kept_cabs"].tolist()]
            assert [d[0] for d in kept_demand] == [temp_demand[k][0] for k in res["kept_dems"].tolist()]
            temp_supply, temp_demand = kept_supply, kept_demand
            if not res["solved"]:                # :188 no input for the solver
                return line
            line += ". Sent to solver: demand=%d, supply=%d. " % (len(temp_demand), len(temp_supply))
        self.m["max_solver_size"] = max(self.m["max_solver_size"], res["n_rest"])
        r2c = res["row_to_col"]
        # analyzeSolution reads cost[s][c] < big_cost (:378-383): a real cell of the thresholded |a - b| model
        count = self.analyze_solution(t, r2c, _RealCells(temp_supply, temp_demand), temp_demand, temp_supply)
        return line + "; OPT count=%d" % count

    # ---- Simulator.java:256-277 printMetrics (wall-clock lines are the caller's: pass them in)
    def metrics_text(self, total_simul_time=0, max_solver_time=0, max_lcm_time=0, max_pool_time=0):
        m = self.m
        lines = ["", "Total customers: %d" % self.d_id.size,
                 "Total dropped customers: %d" % m["total_dropped"],
                 "Total pickedup customers: %d" % m["total_pickup_numb"],
                 "Total customers with assigned cabs: %d" % int((self.d_cab > -1).sum()),
                 "Total simulation time [secs]: %d" % total_simul_time,
                 "Total pickup time: %d" % m["total_pickup_time"]]
        if m["total_pickup_numb"] > 0:
            lines.append("Avg pickup time: %d" % (m["total_pickup_time"] // m["total_pickup_numb"]))
        lines += ["Max model size: %d" % m["max_model_size"], "Max solver size: %d" % m["max_solver_size"],
                  "Max solver time: %d" % max_solver_time, "Max LCM time: %d" % max_lcm_time,
                  "LCM use count: %d" % m["total_LCM_used"], "Max POOL time: %d" % max_pool_time,
                  "Max POOL array size: %d" % m["max_POOL_MEM_size"], "Max POOL size: %d" % m["max_POOL_size"],
                  "Total second customers in POOL: %d" % m["total_second_passengers"]]
        return "\n".join(lines)

    def run(self, t_end=HOURS * 60):
        for t in range(t_end):
            line = self.tick(t)
            if line is not None:
                self.log.append(line)
        return self.log
