"""Full 120-tick replay of the reference's simulation input with every path operation on the GPU
(dev tool; prints timing + the printMetrics block)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taxidispatcher_amd as td
from taxidispatcher_amd import simulator
td.init(0)
rows = simulator.read_demand("tests/golden/taxi_demand.txt.gz")
sim = simulator.Simulator(rows)
t0 = time.time()
per = {"pool": 0.0, "cost": 0.0, "lcm": 0.0, "solve": 0.0}
be = sim.be
def timed(name, fn):
    def w(*a, **k):
        t = time.time(); r = fn(*a, **k); per[name] += time.time() - t; return r
    return w
be.find_pool = timed("pool", be.find_pool); be.calculate_cost = timed("cost", be.calculate_cost)
be.lcm = timed("lcm", be.lcm); be.solve = timed("solve", be.solve)
log = sim.run(120)
dt = time.time() - t0
print("120 ticks in %.2f s (reference: 2603 s, README.md:45); path time on GPU incl. PCIe: %s" % (dt, {k: round(v, 3) for k, v in per.items()}))
print("\n".join(log[-3:]))
print(sim.metrics_text(total_simul_time=int(dt)))
