"""Condenses rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a small
text/JSON summary that is committed under profiles/."""
import csv, glob, json, os, sys, collections

out = sys.argv[1]

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    i = name.find("(")
    return name[:i] if i > 0 else name

import hashlib
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taxidispatcher_amd", "libtaxidispatcher_amd.so")
res = {"kernel_stats": [], "pmc": {},
       # the library these counters were taken with: bench.py prints the same hash of the library IT loaded next to the traffic figure
       "library_sha16": hashlib.sha256(open(_lib, "rb").read()).hexdigest()[:16] if os.path.exists(_lib) else None}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        res["kernel_stats"].append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "total_us": float(r["TotalDurationNs"]) / 1e3,
                                    "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"]),
                                    "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3})
for key, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, key, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                k = short(r["Kernel_Name"])
                acc[k][0] += float(r["Counter_Value"])
                acc[k][1] += 1
    res["pmc"][ctr] = {k: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(1, v[1])} for k, v in acc.items()}
# HBM bytes per dispatch, gfx950 corrections of MI355X_MICROARCH.md (HBM section): counters are in
# KiB... (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes); FETCH_SIZE reads exactly 1/2 of the
# bytes of a wide coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE exact for 16 B/lane stores.
traffic = {}
for k in set(res["pmc"].get("FETCH_SIZE", {})) | set(res["pmc"].get("WRITE_SIZE", {})):
    f = res["pmc"].get("FETCH_SIZE", {}).get(k, {}).get("per_dispatch", 0.0)
    w = res["pmc"].get("WRITE_SIZE", {}).get(k, {}).get("per_dispatch", 0.0)
    traffic[k] = {"fetch_KB_raw": f, "write_KB_raw": w, "hbm_bytes_corrected": (2.0 * f + w) * 1024.0}
res["traffic_per_dispatch"] = traffic
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("%-42s %6s %10s %10s %6s" % ("kernel", "calls", "avg_us", "total_us", "pct"))
for r in sorted(res["kernel_stats"], key=lambda r: -r["total_us"]):
    print("%-42s %6d %10.2f %10.1f %6.2f" % (r["kernel"][:42], r["calls"], r["avg_us"], r["total_us"], r["pct"]))
print()
print("%-42s %14s %14s %16s" % ("kernel", "FETCH_KB/disp", "WRITE_KB/disp", "HBM bytes (corr)"))
for k, t in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"]):
    print("%-42s %14.1f %14.1f %16.0f" % (k[:42], t["fetch_KB_raw"], t["write_KB_raw"], t["hbm_bytes_corrected"]))
try:
    b = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    print()
    print("bench: value=%.0f %s ms_per_step=%.3f roofline=%s" % (b["value"], b["unit"], b["ms_per_step"], json.dumps(b.get("roofline"))))
except Exception as e:
    print("no bench line:", e)
