"""Readable view of a rocprofv3 kernel_stats.csv: python tools/kstats.py <dir or csv> [rows]."""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(p)))[:rows]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(name[:84].ljust(84), r["Calls"].rjust(6), "%10.3f ms  avg %10.1f us  %5s%%" % (float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
