"""tools/stats_table.py KERNEL_STATS.csv STEPS : per-step kernel time table from a rocprofv3 --stats run of bench.py."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    name = re.sub(r"dmet::\(anonymous namespace\)::|void |at::native::|\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:56]
    us = float(r["TotalDurationNs"]) / 1e3 / steps
    tot += us
    if us >= 2.0:
        print(f"{name:56s} calls/step {float(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  per step {us:8.1f} us")
print(f"total per step {tot:8.1f} us")
