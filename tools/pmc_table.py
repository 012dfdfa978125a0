"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean counter value per (kernel, counter) + mean duration.
  python tools/pmc_table.py <counter_collection.csv> [kernel-name substring ...]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if keys and not any(k in n for k in keys):
        continue
    m = re.search(r"(\w+_kernel(<[^(]*>)?)", n)
    short = m.group(1) if m else n[:50]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[(short, r["Dispatch_Id"])] = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])]
for k in acc:
    ds = [v[0] for (s, _), v in dur.items() if s == k]
    print(f"{k}: dispatches {len(ds)}, mean duration {sum(ds) / len(ds) / 1e3:.1f} us")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {sum(v) / len(v):16.1f}")
