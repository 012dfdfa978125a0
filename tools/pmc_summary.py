"""Derive profiles/<tag>_pmc_gather_max.json and the filtered counter csvs from two rocprofv3 --pmc passes.

  python tools/pmc_summary.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <round tag>

The passes: `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ... -- python bench.py --steps 4
--warmup 2 --no-cpu-baseline` and the same with WRITE_SIZE (separate runs).  Counters are KiB; FETCH_SIZE is doubled
(gfx950 reports half of a wide coalesced read; calibrated on node_linear_split_kernel whose byte counts are known).
"""
import collections
import csv
import json
import os
import sys

fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, out_name):
    rows = [r for r in csv.DictReader(open(path)) if "dmet::" in r["Kernel_Name"]]
    with open(os.path.join(root, "profiles", out_name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


f = load(fetch_csv, f"{tag}_pmc_fetch_size_counter_collection.csv")
w = load(write_csv, f"{tag}_pmc_write_size_counter_collection.csv")


def mean_of(d, key):
    names = [n for n in d if key in n]
    vals = [v for n in names for v in d[n]]
    return names, sum(vals) / len(vals), len(vals)


names, gf, n = mean_of(f, "gather_max_lds_kernel<true")
_, gw, _ = mean_of(w, "gather_max_lds_kernel<true")
_, cf, _ = mean_of(f, "node_linear_split_kernel")
_, cw, _ = mean_of(w, "node_linear_split_kernel")
out = {
    "kernel": names[0].replace("void dmet::(anonymous namespace)::", "").split("(float")[0],
    "config": "bench.py train step, 64 events x 4500 nodes, k=16 (N=288000)",
    "FETCH_SIZE_raw_KiB_per_launch": gf,
    "WRITE_SIZE_raw_KiB_per_launch": gw,
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python bench.py "
              f"--steps 4 --warmup 2 --no-cpu-baseline`, mean over the {n} dispatches; bytes = counter*1024; FETCH_SIZE "
              "doubled (gfx950 reports 1/2 of a wide coalesced read, MI355X_MICROARCH.md section HBM). Calibration in the "
              "same runs: node_linear_split_kernel reads 36.86 MB / writes 73.73 MB by construction and shows "
              f"FETCH_SIZE*1024*2 = {cf * 2048 / 1e6:.2f} MB, WRITE_SIZE*1024 = {cw * 1024 / 1e6:.2f} MB.",
    "fetch_bytes_per_launch": gf * 2048,
    "write_bytes_per_launch": gw * 1024,
    "hbm_bytes_per_launch": gf * 2048 + gw * 1024,
    "note": "algorithmic bytes (SURVEY 8d) = 101.4 MB (P rows, int32 ids, out, uint8 arg); the kernel reads the Q table "
            "once (36.9 MB) and takes its ids from the uint16 event-local table (9.2 MB instead of 18.4 MB)",
}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_gather_max.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
