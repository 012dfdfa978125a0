import sys, json
tag = sys.argv[1]
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, d["value"], d["ms_per_step"], "host", d["host_ms_per_step"], "knn", d["kernels"].get("knn", {}).get("avg_us"), "gather", d["kernels"].get("gather_max", {}).get("avg_us"), "frac", d["roofline"]["frac"])
