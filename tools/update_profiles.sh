#!/bin/bash
# tools/update_profiles.sh TAG : copy what tools/collect_profiles.sh / profile_flows.sh / the per-flow bench runs left under
# gpurun_out/ into profiles/TAG_* (run in the repo after a gpurun call merged its output back)
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
P=gpurun_out/prof_$TAG
cp "$(ls -t $P/stats/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_bench_train_kernel_stats.csv
python tools/pmc_summary.py "$(ls -t $P/fetch/runc/*counter_collection.csv | head -1)" "$(ls -t $P/write/runc/*counter_collection.csv | head -1)" $TAG > /dev/null
cpif() { [ -s "$1" ] && cp "$1" "$2" || echo "missing $1"; }
cpif gpurun_out/bench_line_default.json profiles/${TAG}_bench_line.json
cpif gpurun_out/bench_line_torchrun.json profiles/${TAG}_bench_line_torchrun_nccl_world1.json
cpif gpurun_out/bench_line_ragged5008000.json profiles/${TAG}_bench_line_ragged.json
cpif gpurun_out/bench_line_graphstatictablehipgraph.json profiles/${TAG}_bench_line_static_table_hipgraph.json
cpif gpurun_out/bench_line_graphstatictable.json profiles/${TAG}_bench_line_static_table.json
cpif gpurun_out/bench_line_graphstatic.json profiles/${TAG}_bench_line_static.json
cpif gpurun_out/bench_line_modeinfer.json profiles/${TAG}_bench_line_infer.json
cpif gpurun_out/bench_line_dtypebf16.json profiles/${TAG}_bench_line_bf16.json
cpif gpurun_out/bench_line_inputhost.json profiles/${TAG}_bench_line_input_host.json
cpif gpurun_out/bench_line_hipgraph.json profiles/${TAG}_bench_line_hipgraph.json
for d in ragged static_table static; do
  cp "$(ls -t gpurun_out/flows_$TAG/$d/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_bench_${d}_kernel_stats.csv
done
python - <<PY
import json, glob
for f in sorted(glob.glob("profiles/${TAG}_bench_line*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f.split("/")[-1].ljust(48), d["value"], d["ms_per_step"], r.get("frac"), r.get("cold_frac"),
          d["kernels"].get("knn", {}).get("avg_us"), (d.get("cpu_baseline") or {}).get("value"))
PY
