#!/bin/bash
# tools/stock_profile.sh [model]: rocprofv3 kernel statistics of the drop-in training step (bench.py --model stock-*; GPU box)
cd "$(dirname "$0")/.."
ROOT=$PWD
MODEL=${1:-stock-knn-graph}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/stock_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stock_prof -- python3 $ROOT/bench.py --model $MODEL --steps 20 --warmup 5 --prewarm-ms 0 --no-cpu-baseline > /tmp/stock_prof.log 2>&1
f=$(find /tmp/stock_prof -name "*kernel_stats.csv" | head -1)
mkdir -p $ROOT/gpurun_out/r3
cp "$f" $ROOT/gpurun_out/r3/${MODEL}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print("%6.2f%%  calls %5s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:150]))
PY
tail -2 /tmp/stock_prof.log | cut -c1-300
