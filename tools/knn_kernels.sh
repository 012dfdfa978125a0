#!/bin/bash
# tools/knn_kernels.sh : per-kernel durations of one kNN build (rocprofv3 kernel trace of tools/knn_micro.py; GPU box)
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kk_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kk_prof -- python3 $ROOT/tools/knn_micro.py 9 64 > /tmp/kk_prof.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/kk_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    m = re.search(r"(knn_\w+(<[\d, ]+>)?)", r["Name"])
    if m:
        print("   %-36s calls %3s avg %8.1f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "^knn" /tmp/kk_prof.log
