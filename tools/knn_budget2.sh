#!/bin/bash
# tools/knn_budget2.sh : cycle budget of the second filter form (knn_filter12_kernel) by ablation builds + rocprofv3 kernel durations (GPU box).
#   DMET_F2_ABL 1 = no exact re-rank; 2 = + no hit masks / entries; 3 = + no tile minima / tau; 4 = + no half-wave swap
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
for a in 1 2 3 4; do
  [ -f deepmetv2_amd/variants/libdmet_hip_f2abl$a.so ] || bash tools/build_variant.sh f2abl$a "-DDMET_F2_ABL=$a" >/dev/null
done
cd /tmp; export TMPDIR=/tmp
run() { # name, lib
  rm -rf /tmp/kb_$1
  DMET_KNN_PATH=filter_only DMET_HIP_LIB=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_$1 -- python3 $ROOT/tools/knn_ab.py 5 64 > /tmp/kb_$1.log 2>&1
  echo "== $1"; python3 - "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob(f"/tmp/kb_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    m = re.search(r"(knn_\w+(<\d+>)?)", r["Name"])
    if m:
        print("   %-32s calls %3s avg %8.1f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
run product ""
for a in 1 2 3 4; do run f2abl$a $ROOT/deepmetv2_amd/variants/libdmet_hip_f2abl$a.so; done
