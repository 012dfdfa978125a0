#!/bin/bash
# tools/knn_lean.sh : sweep-only timing (DMET_F2_ABL=1 builds) of the second filter form at two and at three wavefronts
# per SIMD (DMET_F2_LEAN: 26 entry slots, no row staging, registers capped at 168) -- rocprofv3 kernel durations (GPU box).
# Build first: tools/build_variant.sh abl1 "-DDMET_F2_ABL=1"; tools/build_variant.sh lean_abl1 "-DDMET_F2_ABL=1 -DDMET_F2_LEAN"
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
for v in abl1 lean_abl1 "$@"; do
  rm -rf /tmp/lean_$v
  DMET_KNN_PATH=filter_only DMET_HIP_LIB=$ROOT/deepmetv2_amd/variants/libdmet_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lean_$v -- python3 $ROOT/tools/knn_micro.py 9 64 > /tmp/lean_$v.log 2>&1
  f=$(find /tmp/lean_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep '^knn' /tmp/lean_$v.log)"
  python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(knn_\w+(<[\d, ]+>)?)", r["Name"])
    if m: print("   %-36s calls %3s avg %8.1f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
