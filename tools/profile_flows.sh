#!/bin/bash
# tools/profile_flows.sh TAG : rocprofv3 kernel stats of the non-default bench flows (ragged configs[4], static radius flow)
set -e
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/flows_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ragged -- $B --ragged 500 8000 > $OUT/ragged.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/static_table -- $B --graph static-table > $OUT/static_table.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/static -- $B --graph static > $OUT/static.log 2>&1
echo collected
