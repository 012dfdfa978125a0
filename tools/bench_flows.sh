#!/bin/bash
# tools/bench_flows.sh : the bench lines of every flow profiles/ quotes (run on the GPU box; writes gpurun_out/bench_line_*.json,
# which tools/update_profiles.sh copies into profiles/TAG_bench_line*.json)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $OUT/bench_line_$name.json 2> $OUT/bench_line_$name.err || echo "FAILED $name"; }
run default
run ragged5008000 --no-cpu-baseline --ragged 500 8000
run graphstatictablehipgraph --no-cpu-baseline --graph static-table --hip-graph
run graphstatictable --no-cpu-baseline --graph static-table
run graphstatic --no-cpu-baseline --graph static
run modeinfer --no-cpu-baseline --mode infer
run dtypebf16 --no-cpu-baseline --dtype bf16
run inputhost --no-cpu-baseline --input host
run hipgraph --no-cpu-baseline --hip-graph
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_line_torchrun.json 2> $OUT/bench_line_torchrun.err || echo "FAILED torchrun"
echo flows done
