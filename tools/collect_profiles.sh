#!/bin/bash
# tools/collect_profiles.sh TAG : the rocprofv3 runs behind profiles/TAG_* (run on the GPU box; writes gpurun_out/prof_TAG/)
set -e
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 10 --warmup 3 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 4 --warmup 2 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 4 --warmup 2 > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/gather_sq -- python3 $ROOT/tools/gather_micro.py > $OUT/gather_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/knn_sq1 -- python3 $ROOT/tools/knn_ab.py 2 64 > $OUT/knn_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/knn_sq2 -- python3 $ROOT/tools/knn_ab.py 2 64 > $OUT/knn_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mlp_sq -- python3 $ROOT/tools/edge_mlp_micro.py 32,48,32 > $OUT/mlp_sq.log 2>&1
$B > $OUT/bench_line.json 2> $OUT/bench_line.err || true
python3 $ROOT/bench.py > $OUT/bench_line_full.json 2> $OUT/bench_line_full.err || true
echo collected
