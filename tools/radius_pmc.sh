#!/bin/bash
# tools/radius_pmc.sh : SQ counters of the static flow's graph kernels (radius_window_kernel, counted gather) on tools/static_micro.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/radius_pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $ROOT/tools/static_micro.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $OUT/b -- python3 $ROOT/tools/static_micro.py > $OUT/b.log 2>&1
python3 $ROOT/tools/pmc_table.py $(ls $OUT/a/*/*counter_collection.csv | head -1) radius_window gather_max_lds_counted > $OUT/summary.txt
python3 $ROOT/tools/pmc_table.py $(ls $OUT/b/*/*counter_collection.csv | head -1) radius_window gather_max_lds_counted >> $OUT/summary.txt
cat $OUT/summary.txt
