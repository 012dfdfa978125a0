#!/bin/bash
# tools/step_pmc.sh KERNEL-SUBSTRING... : SQ counters of the named kernels inside the training step (two rocprofv3 --pmc
# passes over `bench.py --steps 3 --warmup 2 --no-cpu-baseline`; GPU box)
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sp1 /tmp/sp2
B="python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/sp1 -- $B > /tmp/sp1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d /tmp/sp2 -- $B > /tmp/sp2.log 2>&1
for d in sp1 sp2; do
  f=$(find /tmp/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $ROOT/tools/pmc_table.py $f "$@" || tail -3 /tmp/$d.log
done
