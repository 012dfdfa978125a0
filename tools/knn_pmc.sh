#!/bin/bash
# tools/knn_pmc.sh : SQ counters of the kNN filter kernel (two rocprofv3 --pmc passes over tools/knn_ab.py; GPU box)
cd "$(dirname "$0")/.."
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kp1 /tmp/kp2 /tmp/kp3
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/kp1 -- python3 $ROOT/tools/knn_ab.py 2 64 > /tmp/kp1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d /tmp/kp2 -- python3 $ROOT/tools/knn_ab.py 2 64 > /tmp/kp2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_IFETCH --output-format csv -d /tmp/kp3 -- python3 $ROOT/tools/knn_ab.py 2 64 > /tmp/kp3.log 2>&1
for d in kp1 kp2 kp3; do
  f=$(find /tmp/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $ROOT/tools/pmc_table.py $f knn_filter12 || tail -3 /tmp/$d.log
done
