#!/bin/bash
# tools/ab_variants.sh NAME...: same-box A/B of deepmetv2_amd/variants/libdmet_hip_NAME.so against the default library:
# kNN build on a gaussian batch (bits compared with the default library's) and the bench step (hipGraph replay).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
rm -f gpurun_out/ab/ref.pt
for v in base "$@" base; do
  if [ "$v" = base ]; then unset DMET_HIP_LIB; else export DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/ab/log.txt
  timeout -k 10 120 python tools/knn_ab.py 25 64 gpurun_out/ab/ref.pt 2>&1 | grep -v "^stats" | tee -a gpurun_out/ab/log.txt
  timeout -k 10 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --hip-graph 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], 'events/s', d['ms_per_step'], 'ms/step  knn', d['kernels']['knn']['avg_us'], 'us  gather', d['kernels']['gather_max']['avg_us'])" | tee -a gpurun_out/ab/log.txt
done
