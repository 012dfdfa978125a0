#!/bin/bash
# tools/knn_variants.sh NAME... : kNN build time on the model's embeddings (tools/knn_model_stats.py) for the product
# library and for deepmetv2_amd/variants/libdmet_hip_NAME.so (A/B experiments; GPU box)
cd "$(dirname "$0")/.."
echo "== product"; python3 tools/knn_model_stats.py 0 2>&1 | grep "^steps"
for v in "$@"; do
  echo "== $v"; DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_$v.so python3 tools/knn_model_stats.py 0 2>&1 | grep "^steps"
done
