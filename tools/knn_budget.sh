#!/bin/bash
# tools/knn_budget.sh : cycle budget of knn_filter_kernel by ablation builds (run on the GPU box).
#   ABL 2 = keys only (loads + MFMA); 1 = + half-wave swap; 3 = + push VALU; 4 = + push LDS stores (no drains);
#   5 = full selection without the in-place re-rank; 0 = the product kernel.
set -e
cd "$(dirname "$0")/.."
for a in 1 2 3 4 5; do
  [ -f deepmetv2_amd/variants/libdmet_hip_abl$a.so ] || bash tools/build_variant.sh abl$a "-DDMET_FILTER_ABL=$a" >/dev/null
done
echo "== product"; python tools/knn_ab.py 7 64
for a in 2 1 3 4 5; do
  echo "== ABL $a"; DMET_KNN_PATH=filter_only DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_abl$a.so python tools/knn_ab.py 7 64
done
echo "== product, filter_only"; DMET_KNN_PATH=filter_only python tools/knn_ab.py 7 64
