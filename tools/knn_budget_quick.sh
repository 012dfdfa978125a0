set -e
echo "== product"; python tools/knn_ab.py 7 64
for a in 2 4 5; do
  echo "== ABL $a"; DMET_KNN_PATH=filter_only DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_abl$a.so python tools/knn_ab.py 7 64
done
