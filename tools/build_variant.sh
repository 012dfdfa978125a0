#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : builds deepmetv2_amd/variants/libdmet_hip_NAME.so for A/B experiments
# (objects go to a fresh directory: a failed compile can never be papered over by an older object of the same name)
set -e
cd "$(dirname "$0")/.."
mkdir -p deepmetv2_amd/variants
OBJ=$(mktemp -d /tmp/var_$1_XXXXXX)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DNDEBUG $2"
pids=()
for f in knn edgeconv edgemlp misc dense encoder norm edgeconv_bwd head finalize; do
  /opt/rocm/bin/hipcc $FLAGS -c deepmetv2_amd/csrc/$f.hip -o $OBJ/$f.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o deepmetv2_amd/variants/libdmet_hip_$1.so $OBJ/knn.o $OBJ/edgeconv.o $OBJ/edgemlp.o $OBJ/misc.o $OBJ/dense.o $OBJ/encoder.o $OBJ/norm.o $OBJ/edgeconv_bwd.o $OBJ/head.o $OBJ/finalize.o
rm -rf $OBJ
echo built deepmetv2_amd/variants/libdmet_hip_$1.so
