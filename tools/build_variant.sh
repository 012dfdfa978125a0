#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : builds deepmetv2_amd/variants/libdmet_hip_NAME.so for A/B experiments
set -e
cd "$(dirname "$0")/.."
mkdir -p deepmetv2_amd/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DNDEBUG $2"
for f in knn edgeconv edgemlp misc dense encoder norm edgeconv_bwd head; do /opt/rocm/bin/hipcc $FLAGS -c deepmetv2_amd/csrc/$f.hip -o /tmp/var_$1_$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o deepmetv2_amd/variants/libdmet_hip_$1.so /tmp/var_$1_knn.o /tmp/var_$1_edgeconv.o /tmp/var_$1_edgemlp.o /tmp/var_$1_misc.o /tmp/var_$1_dense.o /tmp/var_$1_encoder.o /tmp/var_$1_norm.o /tmp/var_$1_edgeconv_bwd.o /tmp/var_$1_head.o
echo built deepmetv2_amd/variants/libdmet_hip_$1.so
