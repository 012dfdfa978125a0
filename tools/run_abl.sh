# tools/run_abl.sh SCRIPT [ARGS] -- VARIANT...: runs a micro-benchmark under the default library and each variant library
script=(); while [ "$1" != "--" ]; do script+=("$1"); shift; done; shift
for v in base "$@"; do
  if [ "$v" = base ]; then unset DMET_HIP_LIB; else export DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_$v.so; fi
  echo "== $v"; python "${script[@]}" 2>&1 | grep -v amdgpu.ids
done
