for v in base bwdabl1 bwdabl2 bwdabl3 bwdabl4; do
  if [ "$v" = base ]; then unset DMET_HIP_LIB; else export DMET_HIP_LIB=$PWD/deepmetv2_amd/variants/libdmet_hip_$v.so; fi
  echo "== $v"; python tools/bwd_scatter_micro.py 2>&1 | grep "uint16\|k=1"
done
