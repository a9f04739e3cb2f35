for rep in 1 2; do
for lib in oflibpytorch_amd/libofl_hip.so tools/microbench/var/rows_nodedupe.so tools/microbench/var/rows_nopad.so; do
  echo "== $lib"; OFL_HIP_LIB=$PWD/$lib python tools/rows_check.py --batch 16 --sigmas 0.5 8 16 --no-hostile 2>/dev/null | grep sigma
done; done
