for s in 8 12 16; do
  echo "== sigma $s B=64"
  python tools/ab_libs.py --libs oflibpytorch_amd/libofl_hip.so tools/microbench/var/skipov.so --ops apply_t combine3 --batch 64 --sigma $s --rounds 5 --iters 10 2>&1 | grep "sigma" | tail -4
done
