for s in 8 2; do
  python tools/ab_libs.py --libs oflibpytorch_amd/libofl_hip.so $OVF_LIBS --ops combine3 --batch 64 --sigma $s --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
done
