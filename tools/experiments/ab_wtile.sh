for s in 8 2 12; do
  python tools/ab_libs.py --libs $OVF_LIBS --ops apply_t combine3 --batch 64 --sigma $s --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
done
