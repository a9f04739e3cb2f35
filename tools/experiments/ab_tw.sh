for s in 8 2 12; do
  python tools/ab_libs.py --libs tools/microbench/var/tw32.so tools/microbench/var/tw64.so --ops apply_s switch_ref --batch 16 --sigma $s --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
done
python tools/ab_libs.py --libs tools/microbench/var/tw32.so tools/microbench/var/tw64.so --ops apply_s --batch 1 --sigma 8 --rounds 5 --iters 20 2>&1 | grep "sigma\|differ"
