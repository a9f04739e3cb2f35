for s in 8 12 16; do
  echo "== sigma $s B=64"
  python tools/ab_libs.py --libs tools/microbench/var/rt0.so tools/microbench/var/rt1.so --ops apply_t combine3 --batch 64 --sigma $s --rounds 5 --iters 10 --check 2>&1 | grep -v "^$" | tail -8
done
