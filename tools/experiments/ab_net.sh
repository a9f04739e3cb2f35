python tools/fuzz_gpu.py --seconds 240 --seed 31 2>&1 | grep -v "^ \|^$" | tail -4
python -m pytest tests -q -m gpu 2>&1 | tail -2
for s in 8 12 2; do
  python tools/ab_libs.py --libs tools/microbench/var/nonet.so oflibpytorch_amd/libofl_hip.so --ops apply_s switch_ref --batch 16 --sigma $s --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
done
