python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "oversize or lds_and_generic" 2>&1 | tail -3
AB_BATCHES=1,2 python tools/ab_small_batch.py tools/microbench/var/clipn.so tools/microbench/var/clipc.so tools/microbench/var/clipn.so tools/microbench/var/clipc.so 2>&1 | grep "B="
python tools/bench_grad.py 2>&1 | tail -6
