python tools/ab_libs.py --libs oflibpytorch_amd/libofl_hip.so $OVF_LIBS --ops apply_t combine3 --batch 64 --sigma 8 --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
python tools/ab_libs.py --libs oflibpytorch_amd/libofl_hip.so $OVF_LIBS --ops apply_s switch_ref --batch 16 --sigma 8 --rounds 5 --iters 10 --check 2>&1 | grep "sigma\|differ"
