#!/bin/bash
# Run ON the GPU box: dynamic instruction counts of the gather splat's first launch, phase by phase.  Needs the variants
#   for k in 1 2 6; do tools/build_variant.sh skip$k -DOFL_SP2_SKIP=$k; done
# (WRONG results, same launches: 1 no phase S, 2 no phase C, 4 no finalize -- which takes phase C with it --, 8 no records).
# usage: tools/prof_phase_insts.sh <outdir> [sigma ...]
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; shift
for sg in ${@:-2 8}; do for v in ${VARIANTS:-default skip1 skip2 skip6}; do
  lib=$R/tools/microbench/var/$v.so; [ $v = default ] && lib=$R/oflibpytorch_amd/libofl_hip.so
  (cd /tmp && OFL_HIP_LIB=$lib OFL_SIGMA=$sg timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $O/s${sg}_$v -- python3 $R/tools/splat_once.py > /dev/null 2>&1)
done; done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, re
for d in sorted(glob.glob(sys.argv[1] + '/s*_*')):
    f = glob.glob(d + '/*/*counter_collection.csv')
    if not f: print(d, 'no counters'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        m = re.search(r'splat_gather2_kernel<(\d), \w+, float, float, \w+, (true|false)>', r['Kernel_Name'])
        if not m or m.group(2) == 'true': continue
        acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for nc, c in sorted(acc.items()):
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        w = mean.get('SQ_WAVES', 1)
        print("%-28s %s ch: per wave VALU %6.1f SALU %6.1f LDS %5.1f | busy CU cycles %.3g" % (d.split('/')[-1], nc, mean.get('SQ_INSTS_VALU', 0) / w, mean.get('SQ_INSTS_SALU', 0) / w, mean.get('SQ_INSTS_LDS', 0) / w, mean.get('SQ_BUSY_CU_CYCLES', 0)))
PY
