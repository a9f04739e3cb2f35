#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel times of the splat's launches (bin, first, second gather launch) for apply 's' and switch_ref,
# B = 16, at the given roughnesses.   usage: tools/prof_splat_kernels.sh <outdir> [sigma ...]
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; shift
for sg in ${@:-2 8 12}; do
  (cd /tmp && OFL_SIGMA=$sg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$sg -- python3 $R/tools/splat_once.py > /dev/null 2>&1)
done
python3 - "$O" ${@:-2 8 12} <<'PY'
import csv, glob, re, sys
for sg in sys.argv[2:]:
    f = glob.glob('%s/s%s/*/*kernel_stats.csv' % (sys.argv[1], sg))
    if not f: continue
    out = []
    for r in csv.DictReader(open(f[0])):
        if 'splat_' in r['Name']:
            m = re.search(r'splat_gather2_kernel<(\d), \w+, float, float, \w+, (true|false)>', r['Name'])
            name = ("%s ch %s launch" % (m.group(1), 'second' if m.group(2) == 'true' else 'first ')) if m else re.sub(r'.*(splat_\w+).*', r'\1', r['Name'])
            out.append("%s %.1f us" % (name, float(r['AverageNs']) / 1e3))
    print("sigma %s: %s" % (sg, " | ".join(sorted(out))))
PY
