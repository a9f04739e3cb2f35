#!/bin/bash
# Run ON the GPU box with a -DOFL_SP2_DEBUG=1 build (OFL_HIP_LIB): kernel time of the gather splat's one-scan path with phases switched
# off (OFL_SPLAT_DEBUG bits: 1 no phase S, 2 no phase C, 4 no records, 8 no finalize) -- where a tile's time goes.
#   usage: tools/prof_phases.sh <outdir> <lib.so>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
for sg in 2 8; do for d in 0 1 2 3 7 15 8; do
  (cd /tmp && OFL_HIP_LIB=$R/$2 OFL_SIGMA=$sg OFL_SPLAT_DEBUG=$d timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s${sg}_d$d -- python3 $R/tools/splat_once.py > /dev/null 2>&1)
done; done
python3 - <<PY
import csv, glob, re
for sg in (2, 8):
    for d in (0, 1, 2, 3, 7, 15, 8):
        f = glob.glob('$O/s%d_d%d/*/*kernel_stats.csv' % (sg, d))
        if not f: continue
        out = []
        for r in csv.DictReader(open(f[0])):
            if 'splat_gather2' in r['Name'] or 'splat_bin' in r['Name']:
                m = re.search(r'<(\d), true, float, float, true, (true|false)>', r['Name'])
                out.append(("NC=%s %s" % (m.group(1), 'second' if m.group(2) == 'true' else 'first ') if m else 'bin        ') + " %.1f us" % (float(r['AverageNs']) / 1e3))
        print("sigma %2d skip %2d: " % (sg, d) + " | ".join(sorted(out)))
PY
