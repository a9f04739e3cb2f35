#!/bin/bash
# Run ON the GPU box: kernel times of the gather splat's two launches with parts of the second launch switched off (OFL_SPLAT_DEBUG:
# 1 no wave-ordered big cells, 2 no phase S, 4 no phase C, 8 nothing) -- where a banded tile's time goes.  usage: tools/prof_redo.sh <outdir>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
for sg in 8 12; do for d in 0 1 2 6 8; do
  (cd /tmp && OFL_SIGMA=$sg OFL_SPLAT_DEBUG=$d timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s${sg}_d$d -- python3 $R/tools/splat_once.py > /dev/null 2>&1)
  f=$(ls $O/s${sg}_d$d/*/*kernel_stats.csv | head -1)
  echo "sigma $sg debug $d: $(grep splat_gather2 $f | sed 's/(anonymous namespace):://g' | awk -F'","' '{printf "%s calls %s avg %.1f us | ", substr($1,28,40), $2, $4/1000}')"
done; done
