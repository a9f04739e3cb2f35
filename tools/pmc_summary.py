#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc / --kernel-trace CSV output: mean counter value and mean duration per kernel.

    python tools/pmc_summary.py <dir-with-rocprofv3-csvs> [substring-filter]
"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if filt in k:
                    counters[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if filt in k:
                    durs[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k in sorted(set(counters) | set(durs)):
        d = durs.get(k, [])
        print("%s  launches=%d  avg_us=%.1f" % (k[:110], len(d), sum(d) / len(d) if d else float('nan')))
        for c, v in sorted(counters.get(k, {}).items()):
            print("    %-40s mean=%.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
