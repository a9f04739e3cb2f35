import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 200 kernels, print sequence with durations and gaps
rows = rows[-64:]
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
    gap = (s - prev) / 1000.0 if prev else 0.0
    print("%8.1f us gap  %8.1f us  %s" % (gap, (e - s) / 1000.0, name))
    prev = e
