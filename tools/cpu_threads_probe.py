import sys, time
sys.path.insert(0, '.')
import bench
for t in (8, 16, 32, 64, 128, 256):
    r = bench.cpu_baseline(1080, 1920, 3.0, t)
    print(t, r["value"])
