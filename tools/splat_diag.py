import sys, torch
sys.path.insert(0, '/root/repo')
import bench, oflibpytorch_amd as ofl
from oflibpytorch_amd import _native
_native.collect_splat_stats = True
dev = torch.device('cuda', 0)
n, h, w = 4, 1080, 1920
tiles = n * 60 * 68
for sigma in (2.0, 8.0, 12.0):
    f = bench.smooth_flow(n, h, w, sigma, 1000, dev)
    ofl.Flow(f, 's').switch_ref(); torch.cuda.synchronize()
    st = _native._last_splat_stats.cpu().tolist()
    print("sigma %.0f: tiles %d  band passes %d (%.2f per tile), banded tiles %d (%.1f%%), cells with 2 records %.1f per pass, >2 records %.1f, >4 records %.2f, fold tiles %d" % (
        sigma, tiles, st[4], st[4] / tiles, st[5], 100.0 * st[5] / tiles, st[3] / max(st[4], 1), st[6] / max(st[4], 1), st[7] / max(st[4], 1), st[1]))
