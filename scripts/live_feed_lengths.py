import sys, os, json
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from libertem_amd.api import Context
ctx = Context.make_with('hip', gpus=0)
for n in (16384, 32768, 65536):
    r = bench.live_feed(ctx, n_frames=n)
    print(n, 'iterator %.1f GB/s %.2f ms | in-place %.1f GB/s %.2f ms' % (r['GBps'], r['ms_per_scan'], r['in_place']['GBps'], r['in_place']['ms_per_scan']), flush=True)
