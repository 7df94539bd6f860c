import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from libertem_amd.api import Context
from libertem_amd import hip
ctx = Context.make_with('hip', gpus=0)
r = bench.host_streamed(ctx, torch, hip)
for k, v in r.items():
    print(k, f"{v['GBps']:.1f} GB/s of {v['h2d_peak_GBps']:.1f} = {v['frac_of_h2d_peak']:.3f}  {v['ms_per_run']:.1f} ms")
