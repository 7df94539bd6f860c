import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
ctx = Context.make_with('hip', gpus=0)
scan, sig = 512, 512
fr = torch.randint(0, 4096, (scan * scan // 4, sig, sig), device='cuda', dtype=torch.int16)
fr = fr.repeat(4, 1, 1)
ds = ctx.load('memory', data=fr.reshape(scan, scan, sig, sig), dtype=np.uint16, sig_dims=2, num_partitions=1)
an = ctx.create_com_analysis(dataset=ds, cx=256, cy=256)
ctx.run(an)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    ctx.run(an)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
