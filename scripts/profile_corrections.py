import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd.io.corrections import CorrectionSet
ctx = Context.make_with('hip', gpus=0)
frames = torch.zeros((16, 16, 256, 256), device='cuda', dtype=torch.int16)
rng = np.random.default_rng(2)
masks = rng.random((16, 256, 256)).astype(np.float32)
corr = CorrectionSet(dark=rng.random((256, 256)), gain=rng.random((256, 256)) + 0.5)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
for _ in range(3): ctx.run_udf(dataset=ds, udf=udf, corrections=corr)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): ctx.run_udf(dataset=ds, udf=udf, corrections=corr)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
