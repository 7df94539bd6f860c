"""Fixed host overhead of Context.run_udf on the HIP executor (tiny dataset: kernel time ~0)."""
import cProfile
import pstats
import time
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF

ctx = Context.make_with('hip', gpus=0)
frames = torch.zeros((8, 8, 256, 256), dtype=torch.int16, device='cuda')
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                    mask_dtype=np.float32)
for _ in range(5):
    ctx.run_udf(dataset=ds, udf=udf)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    ctx.run_udf(dataset=ds, udf=udf)
torch.cuda.synchronize()
print(f"run_udf fixed overhead: {(time.perf_counter() - t0) / N * 1e6:.0f} us per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    ctx.run_udf(dataset=ds, udf=udf)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(60)
st.sort_stats('cumulative').print_stats(45)
