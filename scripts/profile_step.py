"""cProfile of full-size run_udf steps (C2) to see host time around the kernel."""
import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
ctx = Context.make_with('hip', gpus=0)
frames = torch.zeros((256, 256, 256, 256), dtype=torch.int16, device='cuda')
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
for _ in range(5):
    ctx.run_udf(dataset=ds, udf=udf)
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf); ts.append(time.perf_counter() - t0)
print("step ms: median %.3f min %.3f" % (np.median(ts) * 1e3, np.min(ts) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    ctx.run_udf(dataset=ds, udf=udf)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
