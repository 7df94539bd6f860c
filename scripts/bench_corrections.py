"""C2 with detector corrections (dark + gain + 50 excluded pixels): folded into the masks vs
corrected frames through a scratch buffer (device-resident, whole job)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
import libertem_amd.udf.masks as um
from libertem_amd.io.corrections import CorrectionSet

ctx = Context.make_with('hip', gpus=0)
g = torch.Generator(device='cuda').manual_seed(1)
frames = torch.randint(0, 4096, (256, 256, 256, 256), generator=g, device='cuda', dtype=torch.int16)
rng = np.random.default_rng(2)
masks = rng.random((16, 256, 256)).astype(np.float32)
dark = rng.random((256, 256)) * 20
gain = rng.random((256, 256)) + 0.5
bad = np.zeros((256, 256), dtype=bool)
bad[rng.integers(0, 256, 50), rng.integers(0, 256, 50)] = True
corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                    mask_dtype=np.float32)
res = {}
for label, fold, c in (('no corrections', True, None), ('folded into masks', True, corr),
                       ('corrected frames (scratch)', False, corr)):
    um.FOLD_CORRECTIONS = fold
    for _ in range(2):
        r = ctx.run_udf(dataset=ds, udf=udf, corrections=c)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        r = ctx.run_udf(dataset=ds, udf=udf, corrections=c)
        ts.append(time.perf_counter() - t0)
    res[label] = r['intensity'].data
    print(f"{label:30s} {np.median(ts) * 1e3:8.2f} ms per run  "
          f"{65536 / np.median(ts) / 1e6:6.2f} Mframes/s")
a, b = res['folded into masks'], res['corrected frames (scratch)']
print("folded vs corrected-frames: max rel diff", np.abs(a - b).max() / np.abs(b).max())
