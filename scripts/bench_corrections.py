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

# ---- the same for the other consumers: sparse ring stack (C4 masks) and CrystallinityUDF --------------
from libertem_amd import masks as M
from libertem_amd.udf.crystallinity import CrystallinityUDF


def rings():
    return M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256, n_bins=1024,
                         use_sparse=True, dtype=np.float32)


def timed(udf, c, fold, n=5):
    um.FOLD_CORRECTIONS = fold
    try:
        for _ in range(3):
            ctx.run_udf(dataset=ds, udf=udf, corrections=c)
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            ctx.run_udf(dataset=ds, udf=udf, corrections=c)
            ts.append(time.perf_counter() - t0)
    finally:
        um.FOLD_CORRECTIONS = True
    return float(np.median(ts)) * 1e3


sparse_udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=1024,
                           mask_dtype=np.float32)
cryst = CrystallinityUDF(rad_in=20, rad_out=60, real_center=(128, 128), real_rad=10)
for name, udf in (('1024 sparse ring masks', sparse_udf), ('CrystallinityUDF', cryst)):
    t0 = timed(udf, None, True)
    t1 = timed(udf, corr, True)
    t2 = timed(udf, corr, False)
    print(f"{name:28s} no corrections {t0:7.2f} ms | fused / folded {t1:7.2f} ms ({t1 / t0:.2f}x) | "
          f"corrected copy of the frames {t2:7.2f} ms ({t2 / t0:.2f}x)")
