"""ROI runs of ApplyMasksUDF on a device-resident C2 dataset: whole job per run (Context.run_udf(roi=...)).
The mask kernels read the selected frames through a row list (ltmi_apply_masks_rows), no gathered copy."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd.common.hiparray import HipArray
ctx = Context.make_with('hip', gpus=0)
frames = torch.randint(0, 4096, (256, 256, 256, 256), device='cuda', dtype=torch.int32).to(torch.int16)
if '--f64' in sys.argv:            # int32 detector: float64 results (k_dense_lds64)
    frames = frames.to(torch.int32)
ds = ctx.load('memory', data=HipArray.from_torch(frames, np.int32 if '--f64' in sys.argv else np.uint16),
              sig_dims=2, num_partitions=1)
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
if '--sparse' in sys.argv:          # C4: 1024 sparse ring masks (256 MiB result: kept in HBM here)
    from libertem_amd import masks as pm
    rings = pm.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32)
    udf = ApplyMasksUDF(mask_factories=lambda: rings, use_sparse='scipy.sparse', mask_count=1024,
                        mask_dtype=np.float32) if '--hints' in sys.argv else ApplyMasksUDF(mask_factories=lambda: rings)
    run_kw = dict(result_where='device')
else:
    run_kw = {}
if '--gather' in sys.argv:          # the earlier behaviour: the dataset gathers the selected frames
    ApplyMasksUDF.ACCEPTS_ROW_VIEWS = False
rng = np.random.default_rng(0)
for name, roi in (('none', None), ('random 50 %', rng.random((256, 256)) < 0.5), ('block 50 %', np.arange(65536).reshape(256, 256) < 32768), ('random 10 %', rng.random((256, 256)) < 0.1)):
    for _ in range(3):
        ctx.run_udf(dataset=ds, udf=udf, roi=roi, **run_kw)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf, roi=roi, **run_kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    n = 65536 if roi is None else int(roi.sum())
    t = float(np.median(ts))
    print(f"roi {name:12s}: {n:6d} frames {t*1e3:7.2f} ms  {n / t / 1e6:6.1f} M frames/s", flush=True)
