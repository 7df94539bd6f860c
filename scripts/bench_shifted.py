"""ApplyMasksUDF(shifts=...) on C2-sized data (row f1), device resident, whole job."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF

scan = int(os.environ.get('SCAN', 128))
ctx = Context.make_with('hip', gpus=0)
fr = torch.randint(0, 4096, (scan, scan, 256, 256), device='cuda', dtype=torch.int16)
ds = ctx.load('memory', data=fr, dtype=np.uint16, sig_dims=2, num_partitions=1)
rng = np.random.default_rng(0)
masks = rng.random((16, 256, 256)).astype(np.float32)
shifts = rng.integers(-6, 7, (scan, scan, 2))
aux = ApplyMasksUDF.aux_data(shifts.reshape((-1, 2)).ravel(), kind='nav', extra_shape=(2,), dtype=shifts.dtype)
for label, sh in (('constant shift', (3, -2)), ('per-frame shifts', aux), ('no shifts', None)):
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                        mask_dtype=np.float32, shifts=sh)
    for _ in range(2):
        ctx.run_udf(dataset=ds, udf=udf)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf); ts.append(time.perf_counter() - t0)
    t = np.median(ts)
    print(f"{label:18s}: {t * 1e3:8.2f} ms  {scan * scan / t / 1e6:7.2f} Mframes/s  "
          f"{scan * scan * 131072 / t / 1e9:6.0f} GB/s")
