"""Host-streamed mode: frames in host memory, double-buffered hipMemcpyAsync + kernels (PCIe-bound).
    python scripts/bench_host_streamed.py [scan rows of 256 frames] [dtype, e.g. uint16 or '>u2']
A dtype in the other byte order is uploaded as it is and swapped on the GPU (ltmi_byteswap)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64     # scan rows of 256 frames
ctx = Context.make_with('hip', gpus=0)
rng = np.random.default_rng(1)
dt = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype(np.uint16)
data = rng.integers(0, 4096, (n, 256, 256, 256), dtype=np.uint16).astype(dt)
masks = rng.random((16, 256, 256)).astype(np.float32)
ds = ctx.load('memory', data=data, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
for _ in range(2):
    ctx.run_udf(dataset=ds, udf=udf)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf); ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
nf = n * 256
print(f"host-streamed ({dt.str}): {nf} frames ({data.nbytes / 2**30:.1f} GiB) in {t*1e3:.1f} ms = "
      f"{nf / t / 1e6:.3f} Mframes/s = {data.nbytes / t / 1e9:.1f} GB/s over PCIe")
