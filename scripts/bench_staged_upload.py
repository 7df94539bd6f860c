"""Why the staged upload varies between processes: the rate of ltmi_host_copy out of the source into a page-locked buffer,
beside the staged run_udf, in one process (run several: NUMA placement differs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd import hip
from libertem_amd.udf.masks import ApplyMasksUDF

ctx = Context.make_with('hip', gpus=0)
nbytes = 2 << 30
u16 = np.random.default_rng(0).integers(0, 4096, nbytes // 2, dtype=np.uint16)
keep = torch.from_numpy(u16.view(np.int16)).clone()
src = keep.numpy().view(np.uint16)
pinned = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True).numpy()
piece = pinned.nbytes
srcb = src.view(np.uint8)
for threads in (0, 8, 16, 32):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for off in range(0, nbytes, piece):
            hip.host_copy(pinned, srcb[off:off + piece], threads=threads)
        ts.append(time.perf_counter() - t0)
    print(f"host_copy threads={threads}: {nbytes / min(ts) / 1e9:.1f} GB/s (best of 3), {nbytes / np.median(ts) / 1e9:.1f} median")
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
n2 = nbytes // (256 * 256 * 2)
ds = ctx.load('memory', data=src.reshape((n2 // 256, 256, 256, 256)), sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
for _ in range(2):
    ctx.run_udf(dataset=ds, udf=udf)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf); ts.append(time.perf_counter() - t0)
print(f"staged run_udf: {nbytes / np.median(ts) / 1e9:.1f} GB/s median, {nbytes / min(ts) / 1e9:.1f} best; cpu {os.sched_getaffinity(0).__len__()} cores")
try:
    print(open('/proc/self/numa_maps').read().count('\n'), 'numa_maps lines;', [l for l in open('/proc/self/status') if 'Mems_allowed_list' in l or 'Cpus_allowed_list' in l])
except Exception as e:
    print(e)
