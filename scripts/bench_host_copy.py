"""ltmi_host_copy: pageable -> page-locked staging copy, GB/s by thread count (the upload path's bounce buffers)
    python scripts/bench_host_copy.py [MiB = 1024]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = np.random.default_rng(0).integers(0, 255, mib << 20, dtype=np.uint8)
dst = torch.empty(mib << 20, dtype=torch.uint8, pin_memory=True).numpy()
print(f"cores: {os.cpu_count()}  buffer {mib} MiB")
for th in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 0):
    hip.host_copy(dst, src, th)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); hip.host_copy(dst, src, th); ts.append(time.perf_counter() - t0)
    print(f"  threads {th:3d}: {src.nbytes / min(ts) / 1e9:7.1f} GB/s (best), {src.nbytes / np.median(ts) / 1e9:7.1f} (median)")
assert np.array_equal(dst[::4097], src[::4097])
t0 = time.perf_counter(); dst[...] = src; t = time.perf_counter() - t0
print(f"  numpy assignment: {src.nbytes / t / 1e9:.1f} GB/s")
