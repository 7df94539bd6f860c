"""C2 kernel with and without entries in the float32 tail of the float16-piece path (same box, same frames):
the benchmark's stack (drawn in float64, two weights go through the tail) against float32-drawn weights
(none do) and against the float32 instruction (tuning 37)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

n = 65536
frames = torch.randint(0, 4096, (n, 65536), device='cuda', dtype=torch.int32).to(torch.int16)
out = torch.zeros((n, 16), device='cuda', dtype=torch.float32)
stacks = {
    'float64-drawn (bench)': np.random.default_rng(2).random((16, 65536)).astype(np.float32),
    'float32-drawn': np.random.default_rng(2).random((16, 65536), dtype=np.float32),
}
for name, m in stacks.items():
    for tuning in (0, 37):
        h = hip.MaskHandle.dense(0, m, np.float32)
        if tuning:
            h.set_tuning(0, tuning, 0)
        for _ in range(5):
            h.apply(frames.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            h.apply(frames.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        med = sorted(ts)[len(ts) // 2]
        print(f"{name:24s} tuning {tuning:2d}: {h.last_kernel()}  median {med:.4f} ms  "
              f"{n * 131136 / med / 1e6 / 8000:.3f} of HBM")
        h.close()
