"""515 x 515 uint16 frames (rows at every 2-byte alignment): kernel-level times of the operators with
and without LTMI_ALIGNED_DMA_ONLY=1 (the round-1 dispatch: vector / DMA paths for 16-B aligned rows only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm

nf, h, w = 16384, 515, 515
n_px = h * w
g = torch.Generator(device='cuda').manual_seed(1)
tile = torch.randint(0, 4096, (nf, n_px), generator=g, device='cuda', dtype=torch.int32).to(torch.int16)
dt = np.dtype('uint16')
fb = n_px * 2


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


masks = np.random.default_rng(2).random((16, n_px)).astype(np.float32)
hd = hip.MaskHandle.dense(0, masks, np.float32)
out = torch.zeros((nf, 16), device='cuda')
ms = timeit(lambda: hd.apply(tile.data_ptr(), dt, nf, n_px, out.data_ptr(), 16, False))
print(f"dense 16 masks : {ms:8.3f} ms  {nf * fb / ms / 1e6:6.0f} GB/s  {hd.last_kernel()}")

rings = pm.radial_bins(257, 257, w, h, n_bins=512, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
hs = hip.MaskHandle.csr(0, csr, np.float32)
outs = torch.zeros((nf, 512), device='cuda')
ms = timeit(lambda: hs.apply(tile.data_ptr(), dt, nf, n_px, outs.data_ptr(), 512, False))
ref = tile[:2].cpu().numpy().view(np.uint16).astype(np.float64) @ csr.toarray().astype(np.float64)
err = np.abs(outs[:2].cpu().numpy() - ref).max() / np.abs(ref).max()
print(f"sparse 512 rings: {ms:8.3f} ms  {nf * fb / ms / 1e6:6.0f} GB/s  rel err {err:.1e}  {hs.last_kernel()}")

ss = torch.zeros((nf,), device='cuda')
ms = timeit(lambda: hip.sum_sig(0, tile.data_ptr(), dt, nf, n_px, n_px, ss.data_ptr(), np.float32, False))
print(f"sum over sig   : {ms:8.3f} ms  {nf * fb / ms / 1e6:6.0f} GB/s")
