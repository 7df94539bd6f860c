"""Micro-benchmark of the two sparse kernels on C4 (1024 anti-aliased ring masks, 256x256 uint16):
the blocked image on the matrix cores (k_bell_apply) and the SELL gather kernel (k_sell_apply),
each checked against float64 NumPy on a few frames.

    python scripts/bench_sparse.py [--frames 16384] [--dtype uint16|int16|float32|uint8]
"""
import argparse
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=16384)
ap.add_argument('--bins', type=int, default=1024)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--dtype', default='uint16')
ap.add_argument('--only', type=int, default=0, help='tuning code of the one variant to run')
ap.add_argument('--keep', default='all', choices=['all', 'quarter', 'half'],
                help="timing model of a mirror-folded stack: keep only the entries of a quarter (x >= cx, y even) "
                     "or half (y even) of the pixels -- the frames (and their copies) stay whole")
ap.add_argument('--pad', type=int, default=0, help='pixels of padding at the end of every frame row (frame stride = 65536 + pad)')
args = ap.parse_args()
rings = pm.radial_bins(128, 128, 256, 256, n_bins=args.bins, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
if args.keep != 'all':
    yy, xx = np.divmod(np.arange(65536), 256)
    keep = (yy % 2 == 0) & ((xx >= 128) if args.keep == 'quarter' else True)
    import scipy.sparse as sps
    csr = sps.diags(keep.astype(np.float32)).dot(csr).tocsr()
    csr.eliminate_zeros()
    csr.sort_indices()
print('nnz', csr.nnz)
h = hip.MaskHandle.csr(0, csr, np.float32)
dt = np.dtype(args.dtype)
g = torch.Generator(device='cuda').manual_seed(1)
if dt == np.float32:
    tile = torch.rand((args.frames, 65536), generator=g, device='cuda')
elif dt.itemsize == 1:
    tile = torch.randint(0, 100, (args.frames, 65536), generator=g, device='cuda',
                         dtype=torch.int32).to(torch.uint8)
else:
    tile = torch.randint(0, 4096, (args.frames, 65536), generator=g, device='cuda',
                         dtype=torch.int32).to(torch.int16)
if args.pad:
    padded = torch.zeros((args.frames, 65536 + args.pad), device='cuda', dtype=tile.dtype)
    padded[:, :65536] = tile
    tile = padded[:, :65536]
LD = 65536 + args.pad
out = torch.zeros((args.frames, args.bins), device='cuda', dtype=torch.float32)
dense = csr.toarray().astype(np.float64)                 # (n_px, n_masks)
check = [0, 17, args.frames - 1]
ref = tile[check].cpu().numpy().astype(np.float64) @ dense
fb = 65536 * dt.itemsize + args.bins * 4
for code, name in ((40, 'as dispatched'), (42, 'blocked image (tuning 42)'), (41, 'SELL kernel')):
    if args.only and code != args.only:
        continue
    h.set_tuning(0, code, 0)
    out.zero_()
    for _ in range(2):
        h.apply(tile.data_ptr(), dt, args.frames, LD, out.data_ptr(), args.bins, False)
    torch.cuda.synchronize()
    got = out[check].cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.reps)]
    for a, b in evs:
        a.record()
        h.apply(tile.data_ptr(), dt, args.frames, LD, out.data_ptr(), args.bins, False)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    print(f"{name}: {h.last_kernel()}\n   rel err {err:.2e}  median {med:.3f} ms  "
          f"{args.frames / med / 1e3:.2f} Mframes/s  {args.frames * fb / med / 1e6:.0f} GB/s "
          f"({args.frames * fb / med / 1e6 / 80:.1f}% of 8 TB/s)  "
          f"{2 * csr.nnz * args.frames / med / 1e9:.1f} GFLOP/s useful")
    assert err < 1e-5 or os.environ.get('LTMI_BENCH_NOCHECK'), err
