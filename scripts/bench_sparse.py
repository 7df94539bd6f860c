"""Micro-benchmark of the sparse (SELL) kernel: C4 = 1024 anti-aliased ring masks, 256x256 uint16."""
import argparse
import sys
import os
import numpy as np
import scipy.sparse as sp
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=16384)
ap.add_argument('--bins', type=int, default=1024)
ap.add_argument('--reps', type=int, default=10)
args = ap.parse_args()
rings = pm.radial_bins(128, 128, 256, 256, n_bins=args.bins, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
print('nnz', csr.nnz)
h = hip.MaskHandle.csr(0, csr, np.float32)
g = torch.Generator(device='cuda').manual_seed(1)
tile = torch.randint(0, 4096, (args.frames, 65536), generator=g, device='cuda',
                     dtype=torch.int32).to(torch.int16)
out = torch.zeros((args.frames, args.bins), device='cuda', dtype=torch.float32)
for _ in range(2):
    h.apply(tile.data_ptr(), np.uint16, args.frames, 65536, out.data_ptr(), args.bins, False)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
       for _ in range(args.reps)]
for a, b in evs:
    a.record()
    h.apply(tile.data_ptr(), np.uint16, args.frames, 65536, out.data_ptr(), args.bins, False)
    b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs)
med = ts[len(ts) // 2]
fb = 131072 + args.bins * 4
print(f"{h.last_kernel()} median {med:.3f} ms  {args.frames / med / 1e3:.2f} Mframes/s  "
      f"{args.frames * fb / med / 1e6:.0f} GB/s ({args.frames * fb / med / 1e6 / 80:.1f}% of 8 TB/s) "
      f"{2 * csr.nnz * args.frames / med / 1e9:.1f} GFLOP/s")
