"""Micro-benchmark of ltmi_apply_masks on device-resident frames (kernel-level, no UDF runtime)."""
import argparse
import time
import numpy as np
import torch
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=65536)
ap.add_argument('--sig', type=int, default=256)
ap.add_argument('--masks', type=int, default=16)
ap.add_argument('--dtype', default='uint16')
ap.add_argument('--mask-dtype', default='float32')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--variants', default='auto')
args = ap.parse_args()

n_px = args.sig * args.sig
dt = np.dtype(args.dtype)
g = torch.Generator(device='cuda').manual_seed(1)
if dt.kind in 'iu':
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[dt.itemsize]
    tile = torch.randint(0, 4096 if dt.itemsize > 1 else 200, (args.frames, n_px), generator=g,
                         device='cuda', dtype=torch.int32).to(tdt)
else:
    tile = torch.rand((args.frames, n_px), generator=g, device='cuda',
                      dtype=torch.float64 if dt.itemsize == 8 else torch.float32)
rng = np.random.default_rng(2)
md = np.dtype(args.mask_dtype)
if md.kind == 'c':
    masks = (rng.random((args.masks, n_px)) + 1j * rng.random((args.masks, n_px))).astype(md)
else:
    masks = rng.random((args.masks, n_px)).astype(md)
res_dt = np.result_type(dt, md)
h = hip.MaskHandle.dense(0, masks, res_dt)
out = torch.zeros((args.frames, args.masks), device='cuda',
                  dtype={'complex64': torch.complex64, 'float32': torch.float32,
                         'float64': torch.float64, 'complex128': torch.complex128}[res_dt.name])
frame_bytes = n_px * dt.itemsize + args.masks * res_dt.itemsize

if args.variants == 'auto':
    variants = [dict(mt=0, waves=0, ksplit=0)]
elif args.variants.startswith('mt='):       # explicit list of `mt` codes (f64 results: 1 = direct-load kernel)
    variants = [dict(mt=int(w), waves=0, ksplit=0) for w in args.variants[3:].split(',')]
elif args.variants.startswith('w='):        # explicit list of `waves` codes, e.g. w=0,31,32
    variants = [dict(mt=0, waves=int(w), ksplit=0) for w in args.variants[2:].split(',')]
else:
    variants = [dict(mt=0, waves=0, ksplit=0), dict(mt=0, waves=31, ksplit=0),
                dict(mt=0, waves=32, ksplit=0),
                dict(mt=2, waves=4, ksplit=1), dict(mt=1, waves=4, ksplit=1)]
ref_out = None
for v in variants:
    h.set_tuning(**v)
    for _ in range(3):
        h.apply(tile.data_ptr(), dt, args.frames, n_px, out.data_ptr(), args.masks, False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.reps)]
    for a, b in evs:
        a.record()
        h.apply(tile.data_ptr(), dt, args.frames, n_px, out.data_ptr(), args.masks, False)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    if ref_out is None:
        ref_out = out.clone()
    else:
        err = float((torch.view_as_real(out) if out.is_complex() else out).sub(
            torch.view_as_real(ref_out) if out.is_complex() else ref_out).abs().max())
        print(f"   max |diff| to the first variant: {err:.3e} (scale {float(ref_out.abs().max()):.3e})")
    gbs = args.frames * frame_bytes / (med * 1e-3) / 1e9
    print(f"{v} {h.last_kernel()}  median {med:.3f} ms  min {ts[0]:.3f} ms  "
          f"{args.frames / (med * 1e-3) / 1e6:.2f} Mframes/s  {gbs:.0f} GB/s "
          f"({gbs / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
