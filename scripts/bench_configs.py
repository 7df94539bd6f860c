"""BASELINE.json configs C1/C3/C4/C5 through the public API (Context.run / run_udf) on one MI355X,
device-resident frames, plus a spot check of a few frames against float64 NumPy.

    python scripts/bench_configs.py c3 [--scan 512] [--reps 5]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('config', choices=['c1', 'c3', 'c3r', 'c4', 'c5', 'c5s'])
ap.add_argument('--scan', type=int, default=0, help='scan edge (0 = the BASELINE.json size)')
ap.add_argument('--reps', type=int, default=5)
args = ap.parse_args()

import torch
from libertem_amd.api import Context
from libertem_amd import hip, masks as M
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd.udf.sum import SumUDF

HBM = 8000.0


def device_frames(n_frames, sig, dtype, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    if dtype == np.uint16:
        t = torch.empty((n_frames, sig, sig), dtype=torch.int16, device='cuda')
        step = max(1, (2 << 30) // (sig * sig * 2))
        for i in range(0, n_frames, step):
            n = min(step, n_frames - i)
            t[i:i + n] = torch.randint(0, 4096, (n, sig, sig), generator=g, device='cuda',
                                       dtype=torch.int16)
        return t
    t = torch.empty((n_frames, sig, sig), dtype=torch.float32, device='cuda')
    step = max(1, (2 << 30) // (sig * sig * 4))
    for i in range(0, n_frames, step):
        n = min(step, n_frames - i)
        t[i:i + n] = torch.rand((n, sig, sig), generator=g, device='cuda')
    return t


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    hip.KernelTimer.enabled = True
    hip.KernelTimer.events.clear()
    for _ in range(reps):
        t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    hip.KernelTimer.enabled = False
    kms = {}
    for a, b, n, name in hip.KernelTimer.events:
        kms.setdefault(name, []).append(a.elapsed_time(b))
    return res, float(np.median(ts)), {k: (float(np.median(v)), len(v) // reps) for k, v in kms.items()}


def report(name, n_frames, frame_bytes, result_bytes, wall, kernels):
    fps = n_frames / wall
    print(f"{name}: {n_frames} frames, run wall {wall * 1e3:.2f} ms -> {fps / 1e6:.3f} Mframes/s whole job, "
          f"{fps * frame_bytes / 1e9:.0f} GB/s input")
    for k, (ms, per_run) in kernels.items():
        per_launch = n_frames / max(1, per_run)
        gbs = per_launch * (frame_bytes + result_bytes) / (ms * 1e-3) / 1e9
        print(f"   kernel {k}: {per_run} launch(es)/run, median {ms:.3f} ms -> {gbs:.0f} GB/s "
              f"= {gbs / HBM:.3f} of HBM peak")


ctx = Context.make_with('hip', gpus=0)
if args.config == 'c1':
    # plumbing config: SumUDF 32x32 x 128x128 f32 -- on the HIP executor (device resident)
    scan = args.scan or 32
    fr = device_frames(scan * scan, 128, np.float32, 0)
    ds = ctx.load('memory', data=fr.reshape(scan, scan, 128, 128), sig_dims=2, num_partitions=1)
    res, wall, k = timed(lambda: ctx.run_udf(dataset=ds, udf=SumUDF()), args.reps)
    ref = fr.sum(dim=0).cpu().numpy()
    err = np.abs(res['intensity'].data - ref).max() / np.abs(ref).max()
    print(f"C1 check vs torch sum: rel err {err:.2e}")
    report("C1 SumUDF", scan * scan, 128 * 128 * 4, 0, wall, k)
elif args.config in ('c3', 'c3r'):
    scan = args.scan or 512
    sig = 512
    fr = device_frames(scan * scan, sig, np.uint16, 3)
    ds = ctx.load('memory', data=fr.reshape(scan, scan, sig, sig), dtype=np.uint16, sig_dims=2,
                  num_partitions=1)
    kw = dict(cx=256, cy=256)
    if args.config == 'c3r':
        kw['mask_radius'] = 200
    an = ctx.create_com_analysis(dataset=ds, **kw)
    res, wall, k = timed(lambda: ctx.run(an), args.reps)
    # spot check: centre of mass of 3 frames in float64
    yy, xx = np.mgrid[0:sig, 0:sig]
    w = np.ones((sig, sig)) if args.config == 'c3' else ((yy - 256) ** 2 + (xx - 256) ** 2 <= 200 ** 2)
    for idx in (0, scan * scan // 2, scan * scan - 1):
        f = fr[idx].cpu().numpy().view(np.uint16).astype(np.float64) * w
        cy, cx = (f * yy).sum() / f.sum(), (f * xx).sum() / f.sum()
        gy = res.y.raw_data.reshape(-1)[idx] + 256
        gx = res.x.raw_data.reshape(-1)[idx] + 256
        print(f"   frame {idx}: com ({gy:.4f}, {gx:.4f}) vs float64 ({cy:.4f}, {cx:.4f})")
        assert abs(gy - cy) < 1e-2 and abs(gx - cx) < 1e-2
    report(f"{args.config.upper()} CoM", scan * scan, sig * sig * 2, 12, wall, k)
elif args.config == 'c4':
    scan = args.scan or 256
    sig = 256
    fr = device_frames(scan * scan, sig, np.uint16, 1)
    ds = ctx.load('memory', data=fr.reshape(scan, scan, sig, sig), dtype=np.uint16, sig_dims=2,
                  num_partitions=1)
    udf = ApplyMasksUDF(
        mask_factories=lambda: M.radial_bins(centerX=128, centerY=128, imageSizeX=256,
                                             imageSizeY=256, n_bins=1024, use_sparse=True,
                                             dtype=np.float32),
        use_sparse='scipy.sparse', mask_count=1024, mask_dtype=np.float32)
    res, wall, k = timed(lambda: ctx.run_udf(dataset=ds, udf=udf), args.reps)
    dense = M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256, n_bins=1024,
                          use_sparse=False, dtype=np.float32).reshape(1024, -1).astype(np.float64)
    got = res['intensity'].raw_data
    for idx in (0, scan * scan - 1):
        ref = dense @ fr[idx].cpu().numpy().view(np.uint16).reshape(-1).astype(np.float64)
        err = np.abs(got[idx] - ref).max() / np.abs(ref).max()
        print(f"   frame {idx}: rel err vs float64 {err:.2e}")
        assert err < 1e-5
    report("C4 1024 sparse ring masks", scan * scan, sig * sig * 2, 4096, wall, k)
else:
    scan = args.scan or 128
    sig = 1024
    fr = device_frames(scan * scan, sig, np.float32, 5)
    ds = ctx.load('memory', data=fr.reshape(scan, scan, sig, sig), sig_dims=2, num_partitions=1)
    kw = {}
    if args.config == 'c5s':
        kw['use_sparse'] = True
    an = ctx.create_radial_fourier_analysis(dataset=ds, cx=512, cy=512, n_bins=1, max_order=24, **kw)
    res, wall, k = timed(lambda: ctx.run(an), args.reps)
    print("   result:", type(res).__name__, [n for n in dir(res) if n.startswith('dominant')][:2])
    report(f"{args.config.upper()} RadialFourier 25 orders", scan * scan, sig * sig * 4, 200, wall, k)
    flops = 4.0 * sig * sig * 25 * scan * scan
    for name, (ms, per_run) in k.items():
        print(f"   {flops / max(1, per_run) / (ms * 1e-3) / 1e12:.1f} TFLOP/s algorithmic "
              f"(4 flop per real x complex FMA), f32 matrix peak 157")
