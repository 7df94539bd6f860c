"""ltmi_crystallinity alone (no UDF machinery): N frames of 256 x 256 resident in HBM, HIP events around
REPS calls.  N=16384 REPS=10 RAD_OUT=64 DTYPE=uint16 MASK=1; LTMI_FFT_FUSED=0 -> the hipFFT route."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
from libertem_amd.udf.crystallinity import crystallinity_masks, mask_box

n = int(os.environ.get('N', 16384))
sig = int(os.environ.get('SIG', 256))
reps = int(os.environ.get('REPS', 10))
rad_out = float(os.environ.get('RAD_OUT', 64))
rad_out = int(rad_out) if rad_out == int(rad_out) else rad_out
dt = np.dtype(os.environ.get('DTYPE', 'uint16'))
use_mask = os.environ.get('MASK', '1') != '0'
g = torch.Generator(device='cuda').manual_seed(1)
tdt = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.float64}[dt.itemsize]
if dt.kind == 'f':
    frames = torch.rand((n, sig, sig), generator=g, device='cuda', dtype=tdt if dt.itemsize == 8 else torch.float32) * 4096
else:
    frames = torch.randint(0, 200 if dt.itemsize == 1 else 4096, (n, sig, sig), generator=g, device='cuda',
                           dtype=tdt)
real_mask, half = crystallinity_masks((sig, sig), int(rad_out) // 4, rad_out,
                                      (sig // 2, sig // 2) if use_mask else None, sig // 10 if use_mask else None)
rm = None if real_mask is None else torch.from_numpy(np.ascontiguousarray(real_mask.astype(np.float32))).cuda()
hm = torch.from_numpy(np.ascontiguousarray(half.astype(np.float32))).cuda()
out = torch.zeros(n, dtype=torch.float32, device='cuda')
plan = hip.FFTPlan(0, sig, sig, min(n, int(os.environ.get("BATCH", 1024))))
box = mask_box(half)


tables = None
if os.environ.get('CORR', '0') != '0':                      # dark + gain + 50 dead pixels, fused into the conversion pass
    from libertem_amd.io.corrections import CorrectionSet
    rng = np.random.default_rng(3)
    bad = np.zeros((sig, sig), dtype=bool)
    bad[rng.integers(0, sig, 50), rng.integers(0, sig, 50)] = True
    corr_set = CorrectionSet(dark=rng.random((sig, sig)) * 6, gain=rng.random((sig, sig)) * 0.6 + 0.7, excluded_pixels=bad)
    tables = corr_set.device_tables(0, (sig, sig))


def call():
    if tables is not None:
        plan.crystallinity_corrected(frames.data_ptr(), dt, n, sig * sig, tables, None if rm is None else rm.data_ptr(),
                                     hm.data_ptr(), box, out.data_ptr(), False)
        return
    plan.crystallinity(frames.data_ptr(), dt, n, sig * sig, None if rm is None else rm.data_ptr(),
                       hm.data_ptr(), box, out.data_ptr(), False)


call(); call()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    call()
    ev[i + 1].record()
torch.cuda.synchronize()
ms = np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])
print(f"{plan.last_kernel()}: {n} frames {dt} in {ms:.3f} ms = {n / ms / 1e3:.2f} M frames/s, "
      f"{n * sig * sig * dt.itemsize / ms / 1e6:.0f} GB/s of pixels")
if os.environ.get('LTMI_CRYST_ABLATE') or tables is not None:
    sys.exit(0)        # timing-only variant: garbage results
i = [0, n // 2, n - 1]
fr = frames[i].cpu().numpy()
fr = fr.view(dt) if fr.dtype != dt else fr
got = out[i].cpu().numpy()
for j, fj in enumerate(fr.astype(np.float64)):
    ref = np.sum(abs(np.fft.rfft2(fj * real_mask if real_mask is not None else fj)) * half)
    assert abs(got[j] - ref) <= 1e-5 * abs(ref), (got[j], ref)
print("   3 frames checked against float64: ok")
