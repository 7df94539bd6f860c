"""CrystallinityUDF (batched rfft2 + ring integration) on a C2-sized scan, device-resident."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.crystallinity import CrystallinityUDF

scan = int(os.environ.get('SCAN', 256))
sig = int(os.environ.get('SIG', 256))
ctx = Context.make_with('hip', gpus=0)
g = torch.Generator(device='cuda').manual_seed(1)
frames = torch.randint(0, 4096, (scan, scan, sig, sig), generator=g, device='cuda', dtype=torch.int16)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = CrystallinityUDF(rad_in=sig // 16, rad_out=sig // 4, real_center=(sig / 2, sig / 2),
                       real_rad=sig // 10)
for _ in range(2):
    r = ctx.run_udf(dataset=ds, udf=udf)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    r = ctx.run_udf(dataset=ds, udf=udf)
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
n = scan * scan
print(f"CrystallinityUDF {n} frames {sig}x{sig} u16: {t * 1e3:.2f} ms per run, {n / t / 1e6:.3f} Mframes/s, "
      f"{n * sig * sig * 2 / t / 1e9:.0f} GB/s input")
# spot check of 3 frames against numpy in float64
idx = [0, n // 2, n - 1]
yy, xx = np.ogrid[-sig / 2:sig / 2, -sig / 2:sig / 2]
rm = 1 - 1 * (yy * yy + xx * xx <= (sig // 10) ** 2)
ring = 1 * (yy * yy + xx * xx <= (sig // 4) ** 2) - 1 * (yy * yy + xx * xx <= (sig // 16) ** 2)
half = np.fft.fftshift(ring)[:, :sig // 2 + 1]
got = r['intensity'].raw_data.reshape(-1)
for i in idx:
    f = frames.reshape(n, sig, sig)[i].cpu().numpy().view(np.uint16).astype(np.float64)
    ref = np.sum(abs(np.fft.rfft2(f * rm)) * half)
    print(f"   frame {i}: {got[i]:.6e} vs float64 {ref:.6e}  rel {abs(got[i] - ref) / ref:.2e}")
    assert abs(got[i] - ref) / ref < 1e-5
