"""Cold-start cost of a run: the first Context.run_udf (mask factories, device images, plan) against the second, per config"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd import masks as M
ctx = Context.make_with('hip', gpus=0)
rng = np.random.default_rng(0)
def go(name, ds, make):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = make(); torch.cuda.synchronize(); t1 = time.perf_counter()
    r = make(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:50s} first run {1e3 * (t1 - t0):9.1f} ms   second {1e3 * (t2 - t1):8.2f} ms")
u16 = torch.randint(0, 4096, (64, 256, 256, 256), device='cuda', dtype=torch.int16)
ds2 = ctx.load('memory', data=u16, dtype=np.dtype('uint16'), sig_dims=2, num_partitions=1)
masks = rng.random((16, 256, 256)).astype(np.float32)
udf2 = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
go('C2 16 dense masks, 16384 frames', ds2, lambda: ctx.run_udf(dataset=ds2, udf=udf2))
def rings():
    return M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256, n_bins=1024, use_sparse=True, dtype=np.float32)
udf4 = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=1024, mask_dtype=np.float32)
go('C4 1024 sparse rings, 16384 frames', ds2, lambda: ctx.run_udf(dataset=ds2, udf=udf4))
an3 = ctx.create_com_analysis(dataset=ds2, cx=128, cy=128)
go('CoM analysis 256x256, 16384 frames', ds2, lambda: ctx.run(an3))
del u16, ds2
f32 = torch.rand((8, 128, 1024, 1024), device='cuda')
ds5 = ctx.load('memory', data=f32, dtype=np.dtype('float32'), sig_dims=2, num_partitions=1)
an5 = ctx.create_radial_fourier_analysis(dataset=ds5)
go('C5 radial Fourier defaults, 1024 frames', ds5, lambda: ctx.run(an5))
an5s = ctx.create_radial_fourier_analysis(dataset=ds5, n_bins=16, max_order=24, use_sparse=True)
go('radial Fourier 16 bins sparse, 1024 frames', ds5, lambda: ctx.run(an5s))
