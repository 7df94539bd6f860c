"""SURVEY.md 8(d)'s second runs: C3 with mask_radius=200, C5 with n_bins=16, max_order=24, use_sparse=True (400 complex64 masks)
-- whole job through Context.run, kernels through hip.KernelTimer, result against float64 NumPy on a few frames"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libertem_amd.api as lt
from libertem_amd import hip

ctx = lt.Context.make_with('hip', gpus=0)
which = sys.argv[1:] or ['c3r', 'c5s', 'c5s_u16']


def run(name, an, frames, n):
    res = ctx.run(an)
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.run(an)
    hip.KernelTimer.start()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.run(an)
    dt = (time.perf_counter() - t0) / 3
    ev = hip.KernelTimer.stop()
    by = {}
    for ms, cnt, k in ev:
        by.setdefault(k, []).append(ms)
    print(f"{name}: {dt * 1e3:.2f} ms per run = {n / dt / 1e6:.3f} M frames/s = {frames.numel() * frames.element_size() / dt / 1e9:.0f} GB/s")
    for k, v in by.items():
        print(f"    {len(v) // 3} x {np.mean(v):.3f} ms  {k}")
    return res


if 'c3r' in which:
    n = 16384
    fr = torch.randint(0, 4096, (n, 512, 512), device='cuda', dtype=torch.int16)
    ds = ctx.load('memory', data=fr.reshape((64, 256, 512, 512)), dtype=np.dtype('uint16'), sig_dims=2, num_partitions=1)
    an = ctx.create_com_analysis(dataset=ds, cx=256, cy=256, mask_radius=200)
    res = run('C3, mask_radius=200 (16384 frames of 512x512 uint16)', an, fr, n)
    yy, xx = np.mgrid[0:512, 0:512]
    disk = ((yy - 256) ** 2 + (xx - 256) ** 2) <= 200 ** 2
    for i in (0, 777, n - 1):
        f = fr[i].cpu().numpy().view(np.uint16).astype(np.float64) * disk
        cy, cx = (f * yy).sum() / f.sum() - 256, (f * xx).sum() / f.sum() - 256
        print('    check', abs(res.y.raw_data.reshape(-1)[i] - cy), abs(res.x.raw_data.reshape(-1)[i] - cx))
    del fr, ds, an, res
for key, dtype in (('c5s', torch.float32), ('c5s_u16', torch.int16)):
    if key not in which:
        continue
    n = int(os.environ.get('C5S_FRAMES', 8192))
    if dtype == torch.float32:
        fr = torch.rand((n, 1024, 1024), device='cuda')
        npdt = np.dtype('float32')
    else:
        fr = torch.randint(0, 4096, (n, 1024, 1024), device='cuda', dtype=torch.int16)
        npdt = np.dtype('uint16')
    ds = ctx.load('memory', data=fr.reshape((n // 128, 128, 1024, 1024)), dtype=npdt, sig_dims=2, num_partitions=1)
    NB = int(os.environ.get('C5S_BINS', 16))
    SP = {'1': True, '0': False}.get(os.environ.get('C5S_SPARSE', '1'))
    an = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=NB, max_order=24, use_sparse=SP)
    res = run(f'C5, n_bins={NB}, max_order=24, use_sparse=True ({n} frames of 1024x1024 {npdt})', an, fr, n)
    stack = an.get_mask_factories()()
    stack = stack.todense() if hasattr(stack, 'todense') else np.asarray(stack)
    print('    use_sparse', an.parameters['use_sparse'])
    stack = np.asarray(stack).reshape((25 * NB, -1)).astype(np.complex128)
    raw = res.raw_results.reshape((25 * NB, -1))
    for i in (0, n - 1):
        f = fr[i].cpu().numpy()
        f = (f.view(np.uint16) if npdt == np.dtype('uint16') else f).astype(np.float64).reshape(-1)
        ref = stack @ f
        print('    check rel err', np.abs(raw[:, i] - ref).max() / np.abs(ref).max())
    del fr, ds, an, res
