"""What the non-finite guard (csrc/ltmi_guard.hip) costs: float32 frames against sparse stacks on the fast kernels,
(a) clean frames, guard off (LTMI_NONFINITE_GUARD=0 in the environment of this process) / on,
(b) 1 % of the frames with a NaN in a stored pixel, (c) EVERY frame with a NaN in a pixel no mask stores.
Run once per guard setting; HIP-event times through hip.KernelTimer (fast kernel + the guard's launches).

    python scripts/bench_nonfinite_guard.py [c4f32] [wide64] [rf16]
"""
import os, sys
import numpy as np, torch
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
from libertem_amd import masks as M
from libertem_amd.common.container import MaskContainer
from libertem_amd.common.slice import Slice
from libertem_amd.common.shape import Shape

which = sys.argv[1:] or ['c4f32', 'wide64', 'rf16']
guard = os.environ.get('LTMI_NONFINITE_GUARD', '1')


def handle_of(stack_factory, sig, dtype):
    mc = MaskContainer(stack_factory, dtype=dtype, use_sparse='scipy.sparse', backend='hip')
    sl = Slice(origin=(0,) * len(sig), shape=Shape(tuple(sig), sig_dims=len(sig)))
    rd = np.result_type(np.float32, dtype)
    h = mc.get_handle_for_sig_slice(sl, rd, 0, frame_dtype=np.float32)
    csr = sp.csr_matrix(mc.get_for_sig_slice(sl, dtype=rd, sparse_backend='scipy.sparse.csr', transpose=True))
    return mc, h, csr, rd


def timed(h, t, n, n_px, out, n_masks, reps=5):
    for _ in range(2):
        h.apply(t.data_ptr(), np.float32, n, n_px, out.data_ptr(), n_masks, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        h.apply(t.data_ptr(), np.float32, n, n_px, out.data_ptr(), n_masks, False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def case(name, factory, sig, dtype, n):
    mc, h, csr, rd = handle_of(factory, sig, dtype)
    n_px, n_masks = csr.shape
    stored = np.flatnonzero(np.diff(csr.indptr) > 0)
    unstored = np.flatnonzero(np.diff(csr.indptr) == 0)
    t = torch.rand((n, n_px), device='cuda')
    out = torch.zeros((n, n_masks), device='cuda', dtype=torch.complex64 if np.dtype(rd).kind == 'c' else torch.float32)
    clean = timed(h, t, n, n_px, out, n_masks)
    kern = h.last_kernel()
    line = f"{name}: guard={guard} clean {clean:.3f} ms"
    if guard != '0':
        idx = torch.arange(0, n, 100, device='cuda')
        t[idx, int(stored[len(stored) // 2])] = float('nan')
        one = timed(h, t, n, n_px, out, n_masks)
        line += f" | 1% of the frames NaN in a stored pixel {one:.3f} ms"
        if len(unstored):
            t = torch.rand((n, n_px), device='cuda')
            t[:, int(unstored[len(unstored) // 2])] = float('nan')
            allf = timed(h, t, n, n_px, out, n_masks)
            fin = bool(torch.isfinite(torch.view_as_real(out) if out.is_complex() else out).all())
            line += f" | every frame NaN in an unstored pixel {allf:.3f} ms (results finite: {fin})"
    print(line)
    print("    ", kern, flush=True)
    mc.close()


if 'c4f32' in which:
    case('C4 stack (1024 rings, 256x256) on 16384 float32 frames',
         lambda: M.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32), (256, 256), np.float32, 16384)
if 'wide64' in which:
    case('64 wide rings off centre (densified) on 16384 float32 frames of 256x256',
         lambda: M.radial_bins(120.3, 131.7, 256, 256, radius=110, n_bins=64, use_sparse=True, dtype=np.float32), (256, 256), np.float32, 16384)
    case('48 wide rings about the centre (densified + folded) on 16384 float32 frames of 256x256',
         lambda: M.radial_bins(128, 128, 256, 256, radius=110, n_bins=48, use_sparse=True, dtype=np.float32), (256, 256), np.float32, 16384)
if 'rf16' in which:
    from libertem_amd.analysis.radialfourier import radial_mask_factory
    case('radial Fourier 16 bins x 25 orders, use_sparse (banded) on 2048 float32 frames of 1024x1024',
         radial_mask_factory(1024, 1024, 512., 512., 0., 480., 16, 24, True), (1024, 1024), np.complex64, 2048)
