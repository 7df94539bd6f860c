"""Row-mirror fold of dense stacks (csrc/ltmi_fold.inc): the default radial-Fourier stack of C5 (25 complex masks on
1024 x 1024 float32 frames) through k_dense_fold against the unfolded kernel (tuning 38), checked against a float64
product and -- one-pixel frames -- element-wise against the stored weights.

    python scripts/bench_fold.py [--frames 8192] [--sig 1024] [--bins 1] [--order 24]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm
from libertem_amd.analysis.radialfourier import radial_mask_factory

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=8192)
ap.add_argument('--sig', type=int, default=1024)
ap.add_argument('--bins', type=int, default=1)
ap.add_argument('--order', type=int, default=24)
ap.add_argument('--reps', type=int, default=7)
ap.add_argument('--onepx', type=int, default=4096, help='one-pixel frames of the element-wise check')
args = ap.parse_args()
N = args.sig
ro = pm.bounding_radius(N / 2, N / 2, N, N)
stack = radial_mask_factory(N, N, N / 2, N / 2, 0, ro, args.bins, args.order, False)()
nm = stack.shape[0]
flat = np.ascontiguousarray(stack.reshape(nm, -1))
h = hip.MaskHandle.dense(0, flat, np.complex64)
h.set_sig_shape(N, N)
n_px = N * N
tile = torch.rand((args.frames, n_px), device='cuda', dtype=torch.float32)
out = torch.zeros((args.frames, nm), device='cuda', dtype=torch.complex64)
idx = torch.arange(0, args.frames, max(1, args.frames // 16), device='cuda')[:16]
mt = torch.from_numpy(flat).to('cuda')
ref = tile[idx].to(torch.complex128) @ mt.to(torch.complex128).T
res = {}
variants = [('folded', 30), ('unfolded (tuning 38)', 38), ('folded again', 30)]
if os.environ.get('FOLD_ABLATE'):
    variants += [('folded, no frame loads (timing only)', 31), ('folded, no MFMA (timing only)', 32)]
for name, code in variants:
    h.set_tuning(0, code, 0)
    out.zero_()
    for _ in range(2):
        h.apply(tile.data_ptr(), np.float32, args.frames, n_px, out.data_ptr(), nm, False)
    torch.cuda.synchronize()
    err = ((out[idx].to(torch.complex128) - ref).abs().max() / ref.abs().max()).item()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for a, b in evs:
        a.record(); h.apply(tile.data_ptr(), np.float32, args.frames, n_px, out.data_ptr(), nm, False); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    gb = args.frames * n_px * 4 / ms / 1e6
    tf = 4.0 * args.frames * n_px * nm / ms / 1e9
    print(f"{name:22s} {h.last_kernel():70s} {ms:8.3f} ms  {args.frames / ms / 1e3:6.3f} Mframes/s  {gb:6.0f} GB/s "
          f"({gb / 8000:.3f} of HBM)  {tf:6.1f} TFLOP/s algorithmic ({tf / 157.3:.3f} of the f32 matrix peak)  "
          f"err vs float64 {err:.2e}")
    res[name] = out[idx].clone()
    assert err < 1e-5 or code in (31, 32), err
d = (res['folded'] - res['unfolded (tuning 38)']).abs().max().item() / ref.abs().max().item()
print(f"folded vs unfolded: {d:.2e} of max|result|")

# element-wise: frames with ONE non-zero pixel give pixel * weight for every column, rtol 1e-5, atol 0
rng = np.random.default_rng(7)
px = np.unique(np.concatenate([rng.integers(0, n_px, max(1024, args.onepx - 6 * N)),
                               np.arange(N), np.arange(N) + (N // 2) * N, np.arange(N) + (N - 1) * N,     # rows 0, centre, last
                               np.arange(N) + (N // 2 - 1) * N, np.arange(N) + (N // 2 + 1) * N, np.arange(N) * N + N // 2]))
vals = rng.random(len(px)).astype(np.float32) + 0.5
one = torch.zeros((len(px), n_px), device='cuda', dtype=torch.float32)
one[torch.arange(len(px), device='cuda'), torch.from_numpy(px).to('cuda')] = torch.from_numpy(vals).to('cuda')
o1 = torch.zeros((len(px), nm), device='cuda', dtype=torch.complex64)
h.set_tuning(0, 30, 0)
h.apply(one.data_ptr(), np.float32, len(px), n_px, o1.data_ptr(), nm, False)
torch.cuda.synchronize()
print('one-pixel frames through', h.last_kernel())
got = o1.cpu().numpy()
want = (flat[:, px].T.astype(np.complex128) * vals[:, None].astype(np.float64))
for part in (np.real, np.imag):
    g, w = part(got).astype(np.float64), part(want)
    bad = np.abs(g - w) > 1e-5 * np.abs(w)
    print(f"  element-wise {part.__name__}: {bad.sum()} of {bad.size} entries off by more than 1e-5 relative "
          f"(max rel {np.max(np.abs(g - w) / np.maximum(np.abs(w), 1e-300)):.2e})")
    assert not bad.any()
