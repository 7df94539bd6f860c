import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
os.environ['LTMI_SPARSE_BAND'] = '1'
import numpy as np, torch, scipy.sparse as sp
from libertem_amd import hip
from libertem_amd.analysis.radialfourier import radial_mask_factory
import test_kernels_gpu as T
rng = np.random.default_rng(1)
stack = radial_mask_factory(128, 128, 64, 64, 4, 50, 3, 12, True)()
csr = sp.csr_matrix(stack.to_px_by_masks(dtype=np.complex64))
n_px, n_masks = csr.shape
stored = np.flatnonzero(np.diff(csr.indptr) > 0); unstored = np.flatnonzero(np.diff(csr.indptr) == 0)
data = T._dirty_frames(rng, 48, n_px, stored, unstored)
h = hip.MaskHandle.csr(0, csr, np.complex64); h.set_sig_shape(128, 128); h.set_dense_origin(csr)
t = T._dev(data); out = T._dev(np.full((48, n_masks), 7, np.complex64))
h.apply(t.data_ptr(), np.float32, 48, n_px, out.data_ptr(), n_masks, False); torch.cuda.synchronize()
res = out.cpu().numpy()
dense = np.asarray(csr.todense()).astype(np.complex128)
with np.errstate(invalid='ignore'):
    ref = data.astype(np.complex128) @ dense
    ref2 = data.astype(np.float64) @ dense.real + 1j * (data.astype(np.float64) @ dense.imag)
for name, r in (('zgemm', ref), ('split', ref2)):
    for f in range(48):
        for part in (np.real, np.imag):
            a, b = part(res[f]), part(r[f])
            bad = (np.isnan(a) != np.isnan(b)) | (np.isposinf(a) != np.isposinf(b)) | (np.isneginf(a) != np.isneginf(b))
            if bad.any():
                print(name, 'frame', f, part.__name__, 'cols', np.flatnonzero(bad)[:10], 'res', a[bad][:5], 'ref', b[bad][:5])
P = int(np.flatnonzero(~np.isfinite(data[3]))[0])
print('P', P, divmod(P, 128), data[3, P], 'n nonfinite', (~np.isfinite(data[3])).sum())
cols = csr[P].indices; print('cols storing P', cols, csr[P].data[:4])
print('res[3]', res[3][:14]); print('ref[3]', ref[3][:14])
print('res[3] bin1', res[3][13:27]); print('ref[3] bin1', ref[3][13:27])
print(h.last_kernel())
