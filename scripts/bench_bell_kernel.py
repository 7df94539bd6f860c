import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as M
from libertem_amd.common.container import MaskContainer
import scipy.sparse as sp
fr = torch.randint(0, 4096, (16384, 256*256), device='cuda', dtype=torch.int16)
st = M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256, n_bins=1024, use_sparse=True, dtype=np.float32)
csr = sp.csr_matrix((st.data, (st.px_idx, st.mask_idx)), shape=(65536, 1024))
h = hip.MaskHandle.csr(0, csr, np.float32)
out = torch.empty((16384, 1024), dtype=torch.float32, device='cuda')
def run(n=10):
    for _ in range(3): h.apply(fr.data_ptr(), np.dtype('uint16'), 16384, 65536, out.data_ptr(), 1024, False)
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): h.apply(fr.data_ptr(), np.dtype('uint16'), 16384, 65536, out.data_ptr(), 1024, False)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
print(os.environ.get('LTMI_BELL_V'), os.environ.get('LTMI_BELL_TILES'), os.environ.get('LTMI_BELL_ABLATE'), '%.3f ms' % run(), h.last_kernel())
