"""Which k_dense_lds variant has LDS bank conflicts?  One launch each (run under
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE): float32 / uint16 frames x 16 / 48 / 50 columns."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
n, px = 4096, 512 * 512
rng = np.random.default_rng(0)
f32 = torch.rand((n, px), device='cuda')
u16 = torch.randint(0, 4096, (n, px), device='cuda', dtype=torch.int32).to(torch.int16)
for cols in (16, 32, 48, 50):
    m = rng.random((cols, px)).astype(np.float32)
    out = torch.zeros((n, cols), device='cuda')
    for name, t, dt, tun in (('f32', f32, np.float32, 0), ('u16-f32instr', u16, np.uint16, 37), ('u16-f16', u16, np.uint16, 0)):
        h = hip.MaskHandle.dense(0, m, np.float32)
        if tun:
            h.set_tuning(0, tun, 0)
        h.apply(t.data_ptr(), dt, n, px, out.data_ptr(), cols, False)
        torch.cuda.synchronize()
        print(cols, name, h.last_kernel())
        h.close()
