import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
g = torch.Generator(device='cuda').manual_seed(3)
n = int(os.environ.get('N', 1024))
tile = torch.randint(0, 4096, (n, 65536), generator=g, device='cuda', dtype=torch.int16)
out_t = torch.zeros((n, 16), device='cuda', dtype=torch.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
for _ in range(200):
    h.apply(tile.data_ptr(), np.uint16, n, 65536, out_t.data_ptr(), 16, False)
torch.cuda.synchronize()
print(h.last_kernel())
