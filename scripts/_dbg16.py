import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests')); sys.path.insert(0, os.path.join(os.getcwd(), 'tests', 'golden'))
import numpy as np, torch
import test_kernels_gpu as tk
from libertem_amd import hip
hip.lib()
sig=(128,128)
masks = tk._radial_stack(sig, 1, 24)
rng = np.random.default_rng(1)
d3 = rng.integers(0, 65535, (300, 128*128), endpoint=True).astype(np.uint16); d3[1] = 65535; d3[2] = 0
for name, data in (('const', d3), ('rand', rng.integers(0, 65535, (300, 128*128), endpoint=True).astype(np.uint16)),
                   ('small', rng.integers(0, 100, (300, 128*128), endpoint=True).astype(np.uint16))):
    res, kern = tk._fold_apply(hip, data, masks, sig, np.complex64, tuning=dict(mt=0, waves=30, ksplit=0))
    ref = tk._ref64(data, masks)
    scale = np.abs(data.astype(np.float64)) @ np.abs(masks).astype(np.float64).T
    err = np.abs(res - ref) / scale
    for part in (np.real, np.imag):
        e2 = np.abs(part(res) - part(ref)) / (scale + 1e-30)
        print(name, part.__name__, e2.max(), np.unravel_index(e2.argmax(), e2.shape), part(res)[np.unravel_index(e2.argmax(), e2.shape)], part(ref)[np.unravel_index(e2.argmax(), e2.shape)], scale[np.unravel_index(e2.argmax(), e2.shape)])
    print(name, kern, 'max err/scale', err.max(), 'at', np.unravel_index(err.argmax(), err.shape), 'frames bad', np.unique(np.argwhere(err > 1e-5)[:, 0])[:20], 'cols bad', np.unique(np.argwhere(err > 1e-5)[:, 1])[:30])
