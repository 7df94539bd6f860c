import sys, os, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
import libertem_amd.io.dataset.memory as mem
ctx = Context.make_with('hip', gpus=0)
n_frames, chunk = 16384, 1024
rng = np.random.default_rng(7)
frames = rng.integers(0, 4096, (n_frames, 256, 256), dtype=np.uint16)
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
T0 = [0]
log = []
oi = mem._HipStager.__init__
def init(self, *a, **k):
    t = time.perf_counter(); oi(self, *a, **k); log.append(('stager init', t - T0[0], time.perf_counter() - t))
mem._HipStager.__init__ = init
ou = mem._HipStager.upload
def up(self, *a, **k):
    t = time.perf_counter(); r = ou(self, *a, **k); log.append(('upload call', t - T0[0], time.perf_counter() - t)); return r
mem._HipStager.upload = up
for rep in range(4):
    ds = ctx.load('stream', frames=None, nav_shape=(n_frames // 256, 256), sig_shape=(256, 256),
                  dtype=np.uint16, num_partitions=n_frames // chunk)
    ds.scan_buffer[...] = frames
    def produce(ds=ds):
        for i in range(chunk, n_frames + 1, chunk):
            ds.commit(i)
    log.clear()
    T0[0] = t0 = time.perf_counter()
    th = threading.Thread(target=produce); th.start()
    ts = []
    for part in ctx.run_udf_iter(dataset=ds, udf=udf):
        ts.append(time.perf_counter() - t0)
    th.join()
    print('rep', rep, 'first result %.2f ms' % (ts[0] * 1e3), [(n, round(a * 1e3, 2), round(d * 1e3, 2)) for n, a, d in log[:4]], flush=True)
