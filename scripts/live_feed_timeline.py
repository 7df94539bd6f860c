import sys, os, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
ctx = Context.make_with('hip', gpus=0)
n_frames, chunk = 16384, 1024
rng = np.random.default_rng(7)
frames = rng.integers(0, 4096, (n_frames, 256, 256), dtype=np.uint16)
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16, mask_dtype=np.float32)
import cProfile, pstats
for rep in range(4):
    ds = ctx.load('stream', frames=None, nav_shape=(n_frames // 256, 256), sig_shape=(256, 256),
                  dtype=np.uint16, num_partitions=n_frames // chunk)
    ds.scan_buffer[...] = frames
    def produce(ds=ds):
        for i in range(chunk, n_frames + 1, chunk):
            ds.commit(i)
    pr = cProfile.Profile() if rep == 3 else None
    t0 = time.perf_counter()
    th = threading.Thread(target=produce); th.start()
    ts = []
    if pr: pr.enable()
    for part in ctx.run_udf_iter(dataset=ds, udf=udf):
        ts.append(time.perf_counter() - t0)
    if pr: pr.disable()
    t1 = time.perf_counter() - t0
    th.join()
    print('rep', rep, 'total %.2f ms; first result %.2f ms; gaps %s' % (t1 * 1e3, ts[0] * 1e3, ' '.join('%.2f' % ((b - a) * 1e3) for a, b in zip(ts, ts[1:]))), flush=True)
    if pr:
        pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
