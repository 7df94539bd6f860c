"""SumUDF / SumSigUDF (rows a5 / a6) on C2-sized data, device resident: kernel-level and whole job."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
from libertem_amd.api import Context
from libertem_amd.udf.sum import SumUDF
from libertem_amd.udf.sumsigudf import SumSigUDF

frames, n_px = 65536, 65536
for name, tdt, ndt in (('uint16', torch.int16, np.uint16), ('float32', torch.float32, np.float32)):
    if tdt == torch.int16:
        tile = torch.randint(0, 4096, (frames, n_px), device='cuda', dtype=torch.int16)
    else:
        tile = torch.rand((frames // 2, n_px), device='cuda', dtype=torch.float32)
    n = tile.shape[0]
    nbytes = n * n_px * tile.element_size()
    out_sig = torch.zeros(n_px, device='cuda', dtype=torch.float32)
    out_nav = torch.zeros(n, device='cuda', dtype=torch.float32)
    ws_bytes = hip.sum_frames_workspace(n, n_px, np.float32)
    ws = torch.empty(max(ws_bytes, 4), device='cuda', dtype=torch.uint8)

    def t(fn, reps=10):
        fn(); torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]
    ms = t(lambda: hip.sum_frames(0, tile.data_ptr(), ndt, n, n_px, n_px, out_sig.data_ptr(), np.float32,
                                  False, ws.data_ptr()))
    print(f"ltmi_sum_frames {name}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 8000:.2f} of HBM peak)")
    ms = t(lambda: hip.sum_sig(0, tile.data_ptr(), ndt, n, n_px, n_px, out_nav.data_ptr(), np.float32, False))
    print(f"ltmi_sum_sig    {name}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 8000:.2f} of HBM peak)")
    del tile
ctx = Context.make_with('hip', gpus=0)
fr = torch.randint(0, 4096, (256, 256, 256, 256), device='cuda', dtype=torch.int16)
ds = ctx.load('memory', data=fr, dtype=np.uint16, sig_dims=2, num_partitions=1)
for udf in (SumUDF(), SumSigUDF()):
    for _ in range(2):
        ctx.run_udf(dataset=ds, udf=udf)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); ctx.run_udf(dataset=ds, udf=udf); ts.append(time.perf_counter() - t0)
    print(f"{type(udf).__name__} whole job: {np.median(ts) * 1e3:.2f} ms  {65536 / np.median(ts) / 1e6:.1f} Mframes/s")
