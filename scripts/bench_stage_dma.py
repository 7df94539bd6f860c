"""H2D DMA rate out of page-locked bounce buffers while a staging copy (ltmi_host_copy) runs on the same host:
does the copy's DRAM traffic slow the DMA?  python scripts/bench_stage_dma.py"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
hip.lib()
N = 256 << 20
pinned = [torch.empty(N, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
dev = [torch.empty(N, dtype=torch.uint8, device='cuda') for _ in range(2)]
src = np.random.default_rng(0).integers(0, 255, 2 << 30, dtype=np.uint8)
st = torch.cuda.Stream()


def dma_only(reps=16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        for i in range(reps):
            dev[i & 1].copy_(pinned[i & 1], non_blocking=True)
    st.synchronize()
    return reps * N / (time.perf_counter() - t0) / 1e9


print(f"DMA alone: {dma_only():.1f} GB/s")
for th in (0, 2, 4, 8, 16):
    stop = [False]

    def churn():
        scratch = np.empty(N, dtype=np.uint8)
        while not stop[0]:
            hip.host_copy(scratch, src[:N], th or 1) if th else time.sleep(0.001)
    t = threading.Thread(target=churn)
    t.start()
    time.sleep(0.05)
    r = dma_only()
    stop[0] = True
    t.join()
    print(f"DMA with a {th}-thread pageable->pageable copy running: {r:.1f} GB/s")
# the real pipeline: stage chunk k+1 into pinned[s^1] while DMA of chunk k runs, events as in _HipStager
for th in (4, 8, 16):
    for piece in (32 << 20, 64 << 20, 256 << 20):
        done = [None, None]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(8):
            s = k & 1
            if done[s] is not None:
                done[s].synchronize()
            chunk = src[k * N:(k + 1) * N]
            with torch.cuda.stream(st):
                for a in range(0, N, piece):
                    hip.host_copy(pinned[s][a:a + piece].numpy(), chunk[a:a + piece], th)
                    dev[s][a:a + piece].copy_(pinned[s][a:a + piece], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
                done[s] = ev
        st.synchronize()
        print(f"staged pipeline, {th} threads, pieces of {piece >> 20} MiB: {8 * N / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
