"""ltmi_mib_decode on one MI355X: kernel time (HIP events) and HBM rate (bytes read + bytes written)
per format, frames of a 256x256 detector (quad: 512x512), file bytes already in HBM.  Second part: a
whole `ctx.load('mib', ...)` of files in the page cache (host copy + H2D + decode, the load-once cost)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
from libertem_amd import hip                                   # noqa: E402
from libertem_amd.common.hiparray import HipArray              # noqa: E402

PEAK = 8.0e12
CASES = [  # kind, bits, quad, (H, W), header, storage
    ('u', 8, False, (256, 256), 384, np.uint8), ('u', 16, False, (256, 256), 384, np.uint16),
    ('u', 32, False, (256, 256), 384, np.uint32),
    ('r', 1, False, (256, 256), 384, np.uint8), ('r', 6, False, (256, 256), 384, np.uint8),
    ('r', 12, False, (256, 256), 384, np.uint16), ('r', 24, False, (256, 256), 384, np.float32),
    ('r', 1, True, (512, 512), 768, np.uint8), ('r', 6, True, (512, 512), 768, np.uint8),
    ('r', 12, True, (512, 512), 768, np.uint16),
]


def main():
    n_target_bytes = 4 << 30
    for kind, bits, quad, (h, w), header, storage in CASES:
        payload = h * w * (bits if kind == 'u' else {1: 1, 6: 8, 12: 16, 24: 32}[bits]) // 8
        stride = header + payload
        out_frame = h * w * np.dtype(storage).itemsize
        n = int(n_target_bytes // (stride + out_frame))
        raw = torch.randint(0, 256, (n * stride,), dtype=torch.uint8, device='cuda:0')
        out = HipArray.empty((n, h, w), storage, 0)
        s = torch.cuda.current_stream(0)
        for _ in range(3):
            hip.mib_decode(0, raw.data_ptr(), stride, header, kind, bits, quad, n, h, w, out.data_ptr(),
                           storage, stream=s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(s)
        for _ in range(reps):
            hip.mib_decode(0, raw.data_ptr(), stride, header, kind, bits, quad, n, h, w, out.data_ptr(),
                           storage, stream=s.cuda_stream)
        e1.record(s)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        traffic = n * (payload + out_frame)
        print(f"{kind}{bits:<2d} {'quad' if quad else '    '} {h}x{w} -> {np.dtype(storage).name:8s} "
              f"{n:7d} frames {ms:7.3f} ms  {traffic / ms / 1e9:7.2f} TB/s = {traffic / ms / 1e-3 / PEAK:.2f} "
              f"of HBM   ({n / ms / 1e3:.1f} M frames/s)")
        del raw, out
    # whole load of a series from the page cache
    import recipes
    from libertem_amd.api import Context
    ctx = Context.make_with('hip', gpus=0)
    case = dict(name='big', kind='r', bits=12, sig=(256, 256), frames=(4096,) * 4, nav=(128, 128), seed=1)
    with tempfile.TemporaryDirectory() as d:
        rng = np.random.default_rng(0)
        frame = rng.integers(0, 4096, size=case['sig']).astype(np.uint16)
        one = recipes.mib_header(case, 1) + recipes.mib_frame_payload(frame, case)
        seq = 1
        for i, cnt in enumerate(case['frames']):
            with open(os.path.join(d, f"big{i + 1:06d}.mib"), 'wb') as f:
                f.write(recipes.mib_header(case, seq) + one[384:])
                f.write(one * (cnt - 1))
            seq += cnt
        with open(os.path.join(d, 'big.hdr'), 'w') as f:
            f.write("HDR,\t\nFrames in Acquisition (Number):\t16384\nFrames per Trigger (Number):\t128\nEnd\t")
        for rep in range(3):
            t0 = time.perf_counter()
            ds = ctx.load('mib', path=os.path.join(d, 'big.hdr'))
            dt = time.perf_counter() - t0
            print(f"ctx.load('mib') 16384 frames r12 256x256, 4 files, {ds.decode_bytes / 2**30:.2f} GiB: "
                  f"{dt * 1e3:.0f} ms = {ds.decode_bytes / dt / 1e9:.1f} GB/s "
                  f"(copy + decode part {ds.decode_seconds * 1e3:.0f} ms)")
            del ds
    ctx.close()


if __name__ == '__main__':
    main()
