"""Board power and shader clock (hwmon; the card whose power moves is the one under test) while the C5 fold kernel, its
timing-only ablations and the unfolded kernel run back to back."""
import glob, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm
from libertem_amd.analysis.radialfourier import radial_mask_factory

cards = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
freq_files = [os.path.join(c, 'freq1_input') for c in cards]
power_files = [os.path.join(c, 'power1_input') if os.path.exists(os.path.join(c, 'power1_input'))
               else os.path.join(c, 'power1_average') for c in cards]


def read(f):
    try:
        return int(open(f).read().strip())
    except Exception:
        return -1


N = 1024
frames = int(os.environ.get('FRAMES', 8192))
ro = pm.bounding_radius(N / 2, N / 2, N, N)
flat = np.ascontiguousarray(radial_mask_factory(N, N, N / 2, N / 2, 0, ro, 1, 24, False)().reshape(25, -1))
h = hip.MaskHandle.dense(0, flat, np.complex64)
h.set_sig_shape(N, N)
tile = torch.rand((frames, N * N), device='cuda', dtype=torch.float32)
out = torch.zeros((frames, 25), device='cuda', dtype=torch.complex64)
idle = np.array([[read(f) for f in power_files] for _ in range(20)], dtype=float).mean(axis=0)
for name, code in (('idle', None), ('folded', 30), ('folded, no frame copies', 31), ('folded, no MFMA', 32),
                   ('unfolded', 38), ('folded again', 30)):
    samples, stop = [], False

    def poll():
        while not stop:
            samples.append([read(f) for f in freq_files] + [read(f) for f in power_files])
            time.sleep(0.01)
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.time(), 0
    if code is None:
        time.sleep(1.0)
    else:
        h.set_tuning(0, code, 0)
        while time.time() - t0 < 2.5:
            for _ in range(10):
                h.apply(tile.data_ptr(), np.float32, frames, N * N, out.data_ptr(), 25, False)
            torch.cuda.synchronize()
            n += 10
    el = time.time() - t0
    stop = True
    th.join()
    a = np.array(samples[len(samples) // 3:], dtype=float)
    nc = len(cards)
    pw = np.median(a[:, nc:], axis=0)
    card = int(np.argmax(pw - idle)) if code is not None else 0
    msg = f"{name:28s}"
    if n:
        ms = el / n * 1e3
        msg += f" {ms:7.3f} ms/launch  sclk {np.median(a[:, card]) / 1e6:5.0f} MHz  power {pw[card] / 1e6:5.0f} W  " \
               f"energy {pw[card] / 1e6 * ms / 1e3:6.2f} J/launch  ({cards[card].split('/')[4]})"
    print(msg, flush=True)
