"""Sample GPU clock / power from sysfs (hwmon) while the dense kernel variants run back to back:
separates DVFS (clock drop when MFMA and HBM streams are both active) from structural stalls."""
import glob, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

def find(pattern):
    return sorted(glob.glob(pattern))

freq_files = find('/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input')
power_files = find('/sys/class/drm/card*/device/hwmon/hwmon*/power*_average') + \
    find('/sys/class/drm/card*/device/hwmon/hwmon*/power*_input')
print('freq files', freq_files)
print('power files', power_files)
for f in find('/sys/class/drm/card*/device/pp_dpm_sclk')[:1]:
    print(f, open(f).read().replace('\n', ' | '))

def read(f):
    try:
        return int(open(f).read().strip())
    except Exception:
        return -1

frames, n_px = 65536, 65536
g = torch.Generator(device='cuda').manual_seed(1)
tile = torch.randint(0, 4096, (frames, n_px), generator=g, device='cuda', dtype=torch.int32).to(torch.int16)
masks = np.random.default_rng(2).random((16, n_px)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
out = torch.zeros((frames, 16), device='cuda')
dt = np.dtype('uint16')
for name, v in [('idle', None), ('default', dict(mt=0, waves=0, ksplit=0)),
                ('tiles=1 (8 waves x 16 frames)', dict(mt=0, waves=34, ksplit=0)),
                ('tiles=2 (4 waves x 32 frames)', dict(mt=0, waves=35, ksplit=0)),
                ('tiles=1 again', dict(mt=0, waves=34, ksplit=0)),
                ('no-MFMA (memory stream only)', dict(mt=0, waves=32, ksplit=0)),
                ('no-DMA (compute stream only)', dict(mt=0, waves=31, ksplit=0)),
                ('default again', dict(mt=0, waves=0, ksplit=0))]:
    samples = []
    stop = False
    def poll():
        while not stop:
            samples.append([read(f) for f in freq_files] + [read(f) for f in power_files])
            time.sleep(0.01)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    if v is None:
        time.sleep(1.0)
    else:
        h.set_tuning(**v)
        while time.time() - t0 < 2.5:
            for _ in range(50):
                h.apply(tile.data_ptr(), dt, frames, n_px, out.data_ptr(), 16, False)
            torch.cuda.synchronize(); n += 50
    el = time.time() - t0
    stop = True; th.join()
    a = np.array(samples[len(samples) // 3:], dtype=float)      # steady state
    msg = f"{name:32s}"
    if n:
        msg += f" {el / n * 1e3:.3f} ms/launch "
    msg += ' freq(MHz) ' + ' '.join(f"{x / 1e6:.0f}" for x in np.median(a[:, :len(freq_files)], axis=0))
    msg += ' power(W) ' + ' '.join(f"{x / 1e6:.0f}" for x in np.median(a[:, len(freq_files):], axis=0))
    print(msg, flush=True)
