"""Board power and shader clock (hwmon) while the C4 sparse kernel and its timing-only ablations run back to back
(LTMI_BELL_ABLATE: 1 no frame copies, 3 neither copies nor barriers, 4 copies + record stream but no LDS reads /
VALU / MFMA, 5 copies only): is k_bell_flat bound by the board's power cap like the dense kernels
(profiles/r01_clock_power_probe.txt)?   energy per launch = power x time."""
import glob, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm

freq_files = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input'))
power_files = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average')) + \
    sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'))
print('freq', freq_files, 'power', power_files)


def read(f):
    try:
        return int(open(f).read().strip())
    except Exception:
        return -1


frames = int(os.environ.get('FRAMES', 16384))
rings = pm.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
g = torch.Generator(device='cuda').manual_seed(1)
tile = torch.randint(0, 4096, (frames, 65536), generator=g, device='cuda', dtype=torch.int32).to(torch.int16)
out = torch.zeros((frames, 1024), device='cuda', dtype=torch.float32)
dt = np.dtype('uint16')
runs = [('idle', None, None), ('shipped', '0', None), ('no frame copies', '1', None),
        ('no copies, no barriers', '3', None), ('copies + record stream, no arithmetic', '4', None),
        ('copies only', '5', None), ('shipped again', '0', None), ('k_scatter', '0', '1')]
for name, abl, scat in runs:
    if abl is not None:
        os.environ['LTMI_BELL_ABLATE'] = abl
    if scat:
        os.environ['LTMI_SPARSE_SCATTER'] = scat
    h = hip.MaskHandle.csr(0, csr, np.float32) if abl is not None else None
    samples = []
    stop = False

    def poll():
        while not stop:
            samples.append([read(f) for f in freq_files] + [read(f) for f in power_files])
            time.sleep(0.01)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    if h is None:
        time.sleep(1.0)
    else:
        while time.time() - t0 < 2.5:
            for _ in range(100):
                h.apply(tile.data_ptr(), dt, frames, 65536, out.data_ptr(), 1024, False)
            torch.cuda.synchronize()
            n += 100
    el = time.time() - t0
    stop = True
    th.join()
    a = np.array(samples[len(samples) // 3:], dtype=float)
    f_mhz = np.median(a[:, 0]) / 1e6
    p_w = np.median(a[:, len(freq_files)]) / 1e6
    msg = f"{name:42s}"
    if n:
        ms = el / n * 1e3
        msg += f" {ms:.3f} ms/launch  {h.last_kernel().split(' ')[0]:28s}"
        msg += f" sclk {f_mhz:.0f} MHz  power {p_w:.0f} W  energy {p_w * ms / 1e3:.3f} J/launch"
        h.close()
    else:
        msg += f" sclk {f_mhz:.0f} MHz  power {p_w:.0f} W"
    print(msg, flush=True)
