"""bench.py's small_tiles() alone (C2 launches of 1 024 / 4 096 / 8 192 frames); LTMI_KSPLIT_FUSED=1 in the
environment selects the in-kernel reduction of the K split."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
import bench                                                    # noqa: E402
from libertem_amd import hip                                    # noqa: E402

r = bench.small_tiles(torch, hip)
for k, v in r.items():
    if isinstance(v, dict):
        print(k, v['kernel'], f"{v['avg_launch_ms'] * 1e3:.1f} us  {v['frac_of_hbm_peak']:.3f} of HBM")
