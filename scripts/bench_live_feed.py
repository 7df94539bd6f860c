"""Live feed (row f4) with the feeder's chunk copy on 4 threads (default) and on 1: bench.py's live_feed()."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from libertem_amd.api import Context                            # noqa: E402
from libertem_amd.io.dataset.stream import StreamDataSet        # noqa: E402

ctx = Context.make_with('hip', gpus=0)
for threads in (StreamDataSet.COPY_THREADS, 1):
    StreamDataSet.COPY_THREADS = threads
    r = bench.live_feed(ctx)
    print(threads, 'copy thread(s):', json.dumps({k: r[k] for k in ('frames_per_s', 'GBps', 'ms_per_scan')}))
