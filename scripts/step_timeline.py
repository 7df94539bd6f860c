"""Host timeline of one full-size C2 run_udf step (median start/end offsets of the main phases)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd.api import Context
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd.udf import base as ubase
from libertem_amd import hip
from libertem_amd.executor import hip as hexec
from libertem_amd.io.dataset import base as dsbase

small = '--small' in sys.argv
ctx = Context.make_with('hip', gpus=0)
n = 8 if small else 256
frames = (torch.rand((n, n, 256, 256), device='cuda') * 4096).to(torch.int16)
masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=1)
udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                    mask_dtype=np.float32)
log = []


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        log.append((label, t, time.perf_counter()))
        return r
    setattr(obj, name, w)


wrap(ubase.UDFRunner, '_prepare_run_for_dataset')
wrap(ubase.UDFPartRunner, '_init_udfs')
wrap(ubase.UDFPartRunner, '_run_udfs')
wrap(ubase.UDFPartRunner, '_wrapup_udfs')
wrap(ubase.UDFPartRunner, 'run_for_partition')
wrap(dsbase.Negotiator, 'get_scheme')
wrap(hip.MaskHandle, 'apply', 'kernel launch call')
wrap(hexec.HipJobExecutor, '_to_host')
wrap(hexec.HipJobExecutor, 'merge_results')
wrap(ubase.UDF, '_do_get_results')
wrap(ubase.UDF, 'allocate_for_part')
wrap(ubase.UDF, 'allocate_for_full')
wrap(ubase.UDF, 'init_result_buffers')
wrap(ubase.UDF, 'init_task_data')
from libertem_amd.io.dataset import memory as dsmem
from libertem_amd.udf import masks as umasks
wrap(umasks.ApplyMasksUDF, 'process_tile', 'udf.process_tile')
wrap(umasks.ApplyMasksUDF, 'get_task_data')
wrap(umasks.ApplyMasksUDF, 'get_result_buffers')
wrap(ubase.UDFPartRunner, '_run_tile')
wrap(hexec.HipJobExecutor, 'run_tasks', 'run_tasks (generator creation)')
wrap(hexec.HipJobExecutor, '_merge_on_device')
wrap(ubase.UDFRunner, '_make_udf_tasks') if hasattr(ubase.UDFRunner, '_make_udf_tasks') else None
wrap(dsbase.DataSet, 'get_partitions') if hasattr(dsbase.DataSet, 'get_partitions') else None
for _ in range(5):
    ctx.run_udf(dataset=ds, udf=udf)
runs = []
for _ in range(40):
    torch.cuda.synchronize()
    log.clear()
    t0 = time.perf_counter()
    ctx.run_udf(dataset=ds, udf=udf)
    t1 = time.perf_counter()
    runs.append([(l, a - t0, b - t0) for l, a, b in log] + [('TOTAL', 0.0, t1 - t0)])
labels = [l for l, _, _ in runs[0]]
arr = np.array([[(a, b) for _, a, b in r] for r in runs])
med = np.median(arr, axis=0) * 1e6
order = np.argsort(med[:, 0])
for i in order:
    print(f"{labels[i]:28s} start {med[i, 0]:8.1f}  end {med[i, 1]:8.1f}  dur {med[i, 1] - med[i, 0]:8.1f} us")
