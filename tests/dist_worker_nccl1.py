"""
Worker for tests/test_udf_gpu.py::test_nccl_backend_collectives_through_ltmi_comm: ONE rank on the "nccl"
backend (= RCCL) with LTMI_FORCE_COLLECTIVES=1, so that the multi-GPU result path -- RCCL init, the
library's own communicator (ltmi_comm_*), all_gather of nav rows, all_reduce(sum) of sig buffers, the
lock-step partial results -- is executed on a one-GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from libertem_amd.api import Context
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF

    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    out_dir = sys.argv[1]
    ctx = Context.make_with('hip', gpus=0)
    ex = ctx.executor
    assert ex._collectives_on and ex.world_size == 1
    rng = np.random.default_rng(19)
    data = rng.integers(0, 4000, (6, 8, 32, 32)).astype(np.uint16)
    masks = rng.random((5, 32, 32)).astype(np.float32)
    dev = torch.from_numpy(data.view(np.int16)).to('cuda:0')
    ds = ctx.load('memory', data=dev, dtype=np.uint16, sig_dims=2, num_partitions=3)
    out = {'data': data, 'masks': masks}
    for via in ('rccl', 'auto'):
        os.environ['LTMI_RESULT_VIA'] = via
        res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks), SumUDF(),
                                           SumSigUDF()])
        out[f'{via}_masks'] = np.array(res[0]['intensity'].data)
        out[f'{via}_sum'] = np.array(res[1]['intensity'].data)
        out[f'{via}_sumsig'] = np.array(res[2]['intensity'].data)
        out[f'{via}_via'] = np.array(ex.last_result_via)
        out[f'{via}_collective'] = np.array(getattr(ex, 'last_collective', 'none'))
    os.environ['LTMI_RESULT_VIA'] = 'rccl'
    steps = [np.array(p.buffers[0]['intensity'].data) for p in ctx.run_udf_iter(
        dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))]
    out['iter_last'] = steps[-1]
    out['iter_steps'] = np.array(len(steps))
    # complex frames: complex sums through the communicator (2 real words per element)
    cdata = (rng.random((4, 4, 16, 16)) + 1j * rng.random((4, 4, 16, 16))).astype(np.complex64)
    ds_c = ctx.load('memory', data=cdata, sig_dims=2, num_partitions=2)
    out['c_sum'] = np.array(ctx.run_udf(dataset=ds_c, udf=SumUDF())['intensity'].data)
    out['c_data'] = cdata
    out['c_collective'] = np.array(getattr(ex, 'last_collective', 'none'))
    np.savez(os.path.join(out_dir, 'nccl1.npz'), **out)
    ctx.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
