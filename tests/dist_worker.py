"""
Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo job that drives the
HipJobExecutor's sharding + cross-rank merge with NumPy UDFs (no GPU).
"""
import os
import sys
import json

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from libertem_amd.api import Context
    from libertem_amd.executor.hip import HipJobExecutor
    from libertem_amd.udf.base import UDF

    class MasksDeclared(UDF):          # nav-kind, declares 'disjoint'
        def __init__(self, masks):
            super().__init__(masks=masks)

        def get_result_buffers(self):
            return {'intensity': self.buffer(kind='nav', extra_shape=(len(self.params.masks),),
                                             dtype=np.float32)}

        def process_tile(self, tile):
            m = self.meta.sig_slice.get(self.params.masks, sig_only=True)
            m = m.reshape((len(self.params.masks), -1)).T
            self.results.intensity[:] += tile.reshape((tile.shape[0], -1)) @ m

        def get_dist_merge(self):
            return {'intensity': 'disjoint'}

    class SumDeclared(UDF):            # sig-kind, declares 'sum'
        def get_result_buffers(self):
            return {'intensity': self.buffer(kind='sig', dtype=np.float32)}

        def process_tile(self, tile):
            self.results.intensity[:] += np.sum(tile, axis=0)

        def merge(self, dest, src):
            dest.intensity[:] += src.intensity

        def get_dist_merge(self):
            return {'intensity': 'sum'}

    class MaxGeneric(UDF):             # custom merge, no declaration -> generic object gather
        def get_result_buffers(self):
            return {'mx': self.buffer(kind='sig', dtype=np.float32),
                    'per_frame': self.buffer(kind='nav', dtype=np.float32)}

        def process_tile(self, tile):
            self.results.mx[:] = np.maximum(self.results.mx, tile.max(axis=0))
            self.results.per_frame[:] = np.maximum(
                self.results.per_frame, tile.reshape((tile.shape[0], -1)).max(axis=1))

        def merge(self, dest, src):
            dest.mx[:] = np.maximum(dest.mx, src.mx)
            dest.per_frame[:] = src.per_frame

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    out_dir = sys.argv[1]
    rng = np.random.default_rng(5)
    data = rng.integers(0, 100, (6, 7, 16, 16)).astype(np.uint16)     # same on every rank
    masks = rng.random((3, 16, 16)).astype(np.float32)
    ex = HipJobExecutor(require_gpu=False)
    assert ex.world_size == world and ex.rank == rank
    ctx = Context(executor=ex)
    ds = ctx.load('memory', data=data, num_partitions=5, sig_dims=2)
    roi = np.zeros((6, 7), dtype=bool)
    roi[1:5, 2:6] = True
    r1, r2, r3 = ctx.run_udf(dataset=ds, udf=[MasksDeclared(masks), SumDeclared(), MaxGeneric()])
    my_parts = np.array([t.idx for t in ex.my_tasks(ex._all_tasks)])
    r1_roi = ctx.run_udf(dataset=ds, udf=MasksDeclared(masks), roi=roi)
    # sharded dataset: every rank holds only its block of the first nav axis (bench.py's layout)
    full = rng.integers(0, 100, (world * 3, 5, 16, 16)).astype(np.uint16)
    local = full[rank * 3:(rank + 1) * 3]
    ds_sh = ctx.load('memory', data=local, shard=(rank, world), num_partitions=2, sig_dims=2)
    assert tuple(ds_sh.shape) == (world * 3, 5, 16, 16)
    rs1, rs2 = ctx.run_udf(dataset=ds_sh, udf=[MasksDeclared(masks), SumDeclared()])
    # sizes where a global np.linspace truncates a boundary to k*n_local - 1 (61 frames per rank,
    # 7 partitions per rank), and an ROI that removes every partition of rank 0's shard
    full2 = rng.integers(0, 100, (world * 61, 16, 16)).astype(np.uint16)
    ds_sh2 = ctx.load('memory', data=full2[rank * 61:(rank + 1) * 61], shard=(rank, world),
                      num_partitions=7, sig_dims=2)
    rs3 = ctx.run_udf(dataset=ds_sh2, udf=MasksDeclared(masks))
    roi2 = np.zeros((world * 61,), dtype=bool)
    roi2[61 + 5:61 + 40] = True
    rs4 = ctx.run_udf(dataset=ds_sh2, udf=[MasksDeclared(masks), SumDeclared()], roi=roi2)
    # live feed, one feeder per rank (StreamDataSet(shard=...)) + run_udf_iter across the ranks:
    # the ranks advance in lockstep, every rank yields the same partial result after each step
    import time as _time
    live = rng.integers(0, 100, (world * 4, 3, 16, 16)).astype(np.uint16)
    mine = live[rank * 4:(rank + 1) * 4].reshape((-1, 16, 16))

    def feed():
        for i in range(0, len(mine), 3):
            _time.sleep(0.002 * (rank + 1))          # the ranks' feeds are not in step
            yield mine[i:i + 3]

    ds_live = ctx.load('stream', frames=feed(), nav_shape=(world * 4, 3), sig_shape=(16, 16),
                       dtype=np.uint16, num_partitions=3, shard=(rank, world))
    steps_masks, steps_sum, steps_mx, steps_damage = [], [], [], []
    for part in ctx.run_udf_iter(dataset=ds_live, udf=[MasksDeclared(masks), SumDeclared(),
                                                        MaxGeneric()]):
        steps_masks.append(np.array(part.buffers[0]['intensity'].data))
        steps_sum.append(np.array(part.buffers[1]['intensity'].data))
        steps_mx.append(np.array(part.buffers[2]['mx'].data))
        steps_damage.append(np.array(part.damage.data))
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'),
             live=live, live_masks=np.stack(steps_masks), live_sum=np.stack(steps_sum),
             live_mx=np.stack(steps_mx), live_damage=np.stack(steps_damage),
             sh2_masks=rs3['intensity'].data, sh2_roi_raw=rs4[0]['intensity'].raw_data,
             sh2_roi_sum=rs4[1]['intensity'].data, sh2_full=full2,
             sh_masks=rs1['intensity'].data, sh_sum=rs2['intensity'].data, sh_full=full,
             masks=r1['intensity'].data, sum=r2['intensity'].data, mx=r3['mx'].data,
             per_frame=r3['per_frame'].data, masks_roi=r1_roi['intensity'].data,
             my_parts=my_parts)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({'rank': rank, 'ok': True}))


if __name__ == '__main__':
    main()
