"""
Worker for tests/test_udf_gpu.py::test_two_ranks_*: one rank of a 2-rank job on a ONE-GPU box.
Both ranks drive GPU 0 (LIBERTEM_USE_HIP=0) and rendezvous over gloo -- RCCL refuses two ranks on one
device -- which is enough for the result delivery through the node-shared host segment
(executor/nodeshared.py: no data-path collective); 'sum' buffers go through gloo's all_reduce.
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

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    out_dir = sys.argv[1]
    ctx = Context.make_with('hip', gpus=0)
    ex = ctx.executor
    assert ex.world_size == world and ex.rank == rank and ex._node_shared() is not None
    rng = np.random.default_rng(9)
    masks = rng.random((5, 32, 32)).astype(np.float32)
    out = {}
    # (1) sharded, device-resident (bench.py's layout): every rank holds its block of the scan
    full = rng.integers(0, 4000, (world * 6, 8, 32, 32)).astype(np.uint16)
    mine = full[rank * 6:(rank + 1) * 6]
    dev = torch.from_numpy(mine.view(np.int16)).to('cuda:0')
    ds = ctx.load('memory', data=dev, dtype=np.uint16, sig_dims=2, num_partitions=2,
                  shard=(rank, world))
    for rep in range(7):                  # more runs than ring slots: slots are recycled
        res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks), SumUDF(),
                                           SumSigUDF()])
        if rep == 0:
            first = res[0]['intensity']           # a BufferWrapper kept alive across the recycling
            first_copy = np.array(first.data)
        if rep == 1 and rank == 0:
            # a BARE ndarray view, held on ONE rank only: the slot must not be rewritten by the
            # other rank either (results are caller-owned for as long as they are referenced)
            bare = res[2]['intensity'].data[1:3]
            bare_copy = np.array(bare)
    out['sh_masks'] = res[0]['intensity'].data
    out['sh_sum'] = res[1]['intensity'].data
    out['sh_sumsig'] = res[2]['intensity'].data
    out['sh_first_still_valid'] = np.array(np.array_equal(first.data, first_copy))
    out['sh_bare_still_valid'] = np.array(rank != 0 or np.array_equal(bare, bare_copy))
    out['sh_slots'] = np.array(len(ex._node_shared().slots))
    # every slot held (ring limit 6 in this test): the run falls back to the collectives and the
    # held results stay untouched
    held = []
    for _ in range(8):
        held.append(ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data)
        if os.environ.get('LTMI_TEST_DEBUG'):
            sh = ex._node_shared()
            print(rank, 'held', len(held), 'via', ex.last_result_via, 'slots', len(sh.slots),
                  'common_free', bin(sh.common_free), 'local_free', bin(sh._local_free()),
                  'base', type(held[-1].base), flush=True)
    out['sh_held_equal'] = np.array(all(np.array_equal(h, held[0]) for h in held))
    out['sh_via_when_full'] = np.array(ex.last_result_via)
    del held
    out['sh_full'] = full
    # (2) replicated host dataset, 5 partitions over 2 ranks, with and without ROI
    data = rng.integers(0, 4000, (5, 9, 32, 32)).astype(np.uint16)
    ds2 = ctx.load('memory', data=data, sig_dims=2, num_partitions=5)
    out['rep_masks'] = ctx.run_udf(dataset=ds2, udf=ApplyMasksUDF(
        mask_factories=lambda: masks))['intensity'].data
    roi = rng.random((5, 9)) < 0.5
    part = ctx.run_udf(dataset=ds2, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    out['rep_roi_raw'] = part['intensity'].raw_data
    out['rep_data'] = data
    out['rep_roi'] = roi
    # (3) the device collectives on request (gloo moves the device tensors through the host)
    os.environ['LTMI_RESULT_VIA'] = 'rccl'
    out['coll_masks'] = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
        mask_factories=lambda: masks))['intensity'].data
    del os.environ['LTMI_RESULT_VIA']
    # (4) live feed, one feeder per rank, partial results across the ranks on the device path
    live = rng.integers(0, 4000, (world * 4, 5, 32, 32)).astype(np.uint16)
    mine_live = live[rank * 4:(rank + 1) * 4].reshape((-1, 32, 32))
    ds_live = ctx.load('stream', frames=(mine_live[i:i + 4] for i in range(0, 20, 4)),
                       nav_shape=(world * 4, 5), sig_shape=(32, 32), dtype=np.uint16,
                       num_partitions=2, shard=(rank, world))
    steps = []
    for part in ctx.run_udf_iter(dataset=ds_live, udf=ApplyMasksUDF(mask_factories=lambda: masks)):
        steps.append((np.array(part.buffers[0]['intensity'].data), np.array(part.damage.data)))
    out['live'] = live
    out['live_step_masks'] = np.stack([a for a, _ in steps])
    out['live_step_damage'] = np.stack([b for _, b in steps])
    # (5) a Merlin .mib series: every rank unpacks and holds its block of the scan (shard), ROI run
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import recipes
    case = [c for c in recipes.MIB_CASES if c['name'] == 'r12'][0]           # 6 frames of 32 x 64, 12 bit
    mib_frames, mib_files, mib_hdr = recipes.make_mib_case(case)
    mib_dir = os.path.join(out_dir, 'mib')
    if rank == 0:
        os.makedirs(mib_dir, exist_ok=True)
        for fn, blob in mib_files.items():
            with open(os.path.join(mib_dir, fn), 'wb') as f:
                f.write(blob)
        with open(os.path.join(mib_dir, 'r12.hdr'), 'w') as f:
            f.write(mib_hdr)
    dist.barrier()
    ds_mib = ctx.load('mib', path=os.path.join(mib_dir, 'r12.hdr'), nav_shape=(world, 6 // world),
                      shard=(rank, world))
    mib_masks = rng.random((4, 32, 64)).astype(np.float32)
    mib_roi = np.array([True, False, True, True, True, False]).reshape((world, 6 // world))
    out['mib_full'] = ctx.run_udf(dataset=ds_mib, udf=ApplyMasksUDF(
        mask_factories=lambda: mib_masks))['intensity'].data
    out['mib_roi_raw'] = ctx.run_udf(dataset=ds_mib, udf=ApplyMasksUDF(
        mask_factories=lambda: mib_masks), roi=mib_roi)['intensity'].raw_data
    out['mib_masks'] = mib_masks
    out['mib_frames'] = mib_frames
    out['mib_roi'] = mib_roi
    out['mib_local_frames'] = np.array(ds_mib.decode_bytes // (384 + 32 * 64 * 2))
    # (6) a run whose shared delivery is called off: one UDF declares a 'disjoint' device buffer but
    #     keeps it out of the streamed delivery (postprocess(), no write-once rows) -> all_ok() is
    #     False on every rank and EVERY buffer goes through the collectives -- also the rows of
    #     ApplyMasksUDF / SumSigUDF that the kernels had already written into the host segment
    class LateSumSig(SumSigUDF):
        def get_write_once_buffers(self):
            return ()

        def postprocess(self):
            pass
    res6 = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks), LateSumSig(),
                                        SumSigUDF()])
    out['mix_via'] = np.array(ex.last_result_via)
    out['mix_masks'] = np.array(res6[0]['intensity'].data)
    out['mix_late'] = np.array(res6[1]['intensity'].data)
    out['mix_sumsig'] = np.array(res6[2]['intensity'].data)
    out['masks'] = masks
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **out)
    dist.barrier()
    ctx.close() if hasattr(ctx, 'close') else None
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
