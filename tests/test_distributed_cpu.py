"""
The N>1 path on CPU: world_size-2 (and 3) `gloo` jobs exercising the nav sharding of
HipJobExecutor and its three merge modes (declared 'disjoint', declared 'sum', generic).
Every rank must end up with the complete, identical result == the single-process result.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_sharded_run(tmp_path, world):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env['OMP_NUM_THREADS'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py'), str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    outs = [np.load(os.path.join(tmp_path, f'rank{k}.npz')) for k in range(world)]
    # expected values, single process
    rng = np.random.default_rng(5)
    data = rng.integers(0, 100, (6, 7, 16, 16)).astype(np.uint16)
    masks = rng.random((3, 16, 16)).astype(np.float32)
    flat = data.reshape((42, -1)).astype(np.float32)
    exp_masks = (flat @ masks.reshape((3, -1)).T).reshape((6, 7, 3))
    exp_sum = data.astype(np.float32).sum(axis=(0, 1))
    exp_mx = data.max(axis=(0, 1)).astype(np.float32)
    exp_pf = data.max(axis=(2, 3)).astype(np.float32)
    roi = np.zeros((6, 7), dtype=bool)
    roi[1:5, 2:6] = True
    all_parts = []
    for o in outs:
        assert np.allclose(o['masks'], exp_masks, rtol=1e-6)
        assert np.array_equal(o['sum'], exp_sum)          # integer valued: exact in any order
        assert np.array_equal(o['mx'], exp_mx)
        assert np.array_equal(o['per_frame'], exp_pf)
        assert np.allclose(o['masks_roi'][roi], exp_masks[roi], rtol=1e-6)
        assert np.all(np.isnan(o['masks_roi'][~roi]))
        all_parts.extend(o['my_parts'].tolist())
        # sharded dataset: gathered nav result / reduced sig result == single-process values
        full = o['sh_full']
        exp = (full.reshape((-1, 256)).astype(np.float32) @ masks.reshape((3, -1)).T)
        assert np.allclose(o['sh_masks'], exp.reshape(full.shape[:2] + (3,)), rtol=1e-6)
        assert np.array_equal(o['sh_sum'], full.astype(np.float32).sum(axis=(0, 1)))
        # partition boundaries of sharded data never straddle two ranks; ROI that empties a shard
        full2 = o['sh2_full']
        exp2 = full2.reshape((-1, 256)).astype(np.float32) @ masks.reshape((3, -1)).T
        assert np.allclose(o['sh2_masks'], exp2, rtol=1e-6)
        assert np.allclose(o['sh2_roi_raw'], exp2[61 + 5:61 + 40], rtol=1e-6)
        assert np.array_equal(o['sh2_roi_sum'], full2[61 + 5:61 + 40].astype(np.float32).sum(axis=0))
        # live feed per rank + run_udf_iter: 3 steps (3 partitions per rank); after step k the
        # frames of the first k + 1 partitions of EVERY rank are merged and marked in the damage map
        live = o['live']
        flat_live = live.reshape((world, 12, 256)).astype(np.float32)      # (rank, local frame, px)
        bounds = np.linspace(0, 12, 4, dtype=int)
        assert o['live_masks'].shape[0] == 3
        for k in range(3):
            done = np.zeros((world, 12), dtype=bool)
            done[:, :bounds[k + 1]] = True
            dmg = o['live_damage'][k].reshape((world, 12))
            assert np.array_equal(dmg, done), k
            part_masks = o['live_masks'][k].reshape((world, 12, 3))
            exp_m = flat_live @ masks.reshape((3, -1)).T
            assert np.allclose(part_masks[done], exp_m[done], rtol=1e-6)
            assert np.all(part_masks[~done] == 0)
            assert np.array_equal(o['live_sum'][k],
                                  flat_live[done].sum(axis=0).reshape((16, 16)))
            assert np.array_equal(o['live_mx'][k], flat_live[done].max(axis=0).reshape((16, 16)))
    # identical on every rank
    for o in outs[1:]:
        for k in ('live_masks', 'live_sum', 'live_mx', 'live_damage'):
            assert np.array_equal(o[k], outs[0][k]), k
        for k in ('masks', 'sum', 'mx', 'per_frame'):
            assert np.array_equal(o[k], outs[0][k])
    # every partition processed exactly once, in contiguous blocks
    assert sorted(all_parts) == list(range(5))
