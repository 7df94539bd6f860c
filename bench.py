#!/usr/bin/env python
"""
bench.py -- headline benchmark: ApplyMasksUDF, 16 dense float32 masks, 256x256 scan x 256x256
detector uint16 (BASELINE.json configs[1], "C2"), frames resident in HBM, through Context.run_udf.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE complete `Context.run_udf(dataset, ApplyMasksUDF(...))` job over the per-GPU
dataset (planning + kernels + delivery of the complete nav result to the host of EVERY rank), i.e.
the whole hot path, not just the kernel.  Weak scaling: every rank holds its own 256x256-scan shard
(65536 frames, 8 GiB); the nav grid of the job is N x that.

Rank 0 prints ONE JSON line.  `value` / `roofline` / `cpu_baseline` are the C2 figures the contract
asks for; `roofline` also carries the strict float32-instruction leg of C2 as flat scalars (`f32_instr_*`).
The same line carries, as extra keys (all configs, both timing modes):

  N = 1:  "configs": {"c3": ..., "c4": ..., "c5": ...}   whole job + dominant kernel + roofline + cpu_baseline each
          (C5's roofline is priced against HBM; `mfma_algorithmic_frac` / `mfma_issued_frac` ride beside it)
          "delivery_anchor_n1": the C2 steps with the multi-rank delivery forced on a world of one (shm and rccl),
                                the like-for-like N = 1 points of the first real multi-GPU lines
          "host_streamed": {"c2" .. "c5"}: frames in host memory, double-buffered hipMemcpyAsync, vs
                           the H2D peak measured in the same run
          "crystallinity": row f3, k_cryst_fused and the hipFFT route of the same call;
          "small_tiles": C2 launches of 1 024 / 4 096 / 8 192 frames; "live_feed": run_udf_iter on a
                         StreamDataSet
  N > 1:  "result_via", "per_rank" (kernel / step time of every rank),
          "rccl_path": the same steps with the results gathered by RCCL (all_gather over xGMI)
                       instead of the node-shared host segment,
          "strong_c3": CoM analysis on a FIXED 512x512 scan x 512x512 uint16 (128 GiB) nav-split
                       over the N ranks (strong scaling).

`--config c3|c4|c5` makes that config the measured one (`value` then is ITS frames/s; used by
scripts/profile_round.sh for per-config rocprofv3 passes); `--no-extras` skips the extra keys;
`--result-via shm|rccl` picks the delivery of the nav results across ranks (default: the executor's choice).

`roofline.achieved` = algorithmic bytes (or flops) per launch / average launch duration from HIP
events on the kernel's own stream inside the timed region; `roofline.traffic` = HBM bytes per launch
from the tracked PMC profile (profiles/traffic.json, written by scripts/traffic_from_rocprof.py from
separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over exactly this command), matched on
config, kernel label and frames per launch -- null if the kernel changed since the profile was taken.
`cpu_baseline` is the oracle (the CPU restatement of the reference path) timed on this box's host
cores on a bounded sample; it is the checker and the baseline, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3         # dense f32 matrix peak (same guide)

CONFIGS = {
    'c2': dict(scan=(256, 256), det=(256, 256), dtype='uint16', n_masks=16, result_bytes=64,
               flops=2 * 65536 * 16, bound='hbm', kernel='k_dense', preheat=30,
               desc='ApplyMasksUDF 16 dense f32 masks, 256x256 scan x 256x256 uint16'),
    'c2-small': dict(scan=(64, 64), det=(256, 256), dtype='uint16', n_masks=16, result_bytes=64,
                     flops=2 * 65536 * 16, bound='hbm', kernel='k_dense',
                     desc='ApplyMasksUDF 16 dense f32 masks, 64x64 scan x 256x256 uint16'),
    'c3': dict(scan=(512, 512), det=(512, 512), dtype='uint16', n_masks=3, result_bytes=12,
               flops=2 * 262144 * 3, bound='hbm', kernel='k_dense', preheat=3,
               desc='COMAnalysis (3 masks + post-processing), 512x512 scan x 512x512 uint16'),
    'c4': dict(scan=(256, 256), det=(256, 256), dtype='uint16', n_masks=1024, result_bytes=4096,
               flops=2 * 432407, bound='hbm', kernel='k_bell|k_sell', preheat=8,
               desc='ApplyMasksUDF 1024 sparse ring masks (CSR, nnz 432407), 256x256 scan x '
                    '256x256 uint16'),
    'c5': dict(scan=(128, 128), det=(1024, 1024), dtype='float32', n_masks=25, result_bytes=200,
               flops=4 * 1048576 * 25, bound='hbm', kernel='k_dense', preheat=3,
               # the row-mirror fold multiplies 513 of the 1024 rows against 64 padded columns
               issued_flops=2 * 513 * 1024 * 64,
               desc='RadialFourierAnalysis defaults (25 dense complex64 masks), 128x128 scan x '
                    '1024x1024 float32'),
}


# --------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = the checker; forked BEFORE the HIP runtime is initialised)
# --------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """One CPU worker = one process with ONE BLAS thread (reference: executor/dask.py:251), running the
    oracle's restatement of the reference path for one config over its own frames, `passes` times."""
    name, n_frames, passes, seed = job
    cfg = CONFIGS[name]
    import torch
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import path as opath
    rng = np.random.default_rng(seed)
    det = tuple(cfg['det'])
    if cfg['dtype'] == 'float32':
        data = rng.random((1, n_frames) + det, dtype=np.float32)
    else:
        data = rng.integers(0, 4096, (1, n_frames) + det).astype(cfg['dtype'])
    if name == 'c2':
        masks = np.random.default_rng(2).random((cfg['n_masks'],) + det).astype(np.float32)
        run = lambda d: opath.apply_masks(d, masks, num_partitions=1)                     # noqa: E731
    elif name == 'c3':
        run = lambda d: opath.com_analysis(d, num_partitions=1, cx=256, cy=256)           # noqa: E731
    elif name == 'c4':
        import scipy.sparse as sp
        from oracle import masks as omasks
        csr = sp.csr_matrix(omasks.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32))
        # the reference multiplies with a numba-compiled CSR loop (common/numba/__init__.py:153-184); numba is not
        # installed here and the oracle's restatement of that loop is interpreted Python, so the TIMED product is
        # SciPy's compiled CSR kernel on the same tiles -- the nearest compiled equivalent, same arithmetic
        product = lambda tile, m: np.asarray((m.T @ tile.T).T)                            # noqa: E731
        run = lambda d: opath.apply_masks_sparse(d, csr, num_partitions=1, product=product)   # noqa: E731
    elif name == 'c5':
        # the stack once per worker, like the reference's MaskContainer keeps it (common/container.py:260-314);
        # timed: the product of radial_fourier_analysis (float32 whole-frame tiles @ complex64 stack)
        from oracle import masks as omasks
        p = opath.radial_fourier_parameters(det)
        stack = omasks.radial_mask_stack(det[0], det[1], p['cx'], p['cy'], p['ri'], p['ro'], p['n_bins'], p['max_order'])
        run = lambda d: opath.apply_masks(d, stack, num_partitions=1, mask_dtype=np.complex64)   # noqa: E731
    else:
        raise ValueError(name)
    run(data[:, :min(n_frames, 32 if name in ('c2', 'c4') else 2)])      # warm up (mask stacks, BLAS)
    t0 = time.time()
    for _ in range(passes):
        run(data)
    return n_frames * passes, t0, time.time()


#: frames per worker and pass, and what the sample line says, per config
_CPU_SAMPLES = {
    'c2': (512, "oracle.path.apply_masks: reference tile shape (32,32,256), astype(float32) + torch.mm per tile"),
    'c3': (64, "oracle.path.com_analysis: 3 masks, tiles (32,16,512), torch.mm per tile + the CoM post-processing"),
    'c4': (256, "oracle.path.apply_masks_sparse on (32,32,256) tiles with SciPy's compiled CSR product standing in "
                "for the reference's numba loop (not installable here)"),
    'c5': (4, "the product of oracle.path.radial_fourier_analysis: 25 dense complex64 masks of 1024x1024 built once "
              "per worker, whole-frame tiles, float32 tile @ complex64 stack"),
}


def cpu_baseline(name='c2', budget_s=12.0, pool=None, cores=None):
    """
    The reference's CPU path restated (oracle/path.py), one single-threaded worker process per
    physical core, on a bounded sample.  MUST run before the parent touches the GPU (fork).
    """
    import multiprocessing as mp
    if cores is None:
        cores = physical_cores()
    n_frames, what = _CPU_SAMPLES[name]
    own = pool is None
    if own:
        pool = mp.get_context('fork').Pool(cores)
    try:
        # calibrate with one pass, then size the run to ~budget_s
        # (worker start-up skew -- importing torch in 100+ processes -- must not count)
        res = pool.map(_cpu_worker, [(name, n_frames, 1, 100 + i) for i in range(cores)])
        t1 = float(np.median([r[2] - r[1] for r in res]))
        passes = int(max(1, min(400, budget_s / max(t1, 1e-3))))
        res = pool.map(_cpu_worker, [(name, n_frames, passes, 100 + i) for i in range(cores)])
    finally:
        if own:
            pool.close()
            pool.join()
    total = sum(r[0] for r in res)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    return {"value": total / wall, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{cores} single-threaded worker processes x {n_frames} frames x {passes} "
                      f"passes of the {name.upper()} workload ({what}), {wall:.1f} s wall"}


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def cpu_baselines(names, budget_s):
    """{config: cpu_baseline dict} with ONE pool of worker processes for all of them (forked before HIP)"""
    import multiprocessing as mp
    cores = physical_cores()
    out = {}
    with mp.get_context('fork').Pool(cores) as pool:
        for name in names:
            try:
                out[name] = cpu_baseline(name, budget_s[name], pool=pool, cores=cores)
            except Exception as e:                    # the baseline must never sink the bench line
                out[name] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                             "sample": f"failed: {e!r}"[:300]}
    return out


# --------------------------------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------------------------------
def device_frames(torch, n_frames, det, dtype, seed):
    """synthetic frames generated ON the device (not timed): uint16 counts in [0, 4096) /
    float32 in [0, 1)"""
    g = torch.Generator(device='cuda').manual_seed(seed)
    n_px = det[0] * det[1]
    if np.dtype(dtype) == np.uint16:
        t = torch.empty((n_frames, n_px), dtype=torch.int16, device='cuda')
        step = max(1, (1 << 30) // (n_px * 2))
        for i in range(0, n_frames, step):
            j = min(n_frames, i + step)
            t[i:j] = torch.randint(0, 4096, (j - i, n_px), generator=g, device='cuda',
                                   dtype=torch.int16)
        return t
    t = torch.empty((n_frames, n_px), dtype=torch.float32, device='cuda')
    step = max(1, (1 << 30) // (n_px * 4))
    for i in range(0, n_frames, step):
        j = min(n_frames, i + step)
        t[i:j] = torch.rand((j - i, n_px), generator=g, device='cuda')
    return t


class Workload:
    """One BASELINE.json config on this rank's shard: .step() runs the whole job once through the
    public API, .check(result) compares a few frames with float64 NumPy."""

    def __init__(self, name, ctx, torch, rank, world, sharded, frames_per_rank=None):
        from libertem_amd.udf.masks import ApplyMasksUDF
        from libertem_amd import masks as M
        cfg = CONFIGS[name]
        self.name, self.cfg, self.ctx, self.torch = name, cfg, ctx, torch
        scan, det = cfg['scan'], cfg['det']
        self.det = det
        self.n_px = det[0] * det[1]
        self.itemsize = np.dtype(cfg['dtype']).itemsize
        self.n_local = frames_per_rank if frames_per_rank is not None else scan[0] * scan[1]
        self.world, self.rank = world, rank
        self.use_oracle = False
        self.frames = device_frames(torch, self.n_local, det, cfg['dtype'], 1 + rank + 17 * len(name))
        rows = self.n_local // scan[1]
        assert rows * scan[1] == self.n_local
        data = self.frames.reshape((rows, scan[1]) + det)
        self.ds = ctx.load('memory', data=data, dtype=np.dtype(cfg['dtype']), sig_dims=2,
                           num_partitions=1, shard=(rank, world) if sharded else None)
        self.nav = (rows * (world if sharded else 1), scan[1])
        if name in ('c2', 'c2-small'):
            masks = self.masks = np.random.default_rng(2).random((16,) + det).astype(np.float32)
            # (the factory must not close over `self`: factories are pickled for their size check,
            # like in the reference, common/container.py)
            udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False,
                                mask_count=16, mask_dtype=np.float32)
            ds = self.ds
            self.step = lambda: ctx.run_udf(dataset=ds, udf=udf)
        elif name == 'c3':
            an = ctx.create_com_analysis(dataset=self.ds, cx=256, cy=256)
            self.step = lambda: ctx.run(an)
        elif name == 'c4':
            def rings():
                return M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256,
                                     n_bins=1024, use_sparse=True, dtype=np.float32)
            udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=1024,
                                mask_dtype=np.float32)
            ds = self.ds
            self.step = lambda: ctx.run_udf(dataset=ds, udf=udf)
            # the same job with the 256 MiB result kept in HBM (for a follow-up on the device)
            self.step_device = lambda: ctx.run_udf(dataset=ds, udf=udf, result_where='device')
        elif name == 'c5':
            self.analysis = ctx.create_radial_fourier_analysis(dataset=self.ds)
            assert self.analysis.parameters['use_sparse'] is False
            self.step = lambda: ctx.run(self.analysis)
        else:
            raise ValueError(name)

    def _frame(self, local_idx):
        f = self.frames[local_idx].cpu().numpy()
        if self.cfg['dtype'] == 'uint16':
            f = f.view(np.uint16)
        return f.astype(np.float64)

    def check(self, res, n_check=32):
        """float64 NumPy on `n_check` of THIS rank's frames against the job's result; raises on a
        relative error > 1e-5 (a kernel that skipped pixels or frames does not post a number)."""
        from libertem_amd import masks as M
        rng = np.random.default_rng(1234 + self.rank)
        idx = np.unique(np.concatenate([[0, self.n_local - 1],
                                        rng.integers(0, self.n_local, n_check)]))
        g0 = self.rank * self.n_local if self.ds.shard is not None else 0
        name = self.name
        worst = 0.0
        if name in ('c2', 'c2-small'):
            got = res['intensity'].data
            assert got.shape == self.nav + (16,) and got.dtype == np.float32
            assert np.all(np.isfinite(got))
            got = got.reshape((-1, 16))
            m64 = self.masks.reshape((16, -1)).astype(np.float64)
            for i in idx:
                ref = m64 @ self._frame(i).reshape(-1)
                worst = max(worst, np.abs(got[g0 + i] - ref).max() / np.abs(ref).max())
            if self.use_oracle:
                # the oracle (CPU restatement of the reference's tiled loop) as the checker, on the
                # same frames -- cpu_baseline leg only
                from oracle import path as opath
                sub = np.stack([self.frames[int(i)].cpu().numpy().view(np.uint16) for i in idx])
                ref = opath.apply_masks(sub.reshape((1, len(idx)) + self.det), self.masks)[0]
                worst = max(worst, float(np.abs(got[g0 + idx] - ref).max() / np.abs(ref).max()))
        elif name == 'c3':
            yy, xx = np.mgrid[0:self.det[0], 0:self.det[1]]
            gy = res.y.raw_data.reshape(-1)
            gx = res.x.raw_data.reshape(-1)
            assert res.y.raw_data.shape == self.nav
            for i in idx[:8]:
                f = self._frame(i).reshape(self.det)
                cy, cx = (f * yy).sum() / f.sum() - 256, (f * xx).sum() / f.sum() - 256
                # the shifts are differences of ~256-sized quotients: absolute floor 1e-5 x 256
                worst = max(worst, abs(gy[g0 + i] - cy) / 256., abs(gx[g0 + i] - cx) / 256.)
        elif name == 'c4':
            dense = M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256,
                                  n_bins=1024, use_sparse=False,
                                  dtype=np.float32).reshape((1024, -1)).astype(np.float64)
            got = res['intensity'].raw_data
            assert got.shape == (self.nav[0] * self.nav[1], 1024)
            for i in idx[:8]:
                ref = dense @ self._frame(i).reshape(-1)
                worst = max(worst, np.abs(got[g0 + i] - ref).max() / np.abs(ref).max())
        elif name == 'c5':
            stack = np.asarray(self.analysis.get_mask_factories()()).reshape((25, -1))
            raw = res.raw_results.reshape((25, -1))
            assert res.raw_results.shape == (1, 25) + self.nav
            for i in idx[:3]:
                ref = stack.astype(np.complex128) @ self._frame(i).reshape(-1)
                worst = max(worst, np.abs(raw[:, g0 + i] - ref).max() / np.abs(ref).max())
        if not worst < 1e-5:
            raise SystemExit(f"bench.py: {name} result check failed: rel err {worst:.3e} vs "
                             f"float64 NumPy")
        return float(worst)


def load_traffic(name, kname, frames_per_launch):
    """HBM bytes per launch from the tracked PMC profile, only if it was taken for exactly this
    config, kernel (label incl. grid) and launch size."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as f:
            entries = json.load(f)['entries']
    except Exception:
        return None, None, None
    for e in entries:
        if e.get('config') == name and e.get('kernel') == kname and \
                int(e.get('frames_per_launch', -1)) == int(frames_per_launch):
            return float(e['hbm_bytes_per_launch']), e.get('source'), \
                (e.get('rocprof_avg_us'), e.get('rocprof_pmc_pass_avg_us'))
    return None, None, None


def load_tail(name, kname, frames_per_launch):
    """(stats-pass, counter-pass) average of a second kernel that runs with every launch of the dominant one"""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as f:
            entries = json.load(f)['entries']
    except Exception:
        return None, None
    for e in entries:
        if e.get('config') == name and e.get('kernel') == kname and \
                int(e.get('frames_per_launch', -1)) == int(frames_per_launch):
            return e.get('rocprof_tail_avg_us'), e.get('rocprof_tail_pmc_pass_avg_us')
    return None, None


def measure(wl, steps, warmup, barrier, hip, n_check=32, traffic_name=None):
    """W untimed + K timed steps of a workload; returns whole-job and dominant-kernel figures."""
    import re
    cfg = wl.cfg
    res = wl.step()
    err = wl.check(res, n_check=n_check)
    del res
    # The oracle / float64 checks above leave the GPU idle for ~1 s and its clocks (sclk / mclk DPM)
    # come back over the next ~20 ms of load -- longer than a handful of C2 steps
    # (profiles/r02_step_ramp.txt).  A fixed number of extra UNTIMED steps (same on every rank)
    # brings the chip to its steady state before the W warmup steps; reported as `preheat_steps`.
    # Like timeit: no cyclic-GC passes inside the timed region (a full collection of a process with
    # torch loaded takes ~80 ms = 50 steps; results are reference counted, nothing accumulates).
    # The collection itself runs BEFORE the untimed steps -- it is another 80 ms of idle GPU.
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    preheat = int(os.environ.get('LTMI_BENCH_PREHEAT', cfg.get('preheat', 0)))
    for _ in range(preheat):
        wl.step()
    for _ in range(max(0, warmup - 1)):
        wl.step()
    hip.KernelTimer.start()
    barrier()
    t0 = time.perf_counter()
    marks = []
    try:
        for _ in range(steps):
            wl.step()
            marks.append(time.perf_counter())
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    if os.environ.get('LTMI_BENCH_DEBUG'):
        ring = getattr(wl.ctx.executor, '_pinned_ring', None)
        print('step ms:', ' '.join(f'{(b - a) * 1e3:.2f}' for a, b in zip([t0] + marks, marks)),
              '| ring slots', len(ring.slots) if ring else None, file=sys.stderr)
    events = hip.KernelTimer.stop()
    pat = re.compile(cfg['kernel'])
    kms = [ms for ms, n, k in events if pat.search(k)]
    kname = next((k for ms, n, k in events if pat.search(k)), '')
    if not kms:
        raise SystemExit(f"bench.py: no {cfg['kernel']} launch was timed for {wl.name} -- kernel "
                         f"name filter out of date? events: {events[:3]!r}")
    launches_per_step = max(1, len(kms) // max(1, steps))
    frames_per_launch = wl.n_local / launches_per_step
    avg_ms = float(np.mean(kms))
    alg_bytes = (wl.n_px * wl.itemsize + cfg['result_bytes']) * frames_per_launch
    gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
    tfs = cfg['flops'] * frames_per_launch / (avg_ms * 1e-3) / 1e12
    traffic, traffic_src, prof_us = load_traffic(traffic_name or wl.name, kname, frames_per_launch)
    tail_us = load_tail(traffic_name or wl.name, kname, frames_per_launch)
    roof = {"bound": cfg['bound']}
    if cfg['bound'] == 'hbm':
        roof.update(achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
        if cfg.get('issued_flops'):
            # C5: SURVEY.md 8(d) prices it against the f32 matrix peak by its ALGORITHMIC flops; since the fold
            # (round 5) the kernel issues half of them, the bytes are the larger fraction and the board's power
            # cap is what binds (profiles/r05_fold.txt) -- the HBM fraction is the one to beat, the two matrix
            # fractions ride beside it
            issued = cfg['issued_flops'] * frames_per_launch / (avg_ms * 1e-3) / 1e12
            roof.update(mfma_peak_TFLOPs=MFMA_F32_PEAK_TF, mfma_algorithmic_frac=tfs / MFMA_F32_PEAK_TF,
                        mfma_issued_TFLOPs=issued, mfma_issued_frac=issued / MFMA_F32_PEAK_TF,
                        binding_resource="board power cap (1400 W; profiles/r05_fold.txt)")
    else:
        roof.update(achieved=tfs, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                    frac=tfs / MFMA_F32_PEAK_TF, hbm_GBps=gbs, hbm_frac=gbs / HBM_PEAK_GBS)
    # the same fraction from the tracked rocprofv3 --kernel-trace average of this kernel (all launches
    # of the profiled run, cold ones included), so that the two can be compared without arithmetic
    prof_us, pmc_us = prof_us if isinstance(prof_us, tuple) else (prof_us, None)
    if tail_us[0] is not None and prof_us:
        prof_us += tail_us[0]                    # (C4: k_bell_tail runs with every k_bell_flat launch)
    if tail_us[1] is not None and pmc_us:
        pmc_us += tail_us[1]
    # C4's --stats average is not the kernel: its dispatches overlap the 64 MiB result copies of the previous
    # tile (1.3 ms against 0.6 ms in the serialised counter passes) -- only the counter-pass figure is quoted
    stats_is_artefact = 'k_bell' in kname or 'k_scatter' in kname
    per_s = (alg_bytes / 1e9 / HBM_PEAK_GBS if cfg['bound'] == 'hbm'
             else cfg['flops'] * frames_per_launch / 1e12 / MFMA_F32_PEAK_TF)
    roof.update(profile_avg_launch_ms=prof_us / 1e3 if prof_us else None,
                frac_from_profile=per_s / (prof_us * 1e-6) if prof_us and not stats_is_artefact else None,
                # the kernel alone (rocprofv3 counter passes serialise the dispatches): without the
                # result copies of the previous tile on the HBM -- differs for C4 only
                profile_pmc_pass_avg_launch_ms=pmc_us / 1e3 if pmc_us else None,
                frac_from_profile_pmc_passes=per_s / (pmc_us * 1e-6) if pmc_us else None)
    roof.update(traffic=traffic, traffic_source=traffic_src, kernel=kname, avg_launch_ms=avg_ms,
                launches_timed=len(kms), frames_per_launch=frames_per_launch,
                algorithmic_bytes_per_launch=alg_bytes,
                algorithmic_flops_per_launch=cfg['flops'] * frames_per_launch,
                algorithmic_TFLOPs=tfs)
    step_ms = [(b - a) * 1e3 for a, b in zip([t0] + marks, marks)]
    roof.update(launch_includes="k_bell_tail (float32 products the float16 image leaves out)"
                if 'tail=' in kname and 'k_bell_flat' in kname else None)
    return dict(elapsed=elapsed, ms_per_step=elapsed / steps * 1e3, roofline=roof,
                ms_per_step_median=float(np.median(step_ms)) if step_ms else None,
                kernel_ms_per_step=float(np.sum(kms)) / steps, check_rel_err=err,
                preheat_steps=preheat)


def host_streamed(ctx, torch, hip, gib=2):
    """Timing mode (ii) of SURVEY.md 8(d) for every config: frames in HOST memory (page-locked in
    place), double-buffered hipMemcpyAsync overlapping the kernels, `gib` GiB per config; against the
    H2D peak measured in the same process (one pinned 1 GiB buffer, plain hipMemcpyAsync)."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as M
    nbytes = gib << 30
    rng = np.random.default_rng(1)
    u16 = rng.integers(0, 4096, nbytes // 2, dtype=np.uint16)
    pinned = torch.empty((1 << 30,), dtype=torch.uint8, pin_memory=True)
    dev = torch.empty((1 << 30,), dtype=torch.uint8, device='cuda')
    dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    peak = 4 * (1 << 30) / (time.perf_counter() - t0) / 1e9
    del pinned, dev

    def timed(step, n_frames, data_bytes, what, ds=None):
        for _ in range(2):
            step()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        gbs = data_bytes / t / 1e9
        # which way the bytes went: DMA straight out of the user's array (its memory provably is a mapping of its own,
        # io/dataset/memory.py `_own_mapping`) or staged through the page-locked bounce buffers by ltmi_host_copy
        st = next(iter(getattr(ds, '__dict__', {}).get('_hip_stagers', {}).values()), None) if ds is not None else None
        upload = None if st is None else ("in place (hipHostRegister: the array is a mapping of its own)"
                                          if st.registered is not None else "staged (bounce buffers + ltmi_host_copy)")
        return {"workload": what, "frames_per_s": n_frames / t, "GBps": gbs, "upload": upload,
                "h2d_peak_GBps": peak, "frac_of_h2d_peak": gbs / peak, "ms_per_run": t * 1e3}

    out = {}
    masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
    n2 = nbytes // (256 * 256 * 2)
    ds2 = ctx.load('memory', data=u16.reshape((n2 // 256, 256, 256, 256)), sig_dims=2, num_partitions=1)
    udf2 = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                         mask_dtype=np.float32)
    out['c2'] = timed(lambda: ctx.run_udf(dataset=ds2, udf=udf2), n2, nbytes,
                      f"C2 masks, {n2} frames ({gib} GiB) in host memory, double-buffered hipMemcpyAsync", ds=ds2)
    # the same frames in memory whose mapping the library cannot vouch for (another allocator's): the staged path
    keep = torch.from_numpy(u16.view(np.int16)).clone()
    foreign = keep.numpy().view(np.uint16).reshape((n2 // 256, 256, 256, 256))
    ds2s = ctx.load('memory', data=foreign, sig_dims=2, num_partitions=1)
    out['c2_staged'] = timed(lambda: ctx.run_udf(dataset=ds2s, udf=udf2), n2, nbytes,
                             f"C2 masks, {n2} frames ({gib} GiB) in a torch CPU tensor's memory", ds=ds2s)
    ds2s.close_stagers()
    del ds2s, foreign, keep
    n3 = nbytes // (512 * 512 * 2)
    ds3 = ctx.load('memory', data=u16.reshape((n3 // 512, 512, 512, 512)), sig_dims=2, num_partitions=1)
    an3 = ctx.create_com_analysis(dataset=ds3, cx=256, cy=256)
    out['c3'] = timed(lambda: ctx.run(an3), n3, nbytes,
                      f"C3 CoM analysis, {n3} frames of 512x512 uint16 ({gib} GiB) in host memory", ds=ds3)

    def rings():
        return M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256,
                             n_bins=1024, use_sparse=True, dtype=np.float32)
    udf4 = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=1024,
                         mask_dtype=np.float32)
    out['c4'] = timed(lambda: ctx.run_udf(dataset=ds2, udf=udf4), n2, nbytes,
                      f"C4 ring stack, {n2} frames of 256x256 uint16 ({gib} GiB) in host memory "
                      f"(+ {n2 * 4096 / 2**20:.0f} MiB of results back)", ds=ds2)
    del ds2, ds3, u16
    f32 = rng.random(nbytes // 4, dtype=np.float32)
    n5 = nbytes // (1024 * 1024 * 4)
    ds5 = ctx.load('memory', data=f32.reshape((n5 // 128, 128, 1024, 1024)) if n5 >= 128 else
                   f32.reshape((1, n5, 1024, 1024)), sig_dims=2, num_partitions=1)
    an5 = ctx.create_radial_fourier_analysis(dataset=ds5)
    out['c5'] = timed(lambda: ctx.run(an5), n5, nbytes,
                      f"C5 radial Fourier analysis, {n5} frames of 1024x1024 float32 ({gib} GiB) in "
                      f"host memory", ds=ds5)
    return out


def small_tiles(torch, hip, sizes=(1024, 4096, 8192)):
    """What a live feed produces: C2 launches of a few thousand frames (ltmi_apply_masks on resident
    tiles, HIP events on the launch stream around back-to-back launches)."""
    masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
    h = hip.MaskHandle.dense(0, masks, np.float32)
    g = torch.Generator(device='cuda').manual_seed(3)
    tile = torch.randint(0, 4096, (max(sizes), 65536), generator=g, device='cuda', dtype=torch.int16)
    out_t = torch.zeros((max(sizes), 16), device='cuda', dtype=torch.float32)
    res = {}
    for n in sizes:
        for _ in range(3):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out_t.data_ptr(), 16, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out_t.data_ptr(), 16, False)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = n * (131072 + 64) / ms / 1e6
        res[str(n)] = {"kernel": h.last_kernel(), "avg_launch_ms": ms, "frames_per_s": n / ms * 1e3,
                       "GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
    ref = tile[:4].cpu().numpy().view(np.uint16).astype(np.float64) @ masks.T.astype(np.float64)
    err = float(np.abs(out_t[:4].cpu().numpy() - ref).max() / np.abs(ref).max())
    if not err < 1e-5:
        raise SystemExit(f"bench.py: small-tile check failed: {err:.3e}")
    res["check_rel_err_vs_float64"] = err
    return res


def live_feed(ctx, n_frames=16384, chunk=1024):
    """Row f4: a running acquisition -- frames arrive in chunks of `chunk` from a feeder thread
    (StreamDataSet), partial results after every partition (Context.run_udf_iter); rate over the whole
    scan incl. the uploads."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 4096, (n_frames, 256, 256), dtype=np.uint16)
    masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                        mask_dtype=np.float32)
    ts, n_parts = [], 0
    for rep in range(5):                 # (the first two scans still build plans / page-lock buffers)
        ds = ctx.load('stream', frames=(frames[i:i + chunk] for i in range(0, n_frames, chunk)),
                      nav_shape=(n_frames // 256, 256), sig_shape=(256, 256), dtype=np.uint16,
                      num_partitions=n_frames // chunk)
        t0 = time.perf_counter()
        n_parts = 0
        for part in ctx.run_udf_iter(dataset=ds, udf=udf):
            n_parts += 1
        ts.append(time.perf_counter() - t0)
        last = np.array(part.buffers[0]['intensity'].data)
    t = float(np.median(ts[2:]))
    ref = frames[-1].reshape(-1).astype(np.float64) @ masks.reshape((16, -1)).T.astype(np.float64)
    err = float(np.abs(last.reshape((-1, 16))[-1] - ref).max() / np.abs(ref).max())
    if not err < 1e-5:
        raise SystemExit(f"bench.py: live-feed check failed: {err:.3e}")
    # the same scan with an in-place feed: the acquisition writes into the page-locked scan buffer itself
    # (a detector's DMA target -- emulated: the buffer holds the frames, a producer thread publishes chunk
    # after chunk as fast as the consumer takes them; no feeder memcpy on this side)
    import threading
    ts2 = []
    for rep in range(5):
        ds = ctx.load('stream', frames=None, nav_shape=(n_frames // 256, 256), sig_shape=(256, 256),
                      dtype=np.uint16, num_partitions=n_frames // chunk)
        ds.scan_buffer[...] = frames

        def produce(ds=ds):
            for i in range(chunk, n_frames + 1, chunk):
                ds.commit(i)
        t0 = time.perf_counter()
        th = threading.Thread(target=produce)
        th.start()
        for part in ctx.run_udf_iter(dataset=ds, udf=udf):
            pass
        ts2.append(time.perf_counter() - t0)
        th.join()
        last2 = np.array(part.buffers[0]['intensity'].data)
    t2 = float(np.median(ts2[2:]))
    err2 = float(np.abs(last2.reshape((-1, 16))[-1] - ref).max() / np.abs(ref).max())
    if not err2 < 1e-5:
        raise SystemExit(f"bench.py: in-place live-feed check failed: {err2:.3e}")
    return {"workload": f"C2 masks on {n_frames} frames arriving in chunks of {chunk} "
                        f"(StreamDataSet + run_udf_iter, {n_parts} partial results)",
            "frames_per_s": n_frames / t, "GBps": frames.nbytes / t / 1e9, "ms_per_scan": t * 1e3,
            "check_rel_err_vs_float64": err,
            "in_place": {"note": "frames=None: the producer writes the page-locked scan buffer itself and "
                                 "commits chunks (emulated: pre-filled buffer, commits as fast as they are "
                                 "taken) -- the consumer side of a detector that DMAs into host memory",
                         "frames_per_s": n_frames / t2, "GBps": frames.nbytes / t2 / 1e9,
                         "ms_per_scan": t2 * 1e3, "check_rel_err_vs_float64": err2}}


def crystallinity(torch, hip, reps=10):
    """Row f3: CrystallinityUDF's kernel on frames resident in HBM -- sum(abs(rfft2(frame * real_mask)) * ring)
    per frame (ltmi_crystallinity; uint16 frames, ring sig/16 .. sig/4, real-space disk of radius sig/10 masked
    out) for 256 x 256 frames (k_cryst_fused), 128 x 128 frames (k_cryst_fused128) 512 x 512 and 1024 x 1024 frames
    (k_cryst_rows<N> + k_cryst_cols<N>), the hipFFT route of the same call beside each (LTMI_FFT_FUSED is read per plan)."""
    from libertem_amd.udf.crystallinity import crystallinity_masks, mask_box
    res = {"bound": "vector ALUs + LDS (profiles/r04_crystallinity.txt, r05_crystallinity.txt): the pixels are read once"}
    for sig, n in ((256, 16384), (128, 65536), (512, 4096), (1024, 1024)):
        g = torch.Generator(device='cuda').manual_seed(1)
        frames = torch.randint(0, 4096, (n, sig, sig), generator=g, device='cuda', dtype=torch.int16)
        real_mask, half = crystallinity_masks((sig, sig), sig // 16, sig // 4, (sig // 2, sig // 2), sig // 10)
        rm = torch.from_numpy(np.ascontiguousarray(real_mask.astype(np.float32))).cuda()
        hm = torch.from_numpy(np.ascontiguousarray(half.astype(np.float32))).cuda()
        out = torch.zeros(n, dtype=torch.float32, device='cuda')
        box = mask_box(half)
        r = {"workload": f"{n} frames of {sig}x{sig} uint16, ring {sig // 16}..{sig // 4} of the half spectrum, "
                         "real-space disk masked"}
        for key, env in (("fused", None), ("hipfft_route", "0")):
            old = os.environ.get('LTMI_FFT_FUSED')
            if env is not None:
                os.environ['LTMI_FFT_FUSED'] = env
            try:
                plan = hip.FFTPlan(0, sig, sig, 1024)
            finally:
                if env is not None:
                    if old is None:
                        os.environ.pop('LTMI_FFT_FUSED', None)
                    else:
                        os.environ['LTMI_FFT_FUSED'] = old

            def run():
                plan.crystallinity(frames.data_ptr(), np.uint16, n, sig * sig, rm.data_ptr(), hm.data_ptr(), box,
                                   out.data_ptr(), False)
            run(); run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            got = out[[0, n - 1]].cpu().numpy()
            fr = frames[[0, n - 1]].cpu().numpy().view(np.uint16).astype(np.float64)
            ref = np.array([np.sum(abs(np.fft.rfft2(f * real_mask)) * half) for f in fr])
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            if not err < 1e-5:
                raise SystemExit(f"bench.py: crystallinity check failed ({sig}, {key}): {err:.3e}")
            nbytes = n * sig * sig * 2
            r[key] = {"kernel": plan.last_kernel(), "avg_call_ms": ms, "frames_per_s": n / ms * 1e3,
                      "pixel_GBps": nbytes / ms / 1e6, "frac_of_hbm_peak": nbytes / ms / 1e6 / HBM_PEAK_GBS,
                      "check_rel_err_vs_float64": err}
            plan.close()
        r["speedup"] = r["hipfft_route"]["avg_call_ms"] / r["fused"]["avg_call_ms"]
        # RAW frames with detector corrections (dark + gain + 50 dead pixels): inside the row stage of the fused
        # kernels (round 5), through the conversion pass of round 4, through hipFFT
        try:
            from libertem_amd.io.corrections import CorrectionSet
            from oracle import corrections as oc
            crng = np.random.default_rng(3)
            bad = np.zeros((sig, sig), dtype=bool)
            bad[crng.integers(0, sig, 50), crng.integers(0, sig, 50)] = True
            dark, gain = crng.random((sig, sig)) * 6, crng.random((sig, sig)) * 0.6 + 0.7
            tables = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad).device_tables(0, (sig, sig))
            fr2 = frames[[0, n - 1]].cpu().numpy().view(np.uint16)
            fixed = oc.correct(fr2, (sig, sig), dark=dark, gain=gain,
                               coords=[tuple(c) for c in np.argwhere(bad)]).reshape(2, sig, sig)
            ref_c = np.array([np.sum(abs(np.fft.rfft2(f.astype(np.float64) * real_mask)) * half) for f in fixed])
            corr = {}
            for key, envs in (("row_stage", {}), ("conversion_pass", {"LTMI_CRYST_CORR_PASS": "1"}),
                              ("hipfft_route", {"LTMI_FFT_FUSED": "0"})):
                saved = {k: os.environ.get(k) for k in envs}
                os.environ.update(envs)
                try:
                    plan = hip.FFTPlan(0, sig, sig, 1024)

                    def run_c():
                        plan.crystallinity_corrected(frames.data_ptr(), np.uint16, n, sig * sig, tables, rm.data_ptr(),
                                                     hm.data_ptr(), box, out.data_ptr(), False)
                    run_c(); run_c()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        run_c()
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                    got = out[[0, n - 1]].cpu().numpy()
                    err = float(np.abs(got - ref_c).max() / np.abs(ref_c).max())
                    if not err < 1e-5:
                        raise SystemExit(f"bench.py: corrected crystallinity check failed ({sig}, {key}): {err:.3e}")
                    corr[key] = {"kernel": plan.last_kernel(), "avg_call_ms": ms, "frames_per_s": n / ms * 1e3,
                                 "check_rel_err_vs_float64": err}
                    plan.close()
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            corr["speedup_vs_hipfft"] = corr["hipfft_route"]["avg_call_ms"] / corr["row_stage"]["avg_call_ms"]
            r["corrected_raw_frames"] = corr
        except BaseException as e:                    # noqa: BLE001  (never sinks the entry)
            r["corrected_raw_frames"] = {"error": repr(e)[:300]}
        res[f"frames_{sig}"] = r
        del frames, out
        torch.cuda.empty_cache()
    return res


def second_runs(ctx, torch, hip, reps=3):
    """SURVEY.md 8(d)'s second runs of C3 and C5: CoM with mask_radius=200, radial Fourier with n_bins=16, max_order=24,
    use_sparse=True (400 complex64 masks, a stack of dense column blocks).  Whole job through Context.run on a nav
    subset, the kernels through hip.KernelTimer, a few frames against float64 NumPy."""
    import gc
    out = {}

    def timed(an, n, frame_bytes):
        res = ctx.run(an)
        for _ in range(2):
            ctx.run(an)
        hip.KernelTimer.start()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.run(an)
        dt = (time.perf_counter() - t0) / reps
        ev = hip.KernelTimer.stop()
        by = {}
        for ms, cnt, k in ev:
            by.setdefault(k, []).append(ms)
        kernels = [{"kernel": k, "launches_per_run": len(v) // reps, "avg_launch_ms": float(np.mean(v))}
                   for k, v in by.items()]
        return res, {"ms_per_run": dt * 1e3, "frames_per_s": n / dt, "input_GBps_whole_job": n * frame_bytes / dt / 1e9,
                     "kernels": kernels}

    # C3, mask_radius=200
    n = 16384
    fr = device_frames(torch, n, (512, 512), 'uint16', 31)
    ds = ctx.load('memory', data=fr.reshape((n // 256, 256, 512, 512)), dtype=np.dtype('uint16'), sig_dims=2,
                  num_partitions=1)
    an = ctx.create_com_analysis(dataset=ds, cx=256, cy=256, mask_radius=200)
    res, rec = timed(an, n, 512 * 512 * 2)
    yy, xx = np.mgrid[0:512, 0:512]
    disk = ((yy - 256) ** 2 + (xx - 256) ** 2) <= 200 ** 2
    worst = 0.
    for i in (0, 777, n - 1):
        f = fr[i].cpu().numpy().view(np.uint16).astype(np.float64).reshape((512, 512)) * disk
        cy, cx = (f * yy).sum() / f.sum() - 256, (f * xx).sum() / f.sum() - 256
        worst = max(worst, abs(res.y.raw_data.reshape(-1)[i] - cy) / 256., abs(res.x.raw_data.reshape(-1)[i] - cx) / 256.)
    if not worst < 1e-5:
        raise SystemExit(f"bench.py: C3 mask_radius=200 check failed: {worst:.3e}")
    rec.update(workload=f"CoM analysis, cx=cy=256, mask_radius=200, {n} frames of 512x512 uint16",
               check_rel_err_vs_float64=float(worst),
               frac_of_hbm_peak_kernel=n * 512 * 512 * 2 / (sum(k['avg_launch_ms'] * k['launches_per_run']
                                                                for k in rec['kernels']) * 1e-3) / 1e9 / HBM_PEAK_GBS)
    out['c3_mask_radius_200'] = rec
    del fr, ds, an, res
    gc.collect()
    torch.cuda.empty_cache()

    # C5, 16 bins x 25 orders, sparse
    n = 4096
    fr = device_frames(torch, n, (1024, 1024), 'float32', 32)
    ds = ctx.load('memory', data=fr.reshape((n // 128, 128, 1024, 1024)), dtype=np.dtype('float32'), sig_dims=2,
                  num_partitions=1)
    an = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=16, max_order=24, use_sparse=True)
    res, rec = timed(an, n, 1024 * 1024 * 4)
    stack = an.get_mask_factories()().to_px_by_masks(dtype=np.complex64)          # (n_px, 400) CSR
    raw = res.raw_results.reshape((400, -1))
    worst = 0.
    for i in (0, n - 1):
        f = fr[i].cpu().numpy().astype(np.float64).reshape(-1)
        ref = np.asarray(stack.T.astype(np.complex128) @ f).reshape(-1)
        worst = max(worst, float(np.abs(raw[:, i] - ref).max() / np.abs(ref).max()))
    if not worst < 1e-5:
        raise SystemExit(f"bench.py: C5 sparse radial Fourier check failed: {worst:.3e}")
    rec.update(workload=f"radial Fourier analysis, n_bins=16, max_order=24, use_sparse=True (400 complex64 masks, "
                        f"nnz {stack.nnz}), {n} frames of 1024x1024 float32",
               check_rel_err_vs_float64=worst,
               useful_TFLOPs_kernel=4 * stack.nnz * n / (sum(k['avg_launch_ms'] * k['launches_per_run']
                                                             for k in rec['kernels']) * 1e-3) / 1e12)
    out['c5_sparse_16_bins'] = rec
    del fr, ds, an, res
    gc.collect()
    torch.cuda.empty_cache()

    # ... the same stack on uint16 frames (what a detector delivers)
    fr = device_frames(torch, n, (1024, 1024), 'uint16', 33)
    ds = ctx.load('memory', data=fr.reshape((n // 128, 128, 1024, 1024)), dtype=np.dtype('uint16'), sig_dims=2,
                  num_partitions=1)
    an = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=16, max_order=24, use_sparse=True)
    res, rec = timed(an, n, 1024 * 1024 * 2)
    raw = res.raw_results.reshape((400, -1))
    worst = 0.
    for i in (0, n - 1):
        f = fr[i].cpu().numpy().view(np.uint16).astype(np.float64).reshape(-1)
        ref = np.asarray(stack.T.astype(np.complex128) @ f).reshape(-1)
        worst = max(worst, float(np.abs(raw[:, i] - ref).max() / np.abs(ref).max()))
    if not worst < 1e-5:
        raise SystemExit(f"bench.py: C5 sparse radial Fourier (uint16) check failed: {worst:.3e}")
    rec.update(workload=f"the same analysis on {n} frames of 1024x1024 uint16", check_rel_err_vs_float64=worst)
    out['c5_sparse_16_bins_uint16'] = rec
    del fr, ds, an, res
    gc.collect()
    torch.cuda.empty_cache()
    return out


def mib_decode(torch, hip, n=16384, reps=10):
    """Row f2: .mib frames (12-bit raw words, 256 x 256, 384-byte headers) -> uint16 frames, file bytes
    already in HBM; rate = (payload read + frames written) / kernel time (HIP events on its stream)."""
    from libertem_amd.common.hiparray import HipArray
    h = w = 256
    header, payload = 384, h * w * 2
    stride = header + payload
    raw = torch.randint(0, 256, (n * stride,), dtype=torch.uint8, device='cuda')
    out = HipArray.empty((n, h, w), np.uint16, 0)
    s = torch.cuda.current_stream()

    def run():
        hip.mib_decode(0, raw.data_ptr(), stride, header, 'r', 12, False, n, h, w, out.data_ptr(),
                       np.uint16, stream=s.cuda_stream)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        run()
    e1.record(s)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # spot check of the last frame against the format's definition (big-endian words, pixels of a
    # word in reverse order)
    last = raw[(n - 1) * stride + header:(n - 1) * stride + stride].cpu().numpy()
    expect = last.view('>u2').astype(np.uint16).reshape(-1, 4)[:, ::-1].reshape(h, w)
    ok = bool(np.array_equal(out.rows(n - 1, n).cpu().reshape(h, w), expect))
    nbytes = n * (payload + h * w * 2)
    traffic, source, _ = load_traffic('mib_decode', 'k_mib_decode16', n)
    return {"workload": f"{n} raw 12-bit .mib frames of 256x256 -> uint16 (ltmi_mib_decode)",
            "kernel": "k_mib_decode16", "avg_launch_ms": ms, "frames_per_s": n / ms * 1e3,
            "check_last_frame": ok,
            "roofline": {"bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": float(nbytes),
                         "traffic": traffic, "traffic_source": source}}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command line under
    torch.distributed.run (N ranks on this node, rendezvous on 127.0.0.1 at a free port); the ranks
    inherit stdout / stderr, so the ONE JSON line of rank 0 is this process's output.  Returns the
    launcher's exit code."""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', port,
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--result-via', default='auto', choices=['auto', 'shm', 'rccl'],
                    help="delivery of the nav results across ranks: the node-shared page-locked host segment (shm) or "
                         "the RCCL all-gather over xGMI (rccl); auto = the executor's default (shm on one node).  At "
                         "--gpus 1 a value other than auto forces the multi-rank delivery path with a world of one "
                         "(LTMI_FORCE_COLLECTIVES=1): the N = 1 anchor of that path")
    args = ap.parse_args()
    if args.result_via != 'auto':
        os.environ['LTMI_RESULT_VIA'] = args.result_via
        if args.gpus == 1:
            os.environ['LTMI_FORCE_COLLECTIVES'] = '1'
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and let
        # rank 0's JSON line through on the inherited stdout
        raise SystemExit(self_launch(args.gpus))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test hooks (one-GPU box: several gloo ranks on one device exercise the N>1 flow end to end)
    device_id = int(os.environ.get('LTMI_BENCH_DEVICE', local_rank))
    backend = os.environ.get('LTMI_BENCH_BACKEND', 'nccl')
    extras = not args.no_extras and os.environ.get('LTMI_BENCH_EXTRAS', '1') != '0' \
        and args.config == 'c2'
    cpu_base, cpu_all = None, {}
    if world == 1 and not args.no_cpu_baseline:
        # rank 0 at N=1 only, and BEFORE the HIP runtime is initialised (fork-safe): the measured config, and --
        # with the extras -- every other config whose line rides along (the CPU path beside each)
        names = [args.config if args.config in _CPU_SAMPLES else 'c2']
        if extras:
            names += [n for n in ('c3', 'c4', 'c5') if n not in names]
        cpu_all = cpu_baselines(names, {n: (12.0 if n == names[0] else 6.0) for n in names})
        cpu_base = cpu_all[names[0]]
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(device_id)
    use_dist = world > 1 or os.environ.get('LTMI_FORCE_COLLECTIVES') == '1'
    if use_dist:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # a rank that dies must take the job down within minutes, not hang it
        tmo = datetime.timedelta(seconds=int(os.environ.get('LTMI_BENCH_TIMEOUT_S', '300')))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device_id), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    import logging
    logging.getLogger('libertem_amd').setLevel(logging.ERROR)     # stdout carries ONE JSON line
    from libertem_amd.api import Context
    from libertem_amd import hip

    ctx = Context.make_with('hip', gpus=device_id)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device='cuda')
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_list(x):
        if not use_dist:
            return [x]
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    # ---- the measured config: one global dataset of world x (scan) frames, every rank holds its
    # ---- own contiguous block (weak scaling)
    # (test hook: fewer frames per rank -- many ranks on one GPU stand in for a node; recorded in the line)
    fpr = os.environ.get('LTMI_BENCH_FRAMES_PER_RANK')
    fpr = int(fpr) if fpr else None
    wl = Workload(args.config, ctx, torch, rank, world, sharded=use_dist, frames_per_rank=fpr)
    wl.use_oracle = cpu_base is not None           # N = 1: the oracle also checks 32 result rows
    m = measure(wl, args.steps, args.warmup, barrier, hip)
    result_via = getattr(ctx.executor, 'last_result_via', 'local')
    elapsed_max = max_over_ranks(m['elapsed'])
    per_rank = gather_list({"rank": rank, "ms_per_step": m['ms_per_step'],
                            "kernel_ms_per_step": m['kernel_ms_per_step'],
                            "kernel_avg_launch_ms": m['roofline']['avg_launch_ms']})
    n_frames, n_px, itemsize = wl.n_local, wl.n_px, wl.itemsize
    total_frames = n_frames * world * args.steps
    value = total_frames / elapsed_max

    # the same K steps on the float32 matrix instruction only (v_mfma_f32_16x16x4_f32: the strict reading
    # of "MFMA f32" in the north star), same handle, same plan -- every rank takes this path
    f32_leg = None
    if args.config.startswith('c2'):
        os.environ['LTMI_DENSE_F32_INSTR'] = '1'
        try:
            mf = measure(wl, args.steps, args.warmup, barrier, hip, n_check=8, traffic_name='c2_f32')
            el = max_over_ranks(mf['elapsed'])
            rf = mf['roofline']
            f32_leg = {"instruction": "v_mfma_f32_16x16x4_f32", "steps": args.steps,
                       "ms_per_step": el / args.steps * 1e3,
                       "value": n_frames * world * args.steps / el, "unit": "frames/s",
                       "whole_job_frac_of_hbm": n_frames * world * args.steps / el * n_px * itemsize
                       / 1e9 / HBM_PEAK_GBS / world,
                       "kernel": rf['kernel'], "kernel_avg_launch_ms": rf['avg_launch_ms'],
                       "kernel_frac": rf['frac'], "frac_from_profile": rf.get('frac_from_profile'),
                       "frac_from_profile_pmc_passes": rf.get('frac_from_profile_pmc_passes'),
                       "traffic": rf.get('traffic'), "ms_per_step_median": mf.get('ms_per_step_median'),
                       "check_rel_err_vs_float64": mf['check_rel_err']}
            if ',f16' in rf['kernel']:
                f32_leg["error"] = "the float16-piece kernel ran"
        except BaseException as e:                        # noqa: BLE001  (never sinks the line)
            f32_leg = {"error": repr(e)[:300]}
        finally:
            del os.environ['LTMI_DENSE_F32_INSTR']

    extra = {}
    out = {
        "metric": "frames/sec, ApplyMasksUDF 16 dense f32 masks (+ GB/s vs HBM roofline)"
                  if args.config.startswith('c2') else f"frames/sec, {cfg['desc']}",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preheat_steps": m['preheat_steps'],
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "ms_per_step_median_rank0": m.get('ms_per_step_median'),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (device-generated, resident in HBM: uint16 counts in [0,4096) / "
                "float32 in [0,1))",
        "config": {
            "workload": cfg['desc'] + f", per GPU; {world} GPU(s), nav-sharded (weak)",
            "frames_per_gpu": n_frames, "frame_bytes": n_px * itemsize,
            "arithmetic": "f16x2-piece products, f32 accumulate, 22-bit weights",
            "arithmetic_detail": "uint16 pixel bytes x two f16 pieces of each f32 weight, exact products, f32 sums "
                                 "(v_mfma_f32_16x16x32_f16); f32_instr_* keys: same steps on v_mfma_f32_16x16x4_f32",
            "step": "Context.run_udf / Context.run (plan + kernels + delivery of the complete "
                    "result to every rank's host)",
            "parallelism": f"nav-shard x{world}; results via {result_via}",
        },
        "input_GBps_whole_job": value * n_px * itemsize / 1e9,
        "result_check_rel_err_vs_float64": m['check_rel_err'],
        # (the strict float32-instruction leg as FLAT scalars inside roofline: a record that keeps only scalar values
        #  of the contract's keys still carries it)
        "roofline": dict(m['roofline'], **({
            "f32_instr_kernel_frac": f32_leg.get('kernel_frac'),
            "f32_instr_frac_from_profile": f32_leg.get('frac_from_profile'),
            "f32_instr_frac_from_profile_pmc_passes": f32_leg.get('frac_from_profile_pmc_passes'),
            "f32_instr_kernel_avg_launch_ms": f32_leg.get('kernel_avg_launch_ms'),
            "f32_instr_ms_per_step": f32_leg.get('ms_per_step'),
            "f32_instr_whole_job_frac_of_hbm": f32_leg.get('whole_job_frac_of_hbm'),
            "f32_instr_rel_err": f32_leg.get('check_rel_err_vs_float64'),
        } if isinstance(f32_leg, dict) and 'error' not in f32_leg else {})),
        "f32_instruction": f32_leg,
        "result_via": result_via,
        "value_path": {"shm": "every rank's kernels write its nav rows into a page-locked host segment all "
                              "ranks of the node map (no data-path collective); extras.rccl_path times "
                              "the same steps with the RCCL all-gather over xGMI",
                       "collective": "nav rows gathered on the devices by RCCL (ltmi_comm_all_gather over "
                                     "xGMI), one D2H per rank",
                       "local": "single rank: kernels write into the rank's own page-locked buffer"
                       }.get(result_via, result_via),
        "launch_ahead": hip.LaunchReplay.n_ahead,
        "per_rank": per_rank,
    }
    if fpr is not None:
        out["test_hook_frames_per_rank"] = fpr

    # The headline figure is complete here.  The extras below run several more collectives (RCCL through
    # the library's communicator, a 128 GiB strong-scaling set-up); should one of them hang on a box this
    # was never run on, the run must still yield its line: a watchdog on every rank prints the line
    # without the unfinished extras (rank 0) and ends the process after LTMI_BENCH_EXTRAS_TIMEOUT_S.
    import threading
    printed = threading.Event()

    def emit(reason=None):
        if printed.is_set():
            return
        printed.set()
        if rank == 0:
            line = dict(out)
            line.update(extra)
            if reason:
                line["extras_incomplete"] = reason
            if cpu_base is not None:
                line["cpu_baseline"] = cpu_base
            print(json.dumps(line), flush=True)

    def bark():
        emit(f"extras did not finish within {wd_s} s")
        os._exit(0)
    wd_s = int(os.environ.get('LTMI_BENCH_EXTRAS_TIMEOUT_S', '600' if world > 1 else '900'))
    watchdog = threading.Timer(wd_s, bark)
    watchdog.daemon = True
    if extras:
        watchdog.start()


    def guarded(key, fn):
        """extras never sink the line; every rank takes the same path (the code below is
        deterministic and symmetric), an exception is recorded instead of raised"""
        try:
            extra[key] = fn()
        except BaseException as e:                    # noqa: BLE001  (SystemExit of a failed check too)
            extra[key] = {"error": repr(e)[:300]}

    if extras and world > 1:
        # (a) the same steps with the nav results gathered by RCCL over xGMI (north star) instead of
        #     the node-shared host segment
        def rccl_path():
            if result_via != 'shm':
                return {"skipped": f"default delivery already is {result_via!r}"}
            os.environ['LTMI_RESULT_VIA'] = 'rccl'
            try:
                k = max(3, args.steps // 2)
                mm = measure(wl, k, 2, barrier, hip, n_check=4)
                el = max_over_ranks(mm['elapsed'])
                try:
                    lib_info = hip.Comm.library_info()
                except Exception as e:                    # noqa: BLE001
                    lib_info = {"error": repr(e)[:200]}
                return {"result_via": getattr(ctx.executor, 'last_result_via', None),
                        "collective": getattr(ctx.executor, 'last_collective', None),
                        "rccl_lib": lib_info,
                        "steps": k, "ms_per_step": el / k * 1e3,
                        "value": n_frames * world * k / el, "unit": "frames/s"}
            finally:
                del os.environ['LTMI_RESULT_VIA']
        guarded('rccl_path', rccl_path)

    if extras and use_dist:
        # (b) strong scaling: C3 (CoM analysis, 512x512 scan x 512x512 uint16 = 128 GiB in total)
        #     nav-split over the ranks; value = the FIXED 262144 frames / max-over-ranks time
        def strong_c3():
            c3 = CONFIGS['c3']
            total = c3['scan'][0] * c3['scan'][1]
            if fpr is not None:
                total = min(total, fpr * world // 4 // c3['scan'][1] * c3['scan'][1])     # (test hook)
            if total % (world * c3['scan'][1]) != 0:
                return {"skipped": f"512 scan rows do not split over {world} ranks"}
            w3 = Workload('c3', ctx, torch, rank, world, sharded=True,
                          frames_per_rank=total // world)
            k = 5
            mm = measure(w3, k, 2, barrier, hip, n_check=4)
            el = max_over_ranks(mm['elapsed'])
            return {"workload": c3['desc'] + f", nav-split over {world} GPU(s)",
                    "scaling": "strong", "frames_total": total, "steps": k,
                    "ms_per_step": el / k * 1e3, "value": total * k / el, "unit": "frames/s",
                    "input_GBps": total * k / el * 512 * 512 * 2 / 1e9,
                    "result_via": getattr(ctx.executor, 'last_result_via', None),
                    "kernel_avg_launch_ms_rank0": mm['roofline']['avg_launch_ms']}
        del wl
        torch.cuda.empty_cache()
        guarded('strong_c3', strong_c3)
    elif extras:
        del wl
        torch.cuda.empty_cache()

    if extras and world == 1 and not use_dist:
        # (c) the other BASELINE.json configs, whole job + dominant kernel, device-resident
        def one_config(name):
            def run():
                w = Workload(name, ctx, torch, 0, 1, sharded=False)
                k = 5
                mm = measure(w, k, 3, barrier, hip, n_check=8)
                fps = w.n_local * k / mm['elapsed']
                extra_keys = {}
                if hasattr(w, 'step_device'):
                    # C4: the host link bounds the job (256 MiB result); report it without the D2H
                    import gc
                    for _ in range(3):
                        r = w.step_device()
                    dev_err = w.check(r, n_check=4)
                    del r
                    gc.collect()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(k):
                        w.step_device()
                    torch.cuda.synchronize()
                    t_dev = (time.perf_counter() - t0) / k
                    extra_keys = {"result_on_device": {
                        "ms_per_step": t_dev * 1e3, "value": w.n_local / t_dev, "unit": "frames/s",
                        "check_rel_err_vs_float64": dev_err,
                        "note": "Context.run_udf(result_where='device'): the result stays in HBM as "
                                "a HipArray, no D2H"}}
                return {"workload": CONFIGS[name]['desc'], "steps": k, **extra_keys,
                        "cpu_baseline": cpu_all.get(name),
                        "ms_per_step": mm['ms_per_step'], "value": fps, "unit": "frames/s",
                        "input_GBps_whole_job": fps * w.n_px * w.itemsize / 1e9,
                        "kernel_ms_per_step": mm['kernel_ms_per_step'],
                        "check_rel_err_vs_float64": mm['check_rel_err'],
                        "preheat_steps": mm['preheat_steps'],
                        "roofline": mm['roofline']}
            return run
        cfgs = {}
        for name in ('c3', 'c4', 'c5'):
            try:
                cfgs[name] = one_config(name)()
            except BaseException as e:                # noqa: BLE001
                cfgs[name] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        extra['configs'] = cfgs
        guarded('host_streamed', lambda: host_streamed(ctx, torch, hip))
        guarded('small_tiles', lambda: small_tiles(torch, hip))
        guarded('live_feed', lambda: live_feed(ctx))
        guarded('mib_decode', lambda: mib_decode(torch, hip))
        guarded('crystallinity', lambda: crystallinity(torch, hip))
        guarded('second_runs', lambda: second_runs(ctx, torch, hip))

        # (d) the N = 1 anchors of BOTH multi-rank delivery paths: the same C2 steps with the collectives forced on a
        #     world of one -- what the first real 2 / 4 / 8-GPU lines (shm by default, RCCL as extras.rccl_path) are to
        #     be compared with, like for like.  Separate processes: the process group has to exist from the start.
        def delivery_anchor():
            import subprocess
            out_a = {}
            for via in ('shm', 'rccl'):
                env = dict(os.environ, LTMI_BENCH_EXTRAS='0', MASTER_ADDR='127.0.0.1',
                           MASTER_PORT=str(29600 + (os.getpid() + len(out_a)) % 300), RANK='0', WORLD_SIZE='1',
                           LOCAL_RANK=str(local_rank))
                env.pop('LTMI_RESULT_VIA', None)
                cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(max(5, args.steps // 2)),
                       '--warmup', '2', '--no-extras', '--no-cpu-baseline', '--result-via', via]
                try:
                    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                       timeout=240)
                    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
                    if r.returncode != 0 or not line:
                        out_a[via] = {"error": (r.stderr or r.stdout)[-300:]}
                        continue
                    j = json.loads(line[-1])
                    out_a[via] = {"value": j['value'], "unit": j['unit'], "ms_per_step": j['ms_per_step'],
                                  "steps": j['steps'], "result_via": j.get('result_via'),
                                  "kernel_avg_launch_ms": j['roofline'].get('avg_launch_ms')}
                except Exception as e:                    # noqa: BLE001
                    out_a[via] = {"error": repr(e)[:300]}
            out_a["note"] = ("world of one, collectives forced (bench.py --gpus 1 --result-via shm|rccl): the N = 1 point "
                             "of each delivery path; no multi-GPU hardware figure exists yet")
            return out_a
        guarded('delivery_anchor_n1', delivery_anchor)

    watchdog.cancel()
    emit()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
