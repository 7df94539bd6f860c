"""
Oracle (TEST INFRASTRUCTURE) for detector corrections: scalar-loop restatement of
  io/corrections/detector.py:17-101   (_correct_numba_inplace: (x - dark) * gain, then patch)
  io/corrections/detector.py:104-169  (environments + flatten_filter)
  io/corrections/corrset.py:13-67, 205-260 (disjunct_multiplier, adjust)
Pinned by tests/golden/corrections.npz (generated from the real reference).  Written with plain
loops on purpose: it shares no code with libertem_amd.io.corrections, which it checks.
"""
import numpy as np


def repair_tables(sig_shape, coords):
    """coords: list of index tuples.  -> (exclude_flat, repair_flat (k, 3^d - 1), counts)."""
    sig_shape = tuple(int(s) for s in sig_shape)
    nd = len(sig_shape)
    coords = [tuple(int(x) for x in c) for c in coords]
    bad = set(coords)
    n_off = 3 ** nd - 1
    exclude_flat = np.zeros(len(coords), dtype=np.intp)
    repair_flat = np.zeros((len(coords), n_off), dtype=np.intp)
    counts = np.zeros(len(coords), dtype=np.intp)
    for i, c in enumerate(coords):
        exclude_flat[i] = np.ravel_multi_index(c, sig_shape)
        n = 0
        for code in range(3 ** nd):                     # detector.py:121-124: C-order unravel - 1
            off = np.array(np.unravel_index(code, (3,) * nd)) - 1
            if not np.any(off != 0):
                continue
            q = tuple(int(a + b) for a, b in zip(c, off))
            if any(x < 0 or x >= s for x, s in zip(q, sig_shape)):
                continue
            if q in bad:                                # detector.py:155-165: only good pixels
                continue
            repair_flat[i, n] = np.ravel_multi_index(q, sig_shape)
            n += 1
        counts[i] = n
    return exclude_flat, repair_flat, counts


def correct(data, sig_shape, dark=None, gain=None, coords=None, out_dtype=None):
    """Corrected copy of `data` (*nav, *sig) as result_type(float32, data) (detector.py:218)."""
    sig_shape = tuple(sig_shape)
    n_sig = int(np.prod(sig_shape))
    out_dtype = np.result_type(np.float32, data.dtype) if out_dtype is None else out_dtype
    out = data.astype(out_dtype).reshape((-1, n_sig))
    d = None if dark is None else np.asarray(dark).reshape(-1)
    g = None if gain is None else np.asarray(gain).reshape(-1)
    for p in range(n_sig):                              # detector.py:73: promoted arithmetic,
        col = out[:, p]                                 # rounded once on store
        if d is not None:
            col = col - d[p]
        if g is not None:
            col = col * g[p]
        out[:, p] = col
    if coords is not None and len(coords):
        ex, env, cnt = repair_tables(sig_shape, coords)
        for i in range(len(ex)):
            if cnt[i] > 0:                              # detector.py:84-89
                acc = np.zeros(out.shape[0], dtype=np.float64)
                for j in range(cnt[i]):
                    acc += out[:, env[i, j]]
                out[:, ex[i]] = acc / cnt[i]
    return out.reshape(data.shape)


def dot_masks(masks, gain, coords=None):
    """detector.py:315-338 for dense masks (..., *sig)."""
    sig_shape = gain.shape
    n_sig = int(np.prod(sig_shape))
    flat = np.asarray(masks).reshape((-1, n_sig))
    res = flat.copy()
    if coords is not None and len(coords):
        ex, env, cnt = repair_tables(sig_shape, coords)
        for i in range(len(ex)):
            res[:, ex[i]] = 0
            for j in range(cnt[i]):
                res[:, env[i, j]] = res[:, env[i, j]] + flat[:, ex[i]] / cnt[i]
    res = res * np.asarray(gain).reshape(-1)
    return res.reshape(np.asarray(masks).shape)


def disjunct_multiplier(excluded, sig_shape, base_shape=1, target=1):
    """corrset.py:13-67"""
    approx = int(np.round(target / base_shape))
    cur = base_shape * approx
    mx = int(np.max(excluded))
    is_bad = np.zeros(mx + 1, dtype=bool)
    is_bad[np.asarray(excluded)] = True
    sign = 1 if cur >= target else -1
    for offset in range(mx // base_shape + 1):
        cur += offset * sign * base_shape
        sign *= -1
        if cur <= 0:
            continue
        clear = True
        for mult in range(1, mx // cur + 1):
            idx = cur * mult
            if 0 <= idx < sig_shape and idx <= mx and is_bad[idx]:
                clear = False
                break
        if clear:
            return cur
    return min((mx // base_shape + 1) * base_shape, sig_shape)


def adjust_tileshape(tile_shape, sig_shape, base_shape, coords_by_dim):
    """corrset.py:179-260: coords_by_dim[dim] = excluded positions along that dim."""
    if len(coords_by_dim) == 0 or len(coords_by_dim[0]) == 0:
        return tuple(tile_shape)
    adj = [int(x) for x in tile_shape]
    for dim in range(len(adj)):
        if sig_shape[dim] <= 1:
            continue
        uniq = sorted(set(int(x) for x in coords_by_dim[dim]))
        if len(uniq) > sig_shape[dim] / 3:
            adj[dim] = int(sig_shape[dim])
            continue
        stop = sig_shape[dim]
        forb = [u for u in uniq + [u + 1 for u in uniq] if u <= stop]
        nz = [u for u in forb if u != 0]
        m = min(stop, disjunct_multiplier(nz, sig_shape[dim], base_shape[dim], adj[dim]))
        min_size = max(m, 2) if len(nz) != len(forb) else m
        if adj[dim] < min_size or adj[dim] % m != 0:
            adj[dim] = m
    for dim in range(len(adj)):
        if adj[dim] <= 0 or adj[dim] > sig_shape[dim]:
            adj[dim] = int(sig_shape[dim])
    return tuple(int(x) for x in adj)
