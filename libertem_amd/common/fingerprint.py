"""
Cheap content fingerprints of UDF parameters.

The reference evaluates the mask factories on every run and on every task
(src/libertem/udf/masks.py:331-351, common/container.py:260-314), so an array that a factory closes
over may be modified in place between two `run_udf` calls and the next run sees the new values.
This implementation keeps evaluated stacks (device images) and whole run plans across runs; what
identifies "the same parameters" therefore has to look INTO the objects: the identity of a factory
plus a fingerprint of every array it can see (closure cells, defaults, functools.partial arguments,
module globals it names).  A fingerprint hashes EVERY byte of a buffer of up to 64 MiB (xxh3: ~0.3 ms
for the 4 MiB C2 stack, compared behind the enqueued kernels -- udf/base.py `_prepare_run_for_dataset`):
any in-place edit is seen, also a column band (`m[:, :, 100:110] = 0`) or a single element -- an evenly
strided sample is blind to exactly such edits when its stride shares a factor with the row length
(round-3 review).  Larger buffers are sampled: 8192 pieces of 256 bytes at offsets taken from the
golden-ratio sequence (no common period with any row length: a band of b bytes in rows of L bytes is
missed with probability (1 - (b + 256) / L)^8192), plus head and tail; a single changed element of a
buffer above 64 MiB may go unseen (documented contract: DESIGN.md section 3).
"""
import functools

import numpy as np

try:
    import xxhash

    def _hash(b):
        return xxhash.xxh3_64_intdigest(b)
except Exception:                                            # pragma: no cover
    import hashlib

    def _hash(b):
        return int.from_bytes(hashlib.blake2b(b, digest_size=8).digest(), 'little')

FULL_BYTES = 64 * 1024 * 1024
PIECES = 8192
PIECE_BYTES = 256
_GOLDEN = 0.6180339887498949


def array_fingerprint(a):
    a = np.asarray(a)
    head = (a.shape, a.dtype.str)
    if a.dtype.hasobject:
        return head + (id(a),)
    nbytes = a.nbytes
    if nbytes == 0:
        return head + (0,)
    if not a.flags.c_contiguous:
        if nbytes <= FULL_BYTES:
            return head + (_hash(np.ascontiguousarray(a).view(np.uint8).data),)
        # strided sample, at most ~64 Ki elements
        per_axis = max(1, int(round((a.size / 65536.0) ** (1.0 / max(1, a.ndim)))))
        sub = a[tuple(slice(None, None, per_axis) for _ in range(a.ndim))]
        return head + (a.strides, _hash(np.ascontiguousarray(sub).view(np.uint8).data))
    flat = a.reshape(-1).view(np.uint8)
    if nbytes <= FULL_BYTES:
        return head + (_hash(flat.data),)
    # pieces at the golden-ratio sequence of offsets (8-byte aligned, read as uint64 words)
    words = flat[:nbytes - nbytes % 8].view(np.uint64)
    per = PIECE_BYTES // 8
    frac = (np.arange(1, PIECES + 1, dtype=np.float64) * _GOLDEN) % 1.0
    start = (frac * (words.size - per)).astype(np.int64)
    sample = words[start[:, None] + np.arange(per, dtype=np.int64)[None, :]]
    return head + (_hash(sample.data), _hash(flat[:4096].data), _hash(flat[-4096:].data))


def _sparse_parts(obj):
    """arrays of a scipy.sparse matrix (or anything that looks like one)"""
    parts = []
    for name in ('data', 'indices', 'indptr', 'row', 'col', 'coords'):
        v = getattr(obj, name, None)
        if isinstance(v, np.ndarray):
            parts.append(v)
    return parts


def fingerprint(obj, _depth=0, _inside=False):
    """hashable value that changes when `obj`, or an array `obj` can reach, changes.  Plain numbers
    and strings count by value where a factory captures them directly (closure cell, default,
    partial argument), not inside a captured list / dict -- those are followed for the arrays they
    hold; a counter a factory keeps in a dict is not a mask parameter."""
    if isinstance(obj, np.ndarray):
        return ('nd', id(obj)) + array_fingerprint(obj)
    if obj is None or isinstance(obj, (bool, int, float, complex, str, bytes, np.generic)):
        return ('s',) if _inside else ('v', obj)
    if _depth > 3:
        return ('id', id(obj))
    if isinstance(obj, (list, tuple)):
        return ('seq', id(obj), len(obj)) + tuple(fingerprint(x, _depth + 1, True) for x in obj)
    if isinstance(obj, dict):
        return ('map', id(obj), len(obj)) + tuple(fingerprint(v, _depth + 1, True)
                                                  for v in obj.values())
    if isinstance(obj, functools.partial):
        return ('partial', id(obj), fingerprint(obj.func, _depth + 1),
                fingerprint(obj.args, _depth + 1), fingerprint(obj.keywords, _depth + 1))
    sp_parts = _sparse_parts(obj) if hasattr(obj, 'shape') and hasattr(obj, 'dtype') else None
    if sp_parts:
        return ('sp', id(obj)) + tuple(array_fingerprint(p) for p in sp_parts)
    if callable(obj):
        out = ['fn', id(obj)]
        fn = getattr(obj, '__func__', obj)
        for cell in (getattr(fn, '__closure__', None) or ()):
            try:
                out.append(fingerprint(cell.cell_contents, _depth + 1))
            except ValueError:                                  # empty cell
                out.append(None)
        for d in (getattr(fn, '__defaults__', None) or ()):
            out.append(fingerprint(d, _depth + 1))
        for k, d in (getattr(fn, '__kwdefaults__', None) or {}).items():
            out.append((k, fingerprint(d, _depth + 1)))
        code, glob = getattr(fn, '__code__', None), getattr(fn, '__globals__', None)
        if code is not None and glob is not None:
            for name in code.co_names:
                v = glob.get(name)
                if isinstance(v, np.ndarray):
                    out.append((name, fingerprint(v, _depth + 1)))
        bound = getattr(obj, '__self__', None)
        if bound is not None and not isinstance(bound, type(np)):
            out.append(('self', id(bound)))
        return tuple(out)
    return ('id', id(obj))
