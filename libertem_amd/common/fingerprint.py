"""
Content fingerprints of UDF parameters.

The reference evaluates the mask factories on every run and on every task
(src/libertem/udf/masks.py:331-351, common/container.py:260-314), so an array that a factory closes
over -- or an attribute of the object whose method the factory is -- may be modified in place between two
`run_udf` calls and the next run sees the new values.  This implementation keeps evaluated stacks (device
images) and whole run plans across runs; what identifies "the same parameters" therefore has to look INTO
the objects: the identity of a factory plus a fingerprint of everything it can see -- closure cells,
defaults, functools.partial arguments, the module globals it names, the `__dict__` of the object a bound
method belongs to and of captured objects.  A fingerprint hashes EVERY byte of every array it reaches
(xxh3: ~0.3 ms for the 4 MiB C2 stack, ~0.12 ms per MiB; compared behind the enqueued kernels --
udf/base.py `_prepare_run_for_dataset`): any in-place edit is seen.  (Up to round 4 buffers above 64 MiB were
sampled; a single changed element could go unseen.)

What cannot be seen completely is NOT cached: only known-transparent kinds are looked into -- NumPy arrays,
scipy.sparse matrices, numbers / strings, list / tuple / set / dict (keys and values), functions, bound methods,
partials, and instances of classes DEFINED IN PYTHON (their `__dict__` / `__slots__`).  Everything else -- device
tensors (torch.Tensor, HipArray: their bytes live in HBM), file objects, random generators (random.Random,
numpy.random.Generator / RandomState), memoryviews, any instance of a type implemented in C whose state no
`__dict__` shows -- and anything nested deeper than MAX_DEPTH gives an OPAQUE token that never compares equal, so
the stack is evaluated and the run planned afresh every time, like in the reference.  Files a factory reads and global random state stay
invisible to any fingerprint: `ApplyMasksUDF(..., cache=False)` or `Context.invalidate_caches()`.
"""
import functools
import io
import itertools
import random
import types
import warnings

import numpy as np

try:
    import xxhash

    def _hash(b):
        return xxhash.xxh3_64_intdigest(b)
except Exception:                                            # pragma: no cover
    import hashlib
    warnings.warn("xxhash is not installed: parameter fingerprints fall back to blake2b (~10x slower per byte)")

    def _hash(b):
        return int.from_bytes(hashlib.blake2b(b, digest_size=8).digest(), 'little')

MAX_DEPTH = 4
_OPAQUE_IDS = itertools.count(1)
_STABLE_TYPES = (types.ModuleType, type, types.BuiltinFunctionType, types.BuiltinMethodType, np.ufunc,
                 types.MethodDescriptorType, types.WrapperDescriptorType)


def array_fingerprint(a):
    """(shape, dtype, hash of every byte)"""
    a = np.asarray(a)
    head = (a.shape, a.dtype.str)
    if a.dtype.hasobject:
        return head + (('opaque', next(_OPAQUE_IDS)),)
    if a.nbytes == 0:
        return head + (0,)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    return head + (_hash(a.reshape(-1).view(np.uint8).data),)


def _sparse_parts(obj):
    """arrays of a scipy.sparse matrix (or anything that looks like one)"""
    parts = []
    for name in ('data', 'indices', 'indptr', 'row', 'col', 'coords'):
        v = getattr(obj, name, None)
        if isinstance(v, np.ndarray):
            parts.append(v)
    return parts


def _opaque(obj):
    return ('opaque', id(obj), next(_OPAQUE_IDS))


_HEAPTYPE = 1 << 9                                          # Py_TPFLAGS_HEAPTYPE: a class made by a `class` statement


def _known_opaque(obj):
    """kinds whose state a `__dict__` does not show (or that live on the device): never looked into"""
    if isinstance(obj, (io.IOBase, random.Random, memoryview, np.random.Generator, np.random.RandomState,
                        np.random.BitGenerator, types.GeneratorType)):
        return True
    mod = type(obj).__module__ or ''
    if mod == 'torch' or mod.startswith('torch.'):
        return True                                         # tensors, generators, streams, storages
    try:
        from libertem_amd.common.hiparray import HipArray
        if isinstance(obj, HipArray):
            return True
    except Exception:                                       # pragma: no cover
        pass
    # an instance of a type implemented in C (no `class` statement made it): whatever `__dict__` it has is not its state
    return not (type(obj).__flags__ & _HEAPTYPE)


def is_opaque(fp):
    """True iff the fingerprint contains something that could not be looked into (never equal to any other)"""
    if isinstance(fp, tuple):
        if len(fp) >= 1 and fp[0] == 'opaque':
            return True
        return any(is_opaque(x) for x in fp)
    return False


def fingerprint(obj, _depth=0, _path=()):
    """hashable value that changes when `obj`, or anything `obj` can reach, changes: arrays by content, numbers
    and strings by value wherever they sit (a counter a factory keeps in a captured dict changes with every
    evaluation -- such a factory is evaluated afresh every run, like in the reference), containers, functions,
    bound methods and instances by what they hold."""
    if isinstance(obj, np.ndarray):
        return ('nd', id(obj)) + array_fingerprint(obj)
    if obj is None or isinstance(obj, (bool, int, float, complex, str, bytes, np.generic)):
        return ('v', obj)
    if isinstance(obj, _STABLE_TYPES):
        return ('stable', id(obj))
    if isinstance(obj, np.dtype):
        return ('v', obj.str)
    if isinstance(obj, (range, slice)) or obj is Ellipsis:
        return ('v', repr(obj))
    if id(obj) in _path:
        return ('cycle', id(obj))                           # (an object that holds the factory that holds it)
    if _is_dataset(obj):
        return ('ds', id(obj))                              # the frames are the data, not a mask parameter
    if _depth > MAX_DEPTH:
        return _opaque(obj)
    path = _path + (id(obj),)
    if isinstance(obj, (list, tuple)):
        return ('seq', id(obj), len(obj)) + tuple(fingerprint(x, _depth + 1, path) for x in obj)
    if isinstance(obj, (set, frozenset)):
        # contents, in an order that does not depend on the set's history (ids drop out of the sort key)
        members = [fingerprint(x, _depth + 1, path) for x in obj]
        if any(is_opaque(m) for m in members):
            return _opaque(obj)
        return ('set', id(obj), len(obj)) + tuple(sorted(members, key=repr))
    if isinstance(obj, (bytearray,)):
        return ('v', bytes(obj))
    if isinstance(obj, dict):
        return ('map', id(obj), len(obj)) + tuple(
            (fingerprint(k, _depth + 1, path), fingerprint(v, _depth + 1, path)) for k, v in obj.items())
    if isinstance(obj, functools.partial):
        return ('partial', id(obj), fingerprint(obj.func, _depth + 1, path),
                fingerprint(obj.args, _depth + 1, path), fingerprint(obj.keywords, _depth + 1, path))
    sp_parts = _sparse_parts(obj) if hasattr(obj, 'shape') and hasattr(obj, 'dtype') else None
    if sp_parts:
        return ('sp', id(obj)) + tuple(array_fingerprint(p) for p in sp_parts)
    if callable(obj) and (hasattr(obj, '__code__') or hasattr(obj, '__func__')):
        # (a bound method object is made afresh by every attribute access `holder.make`: its own id() says nothing --
        #  the function's and, below, the object's do)
        fn = getattr(obj, '__func__', obj)
        out = ['fn', id(fn)]
        for cell in (getattr(fn, '__closure__', None) or ()):
            try:
                out.append(fingerprint(cell.cell_contents, _depth + 1, path))
            except ValueError:                                  # empty cell
                out.append(None)
        for d in (getattr(fn, '__defaults__', None) or ()):
            out.append(fingerprint(d, _depth + 1, path))
        for k, d in (getattr(fn, '__kwdefaults__', None) or {}).items():
            out.append((k, fingerprint(d, _depth + 1, path)))
        code, glob = getattr(fn, '__code__', None), getattr(fn, '__globals__', None)
        if code is not None and glob is not None:
            for name in code.co_names:
                if name not in glob:                        # (an attribute name or a builtin)
                    continue
                v = glob[name]
                # module globals the function names: modules, classes and functions are taken as fixed; every
                # other object counts -- by value / content, or OPAQUE when it cannot be looked into
                # (`rng = np.random.default_rng()` at module level)
                if isinstance(v, _STABLE_TYPES) or callable(v):
                    continue
                out.append((name, fingerprint(v, _depth + 1, path)))
        bound = getattr(obj, '__self__', None)
        if bound is not None and not isinstance(bound, _STABLE_TYPES):
            # a bound method: what the method can read of its object
            out.append(('self', fingerprint(bound, _depth + 1, path)))
        return tuple(out)
    if _known_opaque(obj):
        return _opaque(obj)
    d = getattr(obj, '__dict__', None)
    slots = [n for klass in type(obj).__mro__ for n in getattr(klass, '__slots__', ()) if isinstance(n, str)]
    if isinstance(d, dict) or slots:
        # an instance: its attributes (scalars by value: `holder.radius = 3`)
        attrs = list((d or {}).items()) + [(n, getattr(obj, n)) for n in slots
                                           if n not in ('__dict__', '__weakref__') and hasattr(obj, n)]
        return ('obj', id(obj), type(obj).__qualname__, len(attrs)) + tuple(
            (k, fingerprint(v, _depth + 1, path)) for k, v in attrs)
    return _opaque(obj)


def _is_dataset(obj):
    try:
        from libertem_amd.io.dataset.base import DataSet
    except Exception:                                        # pragma: no cover
        return False
    return isinstance(obj, DataSet)
