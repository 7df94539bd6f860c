"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): NumPy restatement of the reference's byte-order
decoders, src/libertem/io/dataset/base/decode.py.  Pinned against tests/golden/decode.npz (outputs of
the reference's own functions on seeded inputs, tests/golden/generate_golden.py:gen_decode) for the
unsigned dtypes the reference tests (tests/io/test_decode_swap.py:166-245).

The reference composes every item as an UNSIGNED word from the file's bytes (most significant first)
and stores it into the read dtype (decode.py:15-20, 32-39, 55-66); for signed outputs it relies on
numba's wrap-around.  `decode_swap` below does the same with explicit wrap-around.
"""
import sys

import numpy as np


def _order(dt):
    """decode.py:103-109"""
    o = np.dtype(dt).byteorder
    if o == '|':
        return '|'
    if o != '=':
        return o
    return {'little': '<', 'big': '>'}[sys.byteorder]


def need_byteswap(native_dtype, read_dtype):
    """DtypeConversionDecoder._need_byteswap (decode.py:124-129)"""
    native_dtype, read_dtype = np.dtype(native_dtype), np.dtype(read_dtype)
    nd, rd = _order(native_dtype), _order(read_dtype)
    if nd == '|':                                   # single bytes have no order
        return False
    return nd != rd and native_dtype.itemsize > 1


def get_native_dtype(inp_native_dtype, read_dtype):
    """decode.py:160-163: raw bytes are handed to the swapping decoders"""
    if need_byteswap(inp_native_dtype, read_dtype):
        return np.dtype(np.uint8)
    return np.dtype(inp_native_dtype)


def byteswap_straight(inp_bytes, itemsize):
    """byteswap_{2,4,8}_straight (decode.py:8-12, 23-29, 42-52): bytes of every item reversed."""
    b = np.asarray(inp_bytes, dtype=np.uint8)
    n = b.size // itemsize
    return b[:n * itemsize].reshape(n, itemsize)[:, ::-1].reshape(-1).copy()


def decode_swap(inp_bytes, itemsize, out_dtype):
    """byteswap_{2,4,8}_decode (decode.py:15-20, 32-39, 55-66): out[i] = unsigned big-endian word i,
    stored into `out_dtype` (floats: value conversion; narrower / signed ints: wrap-around)."""
    b = np.asarray(inp_bytes, dtype=np.uint8)
    n = b.size // itemsize
    words = np.zeros(n, dtype=np.uint64)
    for k in range(itemsize):
        words |= b[k:n * itemsize:itemsize].astype(np.uint64) << np.uint64(8 * (itemsize - 1 - k))
    out_dtype = np.dtype(out_dtype)
    if out_dtype.kind in 'iu':
        return words.astype(np.dtype(f'u{out_dtype.itemsize}')).view(out_dtype) \
            if out_dtype.itemsize < 8 else words.view(out_dtype)
    return words.astype(out_dtype)


def decode(inp, native_dtype, read_dtype):
    """DtypeConversionDecoder.get_decode (decode.py:145-158) applied to one flat buffer: `inp` holds
    the items in the byte order of `native_dtype`; the result is a `read_dtype` array."""
    native_dtype, read_dtype = np.dtype(native_dtype), np.dtype(read_dtype)
    if not need_byteswap(native_dtype, read_dtype):
        # default_decode (decode.py:69-71): out[idx, :] = inp.view(native_dtype)
        return np.asarray(inp).view(native_dtype.newbyteorder('=')).astype(read_dtype)
    if native_dtype.kind in ('f', 'c'):
        raise NotImplementedError("byte swapping for floats not implemented yet")   # decode.py:151-156
    return decode_swap(np.asarray(inp).view(np.uint8), native_dtype.itemsize, read_dtype)
