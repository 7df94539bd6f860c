"""
Minimal n-d sparse mask stack.

The reference represents sparse mask stacks with pydata `sparse.COO` (common/container.py:260-314,
masks.py:290-353); that third-party package is not part of this build, so sparse stacks are held in
`SparseStack`: a COO triplet list over (mask index, flat sig pixel).  It offers exactly what the
mask container needs: concatenation, dtype cast, sig-slicing into the CSR/CSC matrices of
`_build_sparse` (common/container.py:33-71) and densification.
"""
import numpy as np
import scipy.sparse as sp

from .math import prod


class SparseStack:
    def __init__(self, data, mask_idx, px_idx, n_masks, sig_shape):
        self.data = np.asarray(data)
        self.mask_idx = np.asarray(mask_idx, dtype=np.int64)
        self.px_idx = np.asarray(px_idx, dtype=np.int64)
        self.n_masks = int(n_masks)
        self.sig_shape = tuple(int(s) for s in sig_shape)
        assert self.data.shape == self.mask_idx.shape == self.px_idx.shape

    # --- constructors -------------------------------------------------------------------------
    @classmethod
    def from_scipy(cls, m, sig_shape=None):
        """One 2D scipy.sparse matrix = one sig-shaped mask."""
        m = sp.coo_matrix(m)
        sig_shape = m.shape if sig_shape is None else tuple(sig_shape)
        px = np.ravel_multi_index((m.row, m.col), m.shape)
        return cls(m.data, np.zeros(len(px), dtype=np.int64), px, 1, sig_shape)

    @classmethod
    def from_dense(cls, stack):
        stack = np.asarray(stack)
        n = stack.shape[0]
        flat = stack.reshape((n, -1))
        mi, pi = np.nonzero(flat)
        return cls(flat[mi, pi], mi, pi, n, stack.shape[1:])

    @classmethod
    def from_csr_masks_by_px(cls, csr, sig_shape):
        """scipy matrix of shape (n_masks, n_px)."""
        coo = sp.coo_matrix(csr)
        return cls(coo.data, coo.row, coo.col, coo.shape[0], sig_shape)

    @classmethod
    def concatenate(cls, stacks):
        stacks = list(stacks)
        sig = stacks[0].sig_shape
        for s in stacks:
            if s.sig_shape != sig:
                raise ValueError("all masks need the same sig shape")
        offs = np.cumsum([0] + [s.n_masks for s in stacks])
        dtype = np.result_type(*[s.data.dtype for s in stacks])
        return cls(
            np.concatenate([s.data.astype(dtype) for s in stacks]),
            np.concatenate([s.mask_idx + o for s, o in zip(stacks, offs[:-1])]),
            np.concatenate([s.px_idx for s in stacks]),
            int(offs[-1]), sig)

    # --- array-ish API -------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.n_masks,) + self.sig_shape

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def nnz(self):
        return len(self.data)

    def __len__(self):
        return self.n_masks

    def __getitem__(self, k):
        """`stack[i]`: mask i as a sig-shaped stack of one -- `.todense()` gives the 2-D array, like indexing the
        reference's `sparse.COO` stacks (tests/test_masks.py:39-60); a slice / index array: a sub-stack"""
        if isinstance(k, (int, np.integer)):
            i = int(k) + (self.n_masks if k < 0 else 0)
            if not 0 <= i < self.n_masks:
                raise IndexError(f"mask {k} of a stack of {self.n_masks}")
            sel = self.mask_idx == i
            return _SparseMask(self.data[sel], self.px_idx[sel], self.sig_shape)
        idx = np.arange(self.n_masks)[k]
        lut = np.full(self.n_masks, -1, dtype=np.int64)
        lut[idx] = np.arange(len(idx))
        sel = lut[self.mask_idx] >= 0
        return SparseStack(self.data[sel], lut[self.mask_idx[sel]], self.px_idx[sel], len(idx), self.sig_shape)

    def sum(self, axis=None):
        """axis=None: the sum of all entries; axis=0: over the masks (a sig-shaped sparse result with `.todense()`);
        the sig axes (1 ..): per mask, dense"""
        if axis is None:
            return self.data.sum()
        axes = (axis,) if isinstance(axis, (int, np.integer)) else tuple(axis)
        axes = tuple(sorted(a + (len(self.shape) if a < 0 else 0) for a in axes))
        if axes == (0,):
            m = sp.coo_matrix((self.data, (np.zeros(self.nnz, dtype=np.int64), self.px_idx)),
                              shape=(1, prod(self.sig_shape))).tocsr()
            m = m.tocoo()
            return _SparseMask(m.data, m.col.astype(np.int64), self.sig_shape)
        if axes == tuple(range(1, len(self.shape))):
            out = np.zeros(self.n_masks, dtype=self.data.dtype)
            np.add.at(out, self.mask_idx, self.data)
            return out
        return self.todense().sum(axis=axis)

    def astype(self, dtype):
        return SparseStack(self.data.astype(dtype), self.mask_idx, self.px_idx, self.n_masks,
                           self.sig_shape)

    def todense(self):
        out = np.zeros((self.n_masks, prod(self.sig_shape)), dtype=self.data.dtype)
        np.add.at(out, (self.mask_idx, self.px_idx), self.data)
        return out.reshape(self.shape)

    def to_px_by_masks(self, sig_slice=None, dtype=None, fmt='csr'):
        """
        The (px_in_slice, n_masks) matrix of the reference's `_build_sparse`: pixels of the
        (sig-)slice in C order are the rows.  Canonical format (sorted indices, duplicates summed).
        """
        data, mi, pi = self.data, self.mask_idx, self.px_idx
        if dtype is not None:
            data = data.astype(dtype)
        n_px = prod(self.sig_shape)
        if sig_slice is not None:
            sl = sig_slice.get(sig_only=True)
            lut = np.full(n_px, -1, dtype=np.int64)
            idx = np.arange(n_px).reshape(self.sig_shape)[sl].reshape(-1)
            lut[idx] = np.arange(len(idx))
            local = lut[pi]
            keep = local >= 0
            data, mi, pi = data[keep], mi[keep], local[keep]
            n_px = len(idx)
        cls = sp.csc_matrix if fmt == 'csc' else sp.csr_matrix
        m = cls((data, (pi, mi)), shape=(n_px, self.n_masks))
        m.sum_duplicates()
        m.sort_indices()
        return m


def is_sparse(a):
    return isinstance(a, (SparseStack, _SparseMask)) or sp.issparse(a)


def to_dense(a):
    if isinstance(a, (SparseStack, _SparseMask)):
        return a.todense()
    if sp.issparse(a):
        return a.toarray()
    return np.array(a)


def to_sparse(a, shape=None):
    """The sparse form of a mask (reference common/sparse.py:20-32, there a pydata `sparse.COO`): a 2-D array becomes a
    scipy CSR matrix, a stack of masks a SparseStack; sparse inputs pass through.  A list of roi coordinates
    ((y, x), ...) or ((y, x, value), ...) with `shape`: a bool scipy matrix / array with those positions set."""
    if isinstance(a, (tuple, list)):
        if all(isinstance(aa, (int, np.integer)) for aa in a):
            a = (tuple(a) + (True,),)
        values = {bool(aa[-1]) if len(aa) == len(tuple(shape)) + 1 else True for aa in a}
        if len(values) != 1:
            raise ValueError(f'Cannot cast iterable roi coords with more than one truth value {values}')
        val = values.pop()
        out = np.full(tuple(shape), not val, dtype=bool)
        for aa in a:
            out[tuple(aa[:len(tuple(shape))])] = val
        return sp.csr_matrix(out) if out.ndim == 2 else out
    if is_sparse(a) or isinstance(a, _SparseMask):
        return a
    a = np.asarray(a)
    return sp.csr_matrix(a) if a.ndim == 2 else SparseStack.from_dense(a)


def to_sparse_stack(a):
    """Anything mask-like with a leading stack axis -> SparseStack."""
    if isinstance(a, SparseStack):
        return a
    return SparseStack.from_dense(np.asarray(a))


class _SparseMask:
    """one sig-shaped sparse array (what `stack[i]` and `stack.sum(axis=0)` give)"""

    def __init__(self, data, px_idx, sig_shape):
        self.data, self.px_idx, self.shape = np.asarray(data), np.asarray(px_idx, dtype=np.int64), tuple(sig_shape)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def nnz(self):
        return len(self.data)

    def todense(self):
        out = np.zeros(prod(self.shape), dtype=self.data.dtype)
        np.add.at(out, self.px_idx, self.data)
        return out.reshape(self.shape)

    def sum(self):
        return self.data.sum()

    def __array__(self, dtype=None, copy=None):
        a = self.todense()
        return a if dtype is None else a.astype(dtype)
