"""
Slice / Shape algebra of the host runtime (libertem_amd/common/{slice,shape}.py) checked through
properties against plain NumPy indexing on random geometry -- the behaviours the reference pins
case by case in tests/common/test_slice.py and tests/common/test_shape.py (sub-slicing with ragged
edges and origins, intersections, shifts, nav flattening, sig-only / nav-only access, ROI adjustment,
error cases, shape arithmetic).
"""
import itertools
import pickle

import numpy as np
import pytest

from libertem_amd.common import Shape, Slice
from libertem_amd.common.slice import SliceUsageError


def _rand_slice(rng, dims, sig_dims, lim=7):
    origin = tuple(int(x) for x in rng.integers(0, lim, dims))
    shape = tuple(int(x) for x in rng.integers(1, lim, dims))
    return Slice(origin=origin, shape=Shape(shape, sig_dims=sig_dims))


def _mask(sl, extent):
    m = np.zeros(extent, dtype=bool)
    m[sl.get()] = True
    return m


@pytest.mark.parametrize('seed', range(12))
def test_subslices_tile_the_parent_exactly_once(seed):
    rng = np.random.default_rng(seed)
    dims = int(rng.integers(2, 5))
    sig_dims = int(rng.integers(1, dims))
    parent = _rand_slice(rng, dims, sig_dims)
    sub = tuple(int(rng.integers(1, s + 2)) for s in parent.shape)      # may exceed the parent
    extent = tuple(o + s for o, s in zip(parent.origin, parent.shape))
    seen = np.zeros(extent, dtype=np.int32)
    subs = list(parent.subslices(shape=sub))
    for s in subs:
        assert s.shape.sig_dims == sig_dims
        assert all(a <= b for a, b in zip(tuple(s.shape), sub))        # ragged edges only shrink
        seen[s.get()] += 1
    assert np.array_equal(seen == 1, _mask(parent, extent)) and seen.max() == 1
    # emitted in C (ndindex) order of their origins
    assert [s.origin for s in subs] == sorted(s.origin for s in subs)
    n_expected = int(np.prod([-(-p // q) for p, q in zip(tuple(parent.shape), sub)]))
    assert len(subs) == n_expected


@pytest.mark.parametrize('seed', range(12))
def test_intersection_and_shift_match_numpy(seed):
    rng = np.random.default_rng(100 + seed)
    dims = int(rng.integers(2, 5))
    sig_dims = int(rng.integers(1, dims))
    a, b = _rand_slice(rng, dims, sig_dims), _rand_slice(rng, dims, sig_dims)
    extent = tuple(max(x.origin[d] + x.shape[d] for x in (a, b)) for d in range(dims))
    both = _mask(a, extent) & _mask(b, extent)
    inter = a.intersection_with(b)
    assert inter == b.intersection_with(a) or (inter.is_null() and b.intersection_with(a).is_null())
    if both.any():
        assert not inter.is_null() and np.array_equal(_mask(inter, extent), both)
        # the intersection, seen from a's own origin, indexes a's data
        data = rng.random(extent)
        local = inter.shift(a)
        assert np.array_equal(data[a.get()][local.get()], data[inter.get()])
        assert local.origin == tuple(i - o for i, o in zip(inter.origin, a.origin))
    else:
        assert inter.is_null()
    # shifting by a sig offset moves only the sig origin; shift() and shift_by() agree
    off = tuple(int(x) for x in rng.integers(-3, 4, sig_dims))
    moved = a.shift_by(off)
    assert moved.origin[:dims - sig_dims] == a.origin[:dims - sig_dims]
    assert moved.origin[dims - sig_dims:] == tuple(o + d for o, d in zip(a.origin[dims - sig_dims:], off))
    assert tuple(moved.shape) == tuple(a.shape)
    full_off = tuple(int(x) for x in rng.integers(-2, 3, dims))       # full-dimensional offsets too
    assert a.shift_by(full_off).origin == tuple(o + d for o, d in zip(a.origin, full_off))
    with pytest.raises(SliceUsageError):
        a.shift_by(full_off + (1,))


def test_sig_only_nav_only_and_array_access():
    rng = np.random.default_rng(7)
    arr = rng.random((6, 5, 9, 8))
    s = Slice(origin=(2, 1, 3, 2), shape=Shape((3, 2, 4, 5), sig_dims=2))
    assert s.get() == (slice(2, 5), slice(1, 3), slice(3, 7), slice(2, 7))
    assert s.get(sig_only=True) == (slice(3, 7), slice(2, 7))
    assert s.get(nav_only=True) == (slice(2, 5), slice(1, 3))
    assert np.array_equal(s.get(arr), arr[2:5, 1:3, 3:7, 2:7])
    assert np.array_equal(s.get(arr, sig_only=True), arr[..., 3:7, 2:7])
    assert np.array_equal(s.get(arr[:, :, 0, 0], nav_only=True), arr[2:5, 1:3, 0, 0])
    with pytest.raises(SliceUsageError):
        s.get(sig_only=True, nav_only=True)
    assert s.sig.origin == (3, 2) and tuple(s.sig.shape) == (4, 5)
    assert s.nav.origin == (2, 1) and tuple(s.nav.shape) == (3, 2)
    d = s.discard_nav()
    assert d.origin[-2:] == (3, 2) and tuple(d.shape)[-2:] == (4, 5) and d.shape.sig_dims == 2
    full = Slice.from_shape((6, 5, 9, 8), sig_dims=2)
    assert full.origin == (0, 0, 0, 0) and tuple(full.shape) == (6, 5, 9, 8)
    assert pickle.loads(pickle.dumps(s)) == s and hash(pickle.loads(pickle.dumps(s))) == hash(s)


@pytest.mark.parametrize('nav', [(7,), (3, 4), (2, 3, 2)])
def test_flatten_nav_and_roi_adjustment(nav):
    rng = np.random.default_rng(len(nav))
    sig = (3, 2)
    n = int(np.prod(nav))
    # a slice of whole nav rows flattens to a contiguous frame range
    first = tuple([1] + [0] * (len(nav) - 1))
    span = tuple([max(1, nav[0] - 1)] + list(nav[1:]))
    s = Slice(origin=first + (0, 0), shape=Shape(span + sig, sig_dims=2))
    flat = s.flatten_nav(nav + sig)
    per_row = n // nav[0]
    assert flat.origin == (per_row, 0, 0) and tuple(flat.shape) == (span[0] * per_row,) + sig
    # ROI adjustment: origin / length count the selected frames before / inside the slice
    roi = rng.random(n) < 0.5
    part = Slice(origin=(2, 0, 0), shape=Shape((max(1, n - 3),) + sig, sig_dims=2))
    adj = part.adjust_for_roi(roi)
    assert adj.origin[0] == int(np.count_nonzero(roi[:2]))
    assert adj.shape[0] == int(np.count_nonzero(roi[2:2 + part.shape[0]]))
    assert tuple(adj.shape)[1:] == sig and adj.origin[1:] == (0, 0)
    assert part.adjust_for_roi(None) == part
    if len(nav) > 1:
        with pytest.raises(SliceUsageError):
            s.adjust_for_roi(roi)


def test_slice_error_cases():
    with pytest.raises(SliceUsageError):
        Slice(origin=(0, 0), shape=(2, 2))                     # shape must be a Shape
    with pytest.raises(SliceUsageError):
        Slice(origin=(0, 0, 0), shape=Shape((2, 2), sig_dims=1))
    a = Slice(origin=(0, 0, 0), shape=Shape((2, 2, 2), sig_dims=2))
    b = Slice(origin=(0, 0, 0), shape=Shape((2, 2, 2), sig_dims=1))
    with pytest.raises(SliceUsageError):
        a.intersection_with(b)
    with pytest.raises(SliceUsageError):
        list(a.subslices(shape=(1, 1)))
    assert a != b and a == Slice(origin=(0, 0, 0), shape=Shape((2, 2, 2), sig_dims=2))


def test_shape_algebra():
    for dims, sig_dims in itertools.product((2, 3, 4, 5), (1, 2, 3)):
        if sig_dims >= dims + 1:
            continue
        t = tuple(range(2, 2 + dims))
        s = Shape(t, sig_dims=min(sig_dims, dims))
        k = dims - s.sig_dims
        assert tuple(s.nav) == t[:k] and tuple(s.sig) == t[k:]
        # (a shape without dimensions has size 0: reference tests/common/test_shape.py test_size_nav_zero)
        assert s.size == int(np.prod(t)) and s.nav.size == (int(np.prod(t[:k], dtype=np.int64)) if k else 0)
        assert s.sig.size == int(np.prod(t[k:]))
        assert s.dims == dims and s.nav.dims == k and s.sig.dims == s.sig_dims
        if k:
            assert tuple(s.flatten_nav()) == (int(np.prod(t[:k], dtype=np.int64)),) + t[k:]
        assert tuple(s.flatten_sig()) == t[:k] + (int(np.prod(t[k:])),)
        assert s.to_tuple() == t and tuple(s) == t and len(s) == dims and s[0] == t[0] and s[-1] == t[-1]
        assert s == Shape(t, sig_dims=s.sig_dims) and hash(s) == hash(Shape(t, sig_dims=s.sig_dims))
        if s.sig_dims != 1 and dims > 1:
            assert s != Shape(t, sig_dims=1)
        # shape + tuple: more sig dimensions; tuple + shape: more nav dimensions, behind the existing ones
        # (reference tests/common/test_shape.py test_shape_add_1 / _2)
        right, left = s + (9, 8), (9, 8) + s
        assert isinstance(right, Shape) and tuple(right) == t + (9, 8) and right.sig.dims == s.sig_dims + 2
        assert isinstance(left, Shape) and tuple(left) == t[:k] + (9, 8) + t[k:] and left.sig.dims == s.sig_dims
        assert repr(s) == repr(t)
        assert pickle.loads(pickle.dumps(s)) == s
        assert str(t[0]) in repr(s)
    empty = Shape((0, 4, 4), sig_dims=2)
    assert empty.size == 0 and empty.nav.size == 0 and empty.sig.size == 16
    assert Shape((), sig_dims=0).size == 0 and Shape((128, 128), sig_dims=2).nav.size == 0


def test_slice_clip_to():
    """Slice.clip_to (common/slice.py:397-399): the part inside an array of a shape anchored at the origin"""
    from libertem_amd.common.slice import Slice
    from libertem_amd.common.shape import Shape
    s = Slice(origin=(2, 3, 4), shape=Shape((4, 10, 10), sig_dims=2))
    c = s.clip_to(Shape((4, 8, 8), sig_dims=2))
    assert tuple(c.origin) == (2, 3, 4) and tuple(c.shape) == (2, 5, 4)
    inside = Slice(origin=(0, 1, 1), shape=Shape((2, 3, 3), sig_dims=2))
    assert inside.clip_to(Shape((4, 8, 8), sig_dims=2)) == inside
    assert Slice(origin=(5, 0, 0), shape=Shape((2, 3, 3), sig_dims=2)).clip_to(Shape((4, 8, 8), sig_dims=2)).is_null()


def test_scale_rotate_flip_y_round_trip():
    """corrections/coordinates.py:57-93: (scale, angle, flip) of scale() @ rotate() @ flip_y(); shear and unequal
    scales are refused"""
    import pytest
    from libertem_amd.corrections import coordinates as c
    for sc, ang, fl in [(2.5, 0.3, False), (1., -2.1, True), (0.7, 3.0, True), (4., 0., False), (1.5, np.pi / 2, True)]:
        m = c.scale(sc) @ c.rotate(ang) @ (c.flip_y() if fl else c.identity())
        s2, a2, f2 = c.scale_rotate_flip_y(m)
        assert np.isclose(s2, sc) and f2 is fl
        assert np.allclose((np.sin(a2), np.cos(a2)), (np.sin(ang), np.cos(ang)))
        back = c.scale(s2) @ c.rotate(a2) @ (c.flip_y() if f2 else c.identity())
        assert np.allclose(back, m)
    with pytest.raises(ValueError, match='are different'):
        c.scale_rotate_flip_y(np.array([(2., 0.), (0., 1.)]))
    with pytest.raises(ValueError, match='shear'):
        c.scale_rotate_flip_y(np.array([(1., 1.), (0., 1.)]) / np.array([1., np.sqrt(2.)]))


def test_utils_polar_and_rotations():
    """libertem.utils (utils/__init__.py:9-132; reference tests/test_utils.py, tests/corrections/test_coordinates.py):
    polar <-> cartesian round trip, rotations consistent with corrections.coordinates.rotate"""
    from libertem_amd.utils import make_polar, make_cartesian, rotate_deg, rotate_rad
    from libertem_amd.corrections import coordinates as c
    rng = np.random.default_rng(1)
    cart = rng.random((7, 2)) * 10 - 5
    pol = make_polar(cart)
    assert pol.shape == (7, 2) and np.allclose(pol[:, 0], np.hypot(cart[:, 0], cart[:, 1]))
    assert np.allclose(pol[:, 1], np.arctan2(cart[:, 0], cart[:, 1])) and np.allclose(make_cartesian(pol), cart)
    grid = rng.random((3, 4, 2))
    assert np.allclose(make_cartesian(make_polar(grid)), grid)
    y, x = rng.random((2, 7))
    r_y, r_x = c.identity() @ c.rotate(np.pi / 180 * 23) @ (y, x)
    assert np.allclose(rotate_deg(y, x, 23), (r_y, r_x)) and np.allclose(rotate_rad(y, x, np.pi / 180 * 23), (r_y, r_x))
    assert np.allclose(rotate_deg(1., 0., 90), (0., -1.))        # (y down: clockwise)
