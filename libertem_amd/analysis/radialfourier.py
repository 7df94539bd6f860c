"""
Radial Fourier analysis (reference analysis/radialfourier.py): Fourier coefficients of the
intensity on rings, computed -- exactly like the reference, which "doesn't use fast Fourier
transforms" (:172-175) -- as a dot product with `n_bins * (max_order + 1)` complex64 masks
`ring_b * exp(i * o * phi)`.  On MI355X the dense complex stack is the MFMA-f32 kernel with two real
columns per mask; the sparse stack goes to the CSR kernel.
"""
import numpy as np

from libertem_amd import masks
from libertem_amd.common.math import prod
from libertem_amd.common.sparse import SparseStack
from .base import AnalysisResult, AnalysisResultSet
from .masks import BaseMasksAnalysis


class RadialFourierResultSet(AnalysisResultSet):
    pass


def radial_mask_factory(detector_y, detector_x, cx, cy, ri, ro, n_bins, max_order, use_sparse,
                        dtype=np.complex64):
    """Stack factory; mask index = bin * (max_order + 1) + order (analysis/radialfourier.py:106-146)"""
    dtype = np.result_type(dtype, np.complex64)

    def stack():
        rings = masks.radial_bins(centerX=cx, centerY=cy, imageSizeX=detector_x,
                                  imageSizeY=detector_y, radius=ro, radius_inner=ri,
                                  n_bins=n_bins, use_sparse=use_sparse, dtype=dtype)
        orders = np.arange(max_order + 1, dtype=dtype)
        r, phi = masks.polar_map(centerX=cx, centerY=cy, imageSizeX=detector_x,
                                 imageSizeY=detector_y)
        # evaluated in complex64, like the reference (:124-132): exp(phi * order * 1j) element by element -- one order
        # at a time on a few threads (NumPy releases the GIL; the same expression on a slice gives the same bits)
        phi_c = phi.astype(dtype)
        n_orders = max_order + 1

        def modulator_of(o):
            return np.exp(phi_c * orders[o] * 1j)

        from concurrent.futures import ThreadPoolExecutor
        import os
        workers = max(1, min(n_orders, (os.cpu_count() or 2) // 2, 16))
        if use_sparse:
            ring_vals = rings.data.astype(dtype)

            def one(o):
                mod = modulator_of(o).reshape(-1)
                return (ring_vals * mod[rings.px_idx]).astype(dtype)
            with ThreadPoolExecutor(workers) as pool:
                datas = list(pool.map(one, range(n_orders)))
            mis = [rings.mask_idx * n_orders + o for o in range(n_orders)]
            pxs = [rings.px_idx] * n_orders
            return SparseStack(np.concatenate(datas), np.concatenate(mis), np.concatenate(pxs),
                               rings.n_masks * n_orders, (detector_y, detector_x))
        ring_stack = np.empty((rings.shape[0], n_orders) + tuple(rings.shape[1:]), dtype=np.result_type(rings.dtype, dtype))

        def fill(o):
            ring_stack[:, o] = rings * modulator_of(o)
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(fill, range(n_orders)))
        return ring_stack.reshape((-1, detector_y, detector_x))
    return stack


class RadialFourierAnalysis(BaseMasksAnalysis, id_="RADIAL_FOURIER"):
    def get_udf_results(self, udf_results, roi, damage):
        shape = tuple(self.dataset.shape.nav)
        # transposed for historical reasons (analysis/radialfourier.py:189-194)
        data = udf_results['intensity'].data.reshape((prod(shape), -1)).T
        orders = self.parameters['max_order'] + 1
        n_bins = self.parameters['n_bins']
        data = data.reshape((n_bins, orders, *shape))

        def resultlist():
            sets = []
            absolute = np.absolute(data)
            normal = np.maximum(1, absolute[:, 0])
            absolute[:, 0] = 0
            for b in range(n_bins):
                sets.append(AnalysisResult(
                    raw_data=np.argmax(absolute[b], axis=0), key="dominant_%s" % b,
                    title="dominant order of bin %s" % b, desc="dominant order"))
                for o in range(orders):
                    sets.append(AnalysisResult(
                        raw_data=absolute[b, o] / (normal[b] if o else 1),
                        key="absolute_%s_%s" % (b, o), title="bin %s order %s" % (b, o),
                        desc="Absolute value of Fourier component"))
            for b in range(n_bins):
                for o in range(orders):
                    sets.append(AnalysisResult(
                        raw_data=np.angle(data[b, o]), key="phase_%s_%s" % (b, o),
                        title="bin %s order %s" % (b, o), desc="Phase of Fourier component"))
            for b in range(n_bins):
                for o in range(orders):
                    sets.append(AnalysisResult(
                        raw_data=data[b, o], key="complex_%s_%s" % (b, o),
                        title="bin %s order %s" % (b, o), desc="Fourier component"))
            return sets
        return RadialFourierResultSet(resultlist, raw_results=data)

    def get_mask_factories(self):
        if self.dataset.shape.sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        detector_y, detector_x = self.dataset.shape.sig
        p = self.parameters
        return radial_mask_factory(
            detector_y=detector_y, detector_x=detector_x, cx=p['cx'], cy=p['cy'], ri=p['ri'],
            ro=p['ro'], n_bins=p['n_bins'], max_order=p['max_order'], use_sparse=p['use_sparse'])

    def get_parameters(self, parameters):
        detector_y, detector_x = self.dataset.shape.sig
        cx = parameters.get('cx', detector_x / 2)
        cy = parameters.get('cy', detector_y / 2)
        ri = parameters.get('ri', 0)
        ro = parameters.get('ro', masks.bounding_radius(cx, cy, detector_x, detector_y))
        n_bins = parameters.get('n_bins', 1)
        max_order = parameters.get('max_order', 24)
        mask_count = n_bins * (max_order + 1)
        bin_width = (ro - ri) / n_bins
        bin_area = np.pi * ro**2 - np.pi * (ro - bin_width)**2
        stack_size = mask_count * detector_y * detector_x * 8
        default = 'scipy.sparse'
        if stack_size < 2**18:
            default = False                      # fits the L3 cache comfortably
        elif bin_area / (detector_x * detector_y) > 0.05 and n_bins < 10:
            default = False                      # masks are actually dense
        return {
            'cx': cx, 'cy': cy, 'ri': ri, 'ro': ro, 'n_bins': n_bins, 'max_order': max_order,
            'use_sparse': parameters.get('use_sparse', default),
            'mask_count': mask_count, 'mask_dtype': np.complex64,
        }
