"""Mask-based analyses -> ApplyMasksUDF (reference analysis/masks.py:6-184)."""
import numpy as np

from .base import BaseAnalysis, AnalysisResultSet, AnalysisResult
from .getroi import get_roi
from libertem_amd.udf.masks import ApplyMasksUDF


class BaseMasksAnalysis(BaseAnalysis):
    def get_udf(self):
        # the parameters of an analysis are fixed at construction: build the factories once, so
        # that repeated runs of the same analysis hand ApplyMasksUDF the same object and re-use
        # the cached device image of the stack (udf/masks.py `_cached_container`)
        factories = getattr(self, '_mask_factories_memo', None)
        if factories is None:
            factories = self._mask_factories_memo = self.get_mask_factories()
        return ApplyMasksUDF(
            mask_factories=factories,
            use_sparse=self.get_use_sparse(),
            mask_count=self.get_preset_mask_count(),
            mask_dtype=self.get_preset_mask_dtype(),
            preferred_dtype=self.get_preset_dtype(),
        )

    def get_mask_factories(self):
        raise NotImplementedError()

    def get_use_sparse(self):
        return self.parameters.get('use_sparse', None)

    def get_preset_mask_count(self):
        return self.parameters.get('mask_count', None)

    def get_preset_mask_dtype(self):
        return self.parameters.get('mask_dtype', None)

    def get_preset_dtype(self):
        return self.parameters.get('dtype', None)


class SingleMaskResultSet(AnalysisResultSet):
    pass


class MasksResultSet(AnalysisResultSet):
    pass


class SingleMaskAnalysis(BaseMasksAnalysis):
    """One mask on a 2D detector.  Subclasses are declarative: `WHAT` (text of the result),
    `geometry(det_y, det_x, given)` -> the geometric parameters with their defaults filled in, and
    `mask(p, det_y, det_x)` -> the mask (anything a mask factory may return); `SPARSE` fixes
    `use_sparse` for masks that are sparse by nature."""
    WHAT = None
    SPARSE = None

    def geometry(self, det_y, det_x, given):
        raise NotImplementedError

    def mask(self, p, det_y, det_x):
        raise NotImplementedError

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return self.get_generic_results(data[..., 0], damage=damage)

    def get_description(self):
        return "intensity of the integration over the selected %s" % self.WHAT

    def get_use_sparse(self):
        return self.SPARSE if self.SPARSE is not None else super().get_use_sparse()

    def get_mask_factories(self):
        sig = self.dataset.shape.sig
        if sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        det_y, det_x = tuple(sig)
        p = dict(self.parameters)
        return [lambda: self.mask(p, det_y, det_x)]

    def get_parameters(self, parameters):
        det_y, det_x = tuple(self.dataset.shape.sig)
        out = dict(self.geometry(det_y, det_x, parameters))
        if self.SPARSE is None:
            out['use_sparse'] = parameters.get('use_sparse', False)
        out['mask_count'] = 1
        out['mask_dtype'] = np.float32
        return out

    def get_generic_results(self, data, damage):
        if data.dtype.kind == 'c':
            return SingleMaskResultSet(self.get_complex_results(
                data, key_prefix='intensity', title='intensity', desc=self.get_description(),
                damage=damage))
        return SingleMaskResultSet([
            # (keys as in analysis/masks.py:63-76: the linear one is 'intensity', the log-scaled view 'intensity_log';
            #  both carry the same numbers -- scaling is a matter of the plot)
            AnalysisResult(raw_data=data, key='intensity', title='intensity [lin]',
                           desc=f'{self.get_description()} lin-scaled'),
            AnalysisResult(raw_data=data, key='intensity_log', title='intensity [log]',
                           desc=f'{self.get_description()} log-scaled'),
        ])


class MasksAnalysis(BaseMasksAnalysis, id_="APPLY_MASKS"):
    def get_mask_factories(self):
        return self.parameters['factories']

    def get_generic_results(self, data, damage):
        if data.dtype.kind == 'c':
            results = []
            for idx in range(data.shape[-1]):
                results.extend(self.get_complex_results(
                    data[..., idx], key_prefix="mask_%d" % idx, title="mask %d" % idx,
                    desc="integrated intensity for mask %d" % idx, damage=damage))
            return MasksResultSet(results)
        return MasksResultSet([
            AnalysisResult(raw_data=data[..., idx], key="mask_%d" % idx, title="mask %d" % idx,
                           desc="integrated intensity for mask %d" % idx)
            for idx in range(data.shape[-1])
        ])

    def get_roi(self):
        # parameters = {'factories': ..., 'roi': {'shape': 'disk' | 'rect', ...}} (analysis/masks.py:179-180)
        return get_roi(params=self.parameters, shape=self.dataset.shape.nav)

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return self.get_generic_results(data, damage=damage)
