"""Mask-based analyses -> ApplyMasksUDF (reference analysis/masks.py:6-184)."""
from .base import BaseAnalysis, AnalysisResultSet, AnalysisResult
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
    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return self.get_generic_results(data[..., 0], damage=damage)

    def get_description(self):
        raise NotImplementedError

    def get_generic_results(self, data, damage):
        if data.dtype.kind == 'c':
            return SingleMaskResultSet(self.get_complex_results(
                data, key_prefix='intensity', title='intensity', desc=self.get_description(),
                damage=damage))
        return SingleMaskResultSet([
            AnalysisResult(raw_data=data, key='intensity', title='intensity [log]',
                           desc=self.get_description()),
            AnalysisResult(raw_data=data, key='intensity_lin', title='intensity [lin]',
                           desc=self.get_description()),
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

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return self.get_generic_results(data, damage=damage)
