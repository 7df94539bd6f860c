"""Sum over all frames (reference analysis/sum.py:91-133)."""
import numpy as np

from libertem_amd.udf.sum import SumUDF
from .base import BaseAnalysis, AnalysisResult, AnalysisResultSet
from .getroi import get_roi


class SumResultSet(AnalysisResultSet):
    pass


class SumAnalysis(BaseAnalysis, id_="SUM_FRAMES"):
    def get_udf(self):
        dest_dtype = np.dtype(self.dataset.dtype)
        if dest_dtype.kind not in ('c', 'f'):
            dest_dtype = 'float32'
        return SumUDF(dtype=dest_dtype)

    def get_roi(self):
        # parameters = {'roi': {'shape': 'disk' | 'rect', ...}} (analysis/sum.py:100-101)
        return get_roi(params=self.parameters, shape=self.dataset.shape.nav)

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        if data.dtype.kind == 'c':
            return AnalysisResultSet(self.get_complex_results(
                data, key_prefix="intensity", title="intensity", desc="sum of all frames",
                damage=True, default_lin=False))
        return SumResultSet([
            AnalysisResult(raw_data=data, key="intensity", title="intensity [log]",
                           desc="sum of frames log-scaled"),
            AnalysisResult(raw_data=data, key="intensity_lin", title="intensity [lin]",
                           desc="sum of frames lin-scaled"),
        ])
