"""Per-frame sum (reference analysis/sumsig.py)."""
from libertem_amd.udf.sumsigudf import SumSigUDF
from .base import BaseAnalysis, AnalysisResult, AnalysisResultSet


class SumSigAnalysis(BaseAnalysis, id_="SUM_SIG"):
    def get_udf(self):
        return SumSigUDF()

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return AnalysisResultSet([
            AnalysisResult(raw_data=data, key='intensity', title='intensity',
                           desc='result from integration over whole frames'),
        ])
