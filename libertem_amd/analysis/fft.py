"""
Fourier-space analyses (reference analysis/apply_fft_mask.py:33-59, analysis/sumfft.py:36-66):
`ApplyFFTMask` = CrystallinityUDF with GUI-style parameters; `SumfftAnalysis` = sum of all frames
plus the log-scaled, centred spectrum of that sum (what the reference renders as its picture).
"""
import numpy as np

from libertem_amd.masks import _make_circular_mask
from libertem_amd.udf.crystallinity import CrystallinityUDF
from .base import BaseAnalysis, AnalysisResult, AnalysisResultSet
from .sum import SumAnalysis


class ApplyFFTMask(BaseAnalysis, id_="APPLY_FFT_MASK"):
    def get_udf(self):
        p = self.parameters
        real_center = (p.get("real_centery"), p.get("real_centerx"))
        if real_center[0] is None or real_center[1] is None:
            real_center = None
        return CrystallinityUDF(rad_in=p["rad_in"], rad_out=p["rad_out"],
                                real_center=real_center, real_rad=p.get("real_rad"))

    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return AnalysisResultSet([
            AnalysisResult(raw_data=data, key="intensity", title="intensity",
                           desc="result from integration over mask in Fourier space"),
        ])


def log_spectrum(image, real_center=None, real_rad=None):
    """log(|fftshift(fft2(image * real_mask))| + 1)  (analysis/sumfft.py:41-52)."""
    image = np.asarray(image)
    if not (real_center is None or real_rad is None or real_center[0] is None
            or real_center[1] is None):
        sigshape = image.shape
        real_mask = 1 - 1 * _make_circular_mask(real_center[1], real_center[0], sigshape[1],
                                                sigshape[0], real_rad)
        image = image * real_mask
    return np.log(abs(np.fft.fftshift(np.fft.fft2(image))) + 1)


class SumfftAnalysis(SumAnalysis, id_="FFTSUM_FRAMES"):
    def get_udf_results(self, udf_results, roi, damage):
        sum_results = np.array(udf_results['intensity'].data)
        p = self.parameters
        spectrum = log_spectrum(sum_results, (p.get("real_centery"), p.get("real_centerx")),
                                p.get("real_rad"))
        return AnalysisResultSet([
            AnalysisResult(raw_data=sum_results, key="intensity", title="intensity",
                           desc="fft of sum of all frames"),
            AnalysisResult(raw_data=spectrum, key="intensity_fft", title="log spectrum",
                           desc="log(|fftshift(fft2(sum of all frames))| + 1)"),
        ])
