from .base import Analysis, AnalysisResult, AnalysisResultSet
from .masks import MasksAnalysis, BaseMasksAnalysis, SingleMaskAnalysis
from .disk import DiskMaskAnalysis
from .ring import RingMaskAnalysis
from .point import PointMaskAnalysis
from .sum import SumAnalysis
from .sumsig import SumSigAnalysis
from .com import COMAnalysis
from .radialfourier import RadialFourierAnalysis
from .fft import ApplyFFTMask, SumfftAnalysis
from .raw import PickFrameAnalysis, PickFFTFrameAnalysis

__all__ = ['PickFrameAnalysis', 'PickFFTFrameAnalysis', 'Analysis', 'AnalysisResult', 'AnalysisResultSet', 'MasksAnalysis',
           'BaseMasksAnalysis', 'SingleMaskAnalysis', 'DiskMaskAnalysis', 'RingMaskAnalysis',
           'PointMaskAnalysis', 'SumAnalysis', 'SumSigAnalysis', 'COMAnalysis',
           'RadialFourierAnalysis', 'ApplyFFTMask', 'SumfftAnalysis']
