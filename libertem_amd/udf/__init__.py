from .base import (
    UDF, UDFMeta, UDFRunner, UDFData, NoOpUDF, check_cast, UDFFrameMixin, UDFTileMixin, UDFPartitionMixin,
    UDFPostprocessMixin, UDFPreprocessMixin, UDFMergeAllMixin,
)
from libertem_amd.common.udf import UDFMethod
from libertem_amd.common.exceptions import UDFRunCancelled, UDFException
from .auto import AutoUDF

__all__ = ['UDF', 'UDFFrameMixin', 'UDFTileMixin', 'UDFPartitionMixin', 'UDFPostprocessMixin', 'UDFPreprocessMixin',
           'UDFMergeAllMixin', 'UDFMeta', 'UDFRunner', 'UDFData', 'NoOpUDF', 'UDFMethod', 'check_cast', 'AutoUDF',
           'UDFRunCancelled', 'UDFException']
