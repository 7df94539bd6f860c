from .base import UDF, UDFMeta, UDFRunner, UDFData, NoOpUDF, check_cast
from libertem_amd.common.udf import UDFMethod

__all__ = ['UDF', 'UDFMeta', 'UDFRunner', 'UDFData', 'NoOpUDF', 'UDFMethod', 'check_cast']
