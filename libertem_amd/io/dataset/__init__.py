from .memory import MemoryDataSet
from .base import DataSet, DataSetException, Partition, DataTile, TilingScheme, Negotiator


def load(filetype, *args, **kwargs):
    """Only the in-memory dataset is part of this build (file formats are out of scope)."""
    if filetype in ('memory', 'mem'):
        return MemoryDataSet(*args, **kwargs)
    raise DataSetException(
        f"dataset type {filetype!r} is not available: only 'memory' is in scope of this build")


__all__ = ['MemoryDataSet', 'DataSet', 'DataSetException', 'Partition', 'DataTile',
           'TilingScheme', 'Negotiator', 'load']
