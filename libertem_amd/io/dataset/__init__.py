from .memory import MemoryDataSet
from .raw import RawFileDataSet
from .base import DataSet, DataSetException, Partition, DataTile, TilingScheme, Negotiator


def load(filetype, *args, **kwargs):
    """In-memory arrays (host or HBM) and flat binary files; the other file formats of the
    reference are out of scope of this build."""
    if filetype in ('memory', 'mem'):
        return MemoryDataSet(*args, **kwargs)
    if filetype == 'raw':
        return RawFileDataSet(*args, **kwargs)
    raise DataSetException(
        f"dataset type {filetype!r} is not available: 'memory' and 'raw' are in scope of this build")


__all__ = ['MemoryDataSet', 'RawFileDataSet', 'DataSet', 'DataSetException', 'Partition', 'DataTile',
           'TilingScheme', 'Negotiator', 'load']
