from .memory import MemoryDataSet
from .raw import RawFileDataSet
from .stream import StreamDataSet
from .mib import MIBDataSet
from .base import DataSet, DataSetException, Partition, DataTile, TilingScheme, Negotiator


def load(filetype, *args, **kwargs):
    """In-memory arrays (host or HBM), flat binary files, Merlin .mib files and frame streams of a
    running acquisition; the other file formats of the reference are out of scope of this build."""
    if filetype in ('memory', 'mem'):
        return MemoryDataSet(*args, **kwargs)
    if filetype == 'raw':
        return RawFileDataSet(*args, **kwargs)
    if filetype == 'mib':
        return MIBDataSet(*args, **kwargs)
    if filetype in ('stream', 'live'):
        return StreamDataSet(*args, **kwargs)
    raise DataSetException(
        f"dataset type {filetype!r} is not available: 'memory', 'raw', 'mib' and 'stream' are in scope of "
        "this build")


__all__ = ['MemoryDataSet', 'RawFileDataSet', 'StreamDataSet', 'MIBDataSet', 'DataSet', 'DataSetException', 'Partition', 'DataTile',
           'TilingScheme', 'Negotiator', 'load']
