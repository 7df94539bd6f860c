"""
Device selection through environment variables, mirroring the reference's
libertem.common.backend (common/backend.py:22-118) with HIP in place of CUDA:

    LIBERTEM_USE_HIP=<ordinal>   this worker drives GPU <ordinal>
    LIBERTEM_USE_CPU=<n>         this worker is a CPU worker
"""
import os
import contextlib

_USE_HIP = 'LIBERTEM_USE_HIP'
_USE_CPU = 'LIBERTEM_USE_CPU'


def set_use_hip(device):
    os.environ[_USE_HIP] = str(int(device))
    os.environ.pop(_USE_CPU, None)


def set_use_cpu(cpu):
    os.environ[_USE_CPU] = str(int(cpu))
    os.environ.pop(_USE_HIP, None)


def get_use_hip():
    v = os.environ.get(_USE_HIP)
    return int(v) if v not in (None, '') else None


def get_use_cpu():
    v = os.environ.get(_USE_CPU)
    return int(v) if v not in (None, '') else None


def get_device_class():
    """'hip' if this process was told to drive a GPU, else 'cpu' (reference: :96-118)."""
    return 'hip' if get_use_hip() is not None else 'cpu'


@contextlib.contextmanager
def set_device_class(device_class, device=0):
    """Test helper analogous to tests/utils.py:392-416 of the reference."""
    prev = {k: os.environ.get(k) for k in (_USE_HIP, _USE_CPU)}
    try:
        if device_class == 'hip':
            set_use_hip(device)
        elif device_class == 'cpu':
            set_use_cpu(0)
        else:
            raise ValueError(f"unknown device class {device_class!r}")
        yield
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
