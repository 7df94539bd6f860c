from .shape import Shape
from .slice import Slice, SliceUsageError

__all__ = ['Shape', 'Slice', 'SliceUsageError']
