from .shape import Shape
from .slice import Slice

__all__ = ['Shape', 'Slice']
