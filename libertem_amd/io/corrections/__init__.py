from .corrset import CorrectionSet  # noqa: F401
