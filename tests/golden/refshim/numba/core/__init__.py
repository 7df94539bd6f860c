from . import config  # noqa
