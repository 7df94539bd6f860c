"""Throw-away stand-in for the third-party `opentelemetry` package (absent in this image).
Used ONLY by tests/golden/generate_golden.py to import the Python reference; never by the product."""
from . import trace, context  # noqa
