"""Minimal stand-in for the `gym` package (absent here), used ONLY by make_fixtures.py so the
reference's wrapper modules import.  Only `spaces.Box` is provided."""
from . import spaces  # noqa: F401
