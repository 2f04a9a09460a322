"""fourq_amd -- batched FourQ (Curve4Q) scalar multiplication on AMD MI355X (gfx950).

`fourq_amd.curve4q` mirrors the reference module `impl/curve4q.py` (same names and conventions);
`fourq_amd.Engine` is the array-level batch interface over the C ABI (include/fourq_amd.h).
Importing the package touches neither the GPU nor the shared library; the first call does, and
fails loudly if either is missing.
"""
from . import codec, constants  # noqa: F401
from ._lib import FourQError  # noqa: F401
from .engine import Engine, default_engine  # noqa: F401
from .multi import MultiEngine, device_count  # noqa: F401

__version__ = "0.3.0"
