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

from ._lib import ABI_VERSION as _ABI

__version__ = "%d.%d.%d" % (_ABI // 10000, (_ABI // 100) % 100, _ABI % 100)   # = fourq_version() of the library this package binds (10000*major + 100*minor + patch)
