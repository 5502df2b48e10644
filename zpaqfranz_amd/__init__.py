"""zpaqfranz_amd -- MI355X (gfx950) engine for zpaqfranz's block compress/decompress hot path.

The product is the C-ABI shared library built from csrc/*.hip (declared in include/zpaqhip.h);
`zpaqfranz_amd.engine.Engine` is a thin ctypes mirror of that ABI for tests, bench.py and Python
hosts.  There is no CPU fallback: without the HIP library or a gfx950 device every entry point
raises."""
from .engine import Engine, ZpqError, lib_path  # noqa: F401
