"""poolgen_amd -- MI355X-native implementation of poolgen's per-locus regression hot path.

The product is the C-ABI shared library ``csrc/libpoolgen_hip.so`` (hand-written HIP for
gfx950, see ``include/poolgen_hip.h``) plus the C++ ``poolgen`` CLI.  This Python package is
plumbing only: a ctypes binding over the C ABI that lends it torch device memory, streams and
``torch.distributed`` (RCCL).  There is no CPU fallback anywhere in this package.
"""
from ._native import NativeError, load_library, library_path  # noqa: F401
from .engine import Engine, Filter  # noqa: F401

__all__ = ["Engine", "Filter", "NativeError", "load_library", "library_path"]
