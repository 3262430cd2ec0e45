"""raytracer_amd -- MI355X-native ray casting / path tracing behind a C ABI.

The package holds the HIP kernels and host code (csrc/), the ctypes binding of
the C ABI (_capi, api) and the host-side mirror of the reference backend's
entry points (backend).  Importing it does not load the native library; the
first call does, and fails loudly if the library is not built.
"""
from . import build  # noqa: F401

__all__ = ["api", "backend", "scenes", "build"]
