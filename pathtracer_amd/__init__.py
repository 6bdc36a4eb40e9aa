"""pathtracer_amd — MI355X-native path-tracing core (drop-in for the hot path of ReinierMaas/PathTracer).

Only what the path needs: `csrc/` (HIP kernels + the C ABI of libptcore.so), `ptcore` (ctypes
mirror of the reference's Scene / Camera / frame-loop surface), `scenes` (seeded synthetic inputs).
"""
from . import scenes  # noqa: F401

__all__ = ["scenes", "ptcore"]
