"""MI355X-native trace path for clemy/RayTracer scenes.

The package holds only what the hot path needs: the HIP kernels + C ABI (csrc/), the host-side
mirror of the reference interface (host/) and the ctypes plumbing (capi.py).
"""
from .capi import (Bands, Context, RtError, Scene, SceneError, Stats, apng_write, keyframe_eval,  # noqa: F401
                   png_read, png_write, quantize)
