"""volumerenderercl_amd -- MI355X-native (gfx950, hand-written HIP) replacement for the
ray-cast hot path of vbruder/VolumeRendererCL, behind the reference's VolumeRenderCL
interface.  The product is `libvrhip.so` (csrc/, C ABI in include/vrhip.h); this package is
the thin host-side mirror used by tests, bench.py and the multi-GPU tile driver."""
from . import _lib, frontend
from ._lib import (CameraParams, PathtraceParams, RaycastParams, RenderingParams, Stats,
                   UCHAR, USHORT, FLOAT)
from .renderer import VolumeRenderCL

__all__ = ["VolumeRenderCL", "frontend", "_lib", "CameraParams", "RenderingParams",
           "RaycastParams", "PathtraceParams", "Stats", "UCHAR", "USHORT", "FLOAT"]
