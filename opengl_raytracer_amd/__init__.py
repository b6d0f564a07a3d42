"""MI355X-native three-stage compute ray tracer (raygen -> intersect -> color).

The compute path is the HIP library librt_hip.so behind the C ABI of include/rt.h;
this package is its thin host-side mirror.  There is no CPU fallback.
"""
from ._native import (LIGHT_DTYPE, RAY_DTYPE, RT_MAX_LIGHTS, RT_TRACE_CLUSTERED, RT_TRACE_EXACT, RT_TRACE_FILTERED,
                      VERTEX_DTYPE, RtError)
from .raytracer import RayTracer
from . import scenes

__all__ = ["RayTracer", "RtError", "RAY_DTYPE", "VERTEX_DTYPE", "LIGHT_DTYPE", "RT_MAX_LIGHTS",
           "RT_TRACE_FILTERED", "RT_TRACE_EXACT", "RT_TRACE_CLUSTERED", "scenes"]
