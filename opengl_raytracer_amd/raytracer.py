"""Host-side mirror of the reference's stage loop ("OpenGL Raytracer/main.cpp":476-529)
over the C ABI: one RayTracer per GPU, three stage calls, the reference's pass order.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import LIGHT_DTYPE, RAY_DTYPE, VERTEX_DTYPE, RtError


def _f3(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))
    return a


# options every new RayTracer starts with (name -> value, see rt_option in include/rt.h); the test
# suite lowers thresholds here so that small scenes reach the kernels meant for large ones
DEFAULT_OPTIONS = {}


class RayTracer:
    """One context on one GPU (rt_create ... rt_destroy)."""

    def __init__(self, width, height, device=0, options=None):
        self._lib = _native.load()
        self._ctx = C.c_void_p()
        self.width, self.height = int(width), int(height)
        rc = self._lib.rt_create(C.byref(self._ctx), int(device), self.width, self.height)
        if rc != 0:
            msg = self._lib.rt_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise RtError(rc, msg)
        self._keep = {}
        for name, value in {**DEFAULT_OPTIONS, **(options or {})}.items():
            self.set_option(name, value)

    # -- plumbing -----------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene ----------------------------------------------------------------
    def set_mesh(self, vertices):
        v = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        self._chk(self._lib.rt_set_mesh(self._ctx, v.ctypes.data, v.shape[0]))
        self.num_vertices = int(v.shape[0])

    def set_texture(self, texels_rgba8):
        """texels: (h, w, 4) uint8, row 0 = bottom row of the image; None removes it."""
        if texels_rgba8 is None:
            self._chk(self._lib.rt_set_texture_rgba8(self._ctx, None, 0, 0))
            return
        t = np.ascontiguousarray(texels_rgba8, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self._chk(self._lib.rt_set_texture_rgba8(self._ctx, t.ctypes.data, t.shape[1], t.shape[0]))

    def set_lights(self, lights, num_lights=None):
        l = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        n = l.shape[0] if num_lights is None else int(num_lights)
        self._chk(self._lib.rt_set_lights(self._ctx, l.ctypes.data if l.shape[0] else None, n))

    def set_camera(self, pos, direction, up, right):
        p, d, u, r = _f3(pos), _f3(direction), _f3(up), _f3(right)
        self._chk(self._lib.rt_set_camera(self._ctx, p.ctypes.data, d.ctypes.data, u.ctypes.data, r.ctypes.data))

    def set_row_tiles(self, tile_rows, first_tile=0, tile_stride=1):
        self._chk(self._lib.rt_set_row_tiles(self._ctx, int(tile_rows), int(first_tile), int(tile_stride)))

    def set_trace_mode(self, mode):
        self._chk(self._lib.rt_set_trace_mode(self._ctx, int(mode)))

    def set_option(self, name, value):
        """Tuning knob `name` (a key of _native.OPTIONS); grid_bits and band_list_max before set_mesh."""
        self._chk(self._lib.rt_set_option(self._ctx, _native.OPTIONS[name], int(value)))

    @property
    def local_rows(self):
        return self._lib.rt_local_rows(self._ctx)

    @property
    def num_rays(self):
        return self._lib.rt_num_rays(self._ctx)

    def row_map(self):
        rows = np.empty(self.local_rows, dtype=np.int32)
        self._chk(self._lib.rt_get_row_map(self._ctx, rows.ctypes.data))
        return rows

    # -- stages ---------------------------------------------------------------
    def raygen(self):
        self._chk(self._lib.rt_raygen(self._ctx))

    def intersect(self):
        self._chk(self._lib.rt_intersect(self._ctx))

    def color(self):
        self._chk(self._lib.rt_color(self._ctx))

    def render(self, passes=1):
        """raygen, then `passes` x {intersect, color} (main.cpp:476-529)."""
        self._chk(self._lib.rt_render(self._ctx, int(passes)))

    def sync(self):
        self._chk(self._lib.rt_sync(self._ctx))

    # -- results --------------------------------------------------------------
    def read_rays(self):
        out = np.empty(self.num_rays, dtype=RAY_DTYPE)
        self._chk(self._lib.rt_read_rays(self._ctx, out.ctypes.data))
        return out

    def read_rays_static(self):
        """read_rays through the static staging array (no device allocation; at most 838 860 rays)."""
        out = np.empty(self.num_rays, dtype=RAY_DTYPE)
        self._chk(self._lib.rt_read_rays_static(self._ctx, out.ctypes.data))
        return out

    def write_rays(self, rays):
        r = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        assert r.shape[0] == self.num_rays
        self._chk(self._lib.rt_write_rays(self._ctx, r.ctypes.data))

    def read_pixels(self):
        out = np.empty((self.local_rows, self.width, 4), dtype=np.uint8)
        self._chk(self._lib.rt_read_pixels_rgba8(self._ctx, out.ctypes.data))
        return out

    def read_depth(self):
        out = np.empty((self.local_rows, self.width), dtype=np.float32)
        self._chk(self._lib.rt_read_depth(self._ctx, out.ctypes.data))
        return out

    def stage_ms(self):
        ms = (C.c_float * 3)()
        self._chk(self._lib.rt_stage_ms(self._ctx, ms))
        return [float(x) for x in ms]

    def counters(self):
        c = (C.c_uint64 * 12)()
        self._chk(self._lib.rt_get_counters(self._ctx, c))
        names = ["closest_rays", "shadow_rays", "closest_pairs", "shadow_pairs",
                 "closest_candidates", "shadow_candidates", "closest_slow_steps", "shadow_slow_steps",
                 "lit_pixels", "killed_rays"]
        return {n: int(c[i]) for i, n in enumerate(names)}

    def kernel_log(self):
        """Per-launch records of the traversal kernels since the last call (structured array)."""
        out = np.zeros(1024, dtype=_native.KERNEL_RECORD_DTYPE)
        n = C.c_int(0)
        self._chk(self._lib.rt_kernel_log(self._ctx, out.ctypes.data, 1024, C.byref(n)))
        return out[:n.value].copy()

    # -- interop --------------------------------------------------------------
    def bind_stream(self, hip_stream_handle):
        """hip_stream_handle: integer value of a hipStream_t (0 = HIP's default stream)."""
        self._chk(self._lib.rt_bind_stream(self._ctx, C.c_void_p(hip_stream_handle or 0)))

    def unbind_stream(self):
        self._chk(self._lib.rt_unbind_stream(self._ctx))

    def bind_framebuffer(self, device_ptr):
        self._chk(self._lib.rt_bind_framebuffer(self._ctx, C.c_void_p(device_ptr or 0)))
