"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
PARITY UNPINNED by reference fixtures (the reference has none); see rt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

RAY_DTYPE = np.dtype([
    ("origin", "<f4", 3), ("pad0", "<f4"), ("dir", "<f4", 3), ("pad1", "<f4"),
    ("color", "<f4", 3), ("t", "<f4"), ("primitiveID", "<i4"), ("pad2", "<f4", 3),
    ("n", "<f4", 3), ("pad5", "<f4"),
])
VERTEX_DTYPE = np.dtype([("pos", "<f4", 3), ("pad0", "<f4"), ("texCoord", "<f4", 2), ("pad1", "<f4", 2)])
LIGHT_DTYPE = np.dtype([("pos", "<f4", 3), ("pad0", "<f4"), ("color", "<f4", 3), ("pad1", "<f4")])


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        P = C.c_void_p
        lib.rto_raygen.argtypes = [P, C.c_int, C.c_int, P, C.c_int, C.POINTER(Camera)]
        lib.rto_intersect.argtypes = [P, C.c_int, P, C.c_int, C.POINTER(Camera), P]
        lib.rto_color.argtypes = [P, C.c_int, P, C.c_int, P, C.c_int, C.POINTER(Camera), P, C.c_int, C.c_int, P]
        lib.rto_render.argtypes = [P, C.c_int, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int, C.POINTER(Camera), P,
                                   C.c_int, C.c_int, C.c_int, P, P]
        lib.rto_ray_vs_triangle.argtypes = [P, P, P, P, P, P]
        lib.rto_ray_vs_triangle.restype = C.c_int
        lib.rto_ray_sphere.argtypes = [P, P, P, C.c_float, P]
        lib.rto_ray_sphere.restype = C.c_int
        lib.rto_set_threads.argtypes = [C.c_int]
        lib.rto_get_threads.restype = C.c_int
        _lib = lib
    return _lib


def _cam(pos, direction, up, right):
    c = Camera()
    for name, v in (("pos", pos), ("dir", direction), ("up", up), ("right", right)):
        a = np.asarray(v, dtype=np.float32).reshape(3)
        setattr(c, name, (C.c_float * 3)(*[float(x) for x in a]))
    return c


class Oracle:
    """Same surface as the product's RayTracer, computed on the CPU."""

    def __init__(self, width, height, threads=1):
        self.lib = load()
        self.width, self.height = int(width), int(height)
        self.rows = np.arange(self.height, dtype=np.int32)
        self.threads = threads
        self.vertices = np.zeros(0, dtype=VERTEX_DTYPE)
        self.lights = np.zeros(0, dtype=LIGHT_DTYPE)
        self.tex = None
        self.cam = None
        self._alloc()

    def _alloc(self):
        n = self.rows.shape[0] * self.width
        self.rays = np.zeros(n, dtype=RAY_DTYPE)
        self.pixels = np.zeros((self.rows.shape[0], self.width, 4), dtype=np.uint8)
        self.depth = np.zeros((self.rows.shape[0], self.width), dtype=np.float32)

    @property
    def num_rays(self):
        return self.rays.shape[0]

    def set_mesh(self, vertices):
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)

    def set_texture(self, texels):
        self.tex = None if texels is None else np.ascontiguousarray(texels, dtype=np.uint8)

    def set_lights(self, lights, num_lights=None):
        l = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        self.lights = l if num_lights is None else l[:num_lights].copy()

    def set_camera(self, pos, direction, up, right):
        self.cam = _cam(pos, direction, up, right)

    def set_row_tiles(self, tile_rows, first_tile=0, tile_stride=1):
        ntiles = (self.height + tile_rows - 1) // tile_rows
        rows = []
        for t in range(first_tile, ntiles, tile_stride):
            rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, self.height)))
        self.rows = np.array(rows, dtype=np.int32)
        self._alloc()

    def row_map(self):
        return self.rows.copy()

    @property
    def local_rows(self):
        return int(self.rows.shape[0])

    def raygen(self):
        self.lib.rto_set_threads(self.threads)
        self.lib.rto_raygen(self.rays.ctypes.data, self.width, self.height, self.rows.ctypes.data,
                            self.rows.shape[0], C.byref(self.cam))

    def intersect(self):
        self.lib.rto_set_threads(self.threads)
        self.lib.rto_intersect(self.rays.ctypes.data, self.num_rays, self.vertices.ctypes.data,
                               self.vertices.shape[0], C.byref(self.cam), self.depth.ctypes.data)

    def color(self):
        self.lib.rto_set_threads(self.threads)
        tex = self.tex
        self.lib.rto_color(self.rays.ctypes.data, self.num_rays, self.vertices.ctypes.data, self.vertices.shape[0],
                           self.lights.ctypes.data if self.lights.shape[0] else None, self.lights.shape[0],
                           C.byref(self.cam), tex.ctypes.data if tex is not None else None,
                           tex.shape[1] if tex is not None else 0, tex.shape[0] if tex is not None else 0,
                           self.pixels.ctypes.data)

    def render(self, passes=1):
        self.raygen()
        for _ in range(passes):
            self.intersect()
            self.color()

    def read_rays(self):
        return self.rays.copy()

    def write_rays(self, rays):
        self.rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE).copy()

    def read_pixels(self):
        return self.pixels.copy()

    def read_depth(self):
        return self.depth.copy()


def ray_vs_triangle(o, d, x, y, z):
    lib = load()
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (o, d, x, y, z)]
    t = C.c_float(0)
    hit = lib.rto_ray_vs_triangle(*[a.ctypes.data for a in arrs], C.byref(t))
    return bool(hit), np.float32(t.value)


def ray_sphere(o, d, c, r):
    lib = load()
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (o, d, c)]
    t = C.c_float(0)
    hit = lib.rto_ray_sphere(*[a.ctypes.data for a in arrs], C.c_float(r), C.byref(t))
    return bool(hit), np.float32(t.value)
