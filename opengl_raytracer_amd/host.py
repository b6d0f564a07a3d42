"""ctypes binding of librt_host.so (the C ABI declared in include/rt_host.h): OBJ/MTL meshes,
TGA/PNG images, camera basis and matrices, the light rig, the benchmark CSV line and the
light-marker overlay.  Plain CPU code; the parsing and decoding all happen in the C++ library.
"""
import ctypes as C
import os

import numpy as np

from ._native import LIGHT_DTYPE, RT_MAX_LIGHTS, VERTEX_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HOST_LIB") or os.path.join(_HERE, "librt_host.so")

# struct VertexData of "OpenGL Raytracer/OBJ.h":7-24
VERTEX_DATA_DTYPE = np.dtype([("point", "<f4", 3), ("texCoord", "<f4", 2), ("normal", "<f4", 3)])
assert VERTEX_DATA_DTYPE.itemsize == 32


class Material(C.Structure):
    """struct MeshMaterialData of OBJ.h:26-47 as rt_host.h lays it out."""
    _fields_ = [("name", C.c_char * 128), ("illum", C.c_uint32),
                ("Kd", C.c_float * 3), ("Ka", C.c_float * 3), ("Tf", C.c_float * 3),
                ("map_Kd", C.c_char * 256), ("Ni", C.c_float)]

    def as_dict(self):
        return {"name": self.name.decode(), "illum": int(self.illum), "Kd": list(self.Kd), "Ka": list(self.Ka),
                "Tf": list(self.Tf), "map_Kd": self.map_Kd.decode(), "Ni": float(self.Ni)}


_P = C.c_void_p
_F = C.POINTER(C.c_float)
PROTOTYPES = {
    "rth_mesh_load": (C.c_int, [C.POINTER(_P), C.c_char_p, C.c_char_p]),
    "rth_mesh_free": (None, [_P]),
    "rth_mesh_group_count": (C.c_int, [_P]),
    "rth_mesh_group_name": (C.c_char_p, [_P, C.c_int]),
    "rth_mesh_vertex_count": (C.c_int, [_P, C.c_int]),
    "rth_mesh_vertex_data_size": (C.c_int, [_P, C.c_int]),
    "rth_mesh_vertex_data": (_P, [_P, C.c_int]),
    "rth_mesh_material": (C.c_int, [_P, C.c_int, C.POINTER(Material)]),
    "rth_load_materials": (C.c_int, [C.c_char_p, C.POINTER(Material), C.c_int]),
    "rth_pack_vertices": (None, [_P, C.c_int, _P]),
    "rth_image_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P)]),
    "rth_image_load_memory": (C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(_P)]),
    "rth_image_free": (None, [_P]),
    "rth_write_tga": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int, C.c_int]),
    "rth_write_png": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "rth_camera_basis": (None, [C.c_float, C.c_float, _F, _F, _F]),
    "rth_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _F]),
    "rth_look_at": (None, [_F, _F, _F, _F]),
    "rth_default_lights": (None, [_P]),
    "rth_set_light_positions": (None, [_P, C.c_int, C.c_float]),
    "rth_light_angle_offset": (C.c_float, [C.c_double]),
    "rth_csv_append": (C.c_int, [C.c_char_p] + [C.c_int] * 9),
    "rth_draw_light_markers": (None, [_P, C.c_int, C.c_int, _F, _F, _P, C.c_int, C.c_float]),
    "rth_last_error": (C.c_char_p, []),
}

_lib = None


class HostError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C opengl_raytracer_amd/csrc/host` "
                          "(or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _err():
    return (load().rth_last_error() or b"").decode(errors="replace")


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


class Mesh:
    """The reference's `Mesh` ("OpenGL Raytracer/OBJ.h":56-79) over rth_mesh_*."""

    def __init__(self):
        self._h = None

    def LoadFromObjFile(self, dir, filename):
        lib = load()
        self.close()
        h = _P()
        if lib.rth_mesh_load(C.byref(h), os.fsencode(dir), os.fsencode(filename)) != 0:
            raise HostError(_err())
        self._h = h
        return self

    def close(self):
        if self._h is not None:
            load().rth_mesh_free(self._h)
            self._h = None

    __del__ = close

    def GetGroupCount(self):
        return load().rth_mesh_group_count(self._h)

    def GetGroupName(self, group):
        s = load().rth_mesh_group_name(self._h, group)
        if s is None:
            raise IndexError(group)
        return s.decode()

    def GetVertexCount(self, group):
        return load().rth_mesh_vertex_count(self._h, group)

    def GetVertexDataSize(self, group):
        return load().rth_mesh_vertex_data_size(self._h, group)

    def GetVertexData(self, group):
        """A copy of the group's VertexData records as a structured array."""
        lib = load()
        n = lib.rth_mesh_vertex_count(self._h, group)
        p = lib.rth_mesh_vertex_data(self._h, group)
        if not p or n == 0:
            return np.zeros(0, dtype=VERTEX_DATA_DTYPE)
        return np.frombuffer(C.string_at(p, n * 32), dtype=VERTEX_DATA_DTYPE).copy()

    def GetGroupMaterialData(self, group):
        m = Material()
        if load().rth_mesh_material(self._h, group, C.byref(m)) != 0:
            raise IndexError(group)
        return m.as_dict()

    def packed(self, group=0):
        """The group as the 32-byte records rt_set_mesh takes (UploadToSSBO, main.cpp:148-163)."""
        lib = load()
        n = lib.rth_mesh_vertex_count(self._h, group)
        out = np.zeros(n, dtype=VERTEX_DTYPE)
        if n:
            lib.rth_pack_vertices(lib.rth_mesh_vertex_data(self._h, group), n, out.ctypes.data)
        return out


def load_materials(path, max_materials=64):
    arr = (Material * max_materials)()
    n = load().rth_load_materials(os.fsencode(path), arr, max_materials)
    if n < 0:
        raise HostError(_err())
    return [arr[i].as_dict() for i in range(min(n, max_materials))]


def _take_image(rc, w, h, p):
    lib = load()
    if rc != 0:
        raise HostError(_err())
    try:
        return np.frombuffer(C.string_at(p, w.value * h.value * 4), dtype=np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        lib.rth_image_free(p)


def load_image(path, invert_y=True):
    """(h, w, 4) uint8; invert_y=True gives row 0 = bottom, the order rt_set_texture_rgba8 expects."""
    w, h, p = C.c_int(), C.c_int(), _P()
    rc = load().rth_image_load(os.fsencode(path), int(invert_y), C.byref(w), C.byref(h), C.byref(p))
    return _take_image(rc, w, h, p)


def decode_image(data, invert_y=True):
    w, h, p = C.c_int(), C.c_int(), _P()
    buf = bytes(data)
    rc = load().rth_image_load_memory(buf, len(buf), int(invert_y), C.byref(w), C.byref(h), C.byref(p))
    return _take_image(rc, w, h, p)


def write_tga(path, rgba_bottom_up, rle=False):
    a = np.ascontiguousarray(rgba_bottom_up, dtype=np.uint8)
    if load().rth_write_tga(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], int(rle)) != 0:
        raise HostError(_err())


def write_png(path, rgba_bottom_up):
    a = np.ascontiguousarray(rgba_bottom_up, dtype=np.uint8)
    if load().rth_write_png(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]) != 0:
        raise HostError(_err())


def camera_basis(horizontal_angle, vertical_angle):
    """(direction, right, up) as float32 arrays, camera.cpp:50-65."""
    d, r, u = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    load().rth_camera_basis(horizontal_angle, vertical_angle, d, r, u)
    return tuple(np.array(list(v), dtype=np.float32) for v in (d, r, u))


def perspective(fovy, aspect, z_near, z_far):
    m = (C.c_float * 16)()
    load().rth_perspective(fovy, aspect, z_near, z_far, m)
    return np.array(list(m), dtype=np.float32)


def look_at(eye, center, up):
    m = (C.c_float * 16)()
    load().rth_look_at(_f3(eye), _f3(center), _f3(up), m)
    return np.array(list(m), dtype=np.float32)


def default_lights():
    lights = np.zeros(RT_MAX_LIGHTS, dtype=LIGHT_DTYPE)
    load().rth_default_lights(lights.ctypes.data)
    return lights


def set_light_positions(lights, n, angle_offset):
    assert lights.dtype == LIGHT_DTYPE and lights.flags.c_contiguous and 0 <= n <= len(lights)
    load().rth_set_light_positions(lights.ctypes.data, n, angle_offset)


def light_angle_offset(time_seconds):
    return float(load().rth_light_angle_offset(time_seconds))


def csv_append(filename, thread_group_size, width, height, passes, num_lights, num_triangles, raygen_ms,
               intersect_ms, color_ms):
    if load().rth_csv_append(os.fsencode(filename), thread_group_size, width, height, passes, num_lights,
                             num_triangles, int(raygen_ms), int(intersect_ms), int(color_ms)) != 0:
        raise HostError(_err())


def draw_light_markers(rgba_bottom_up, view, projection, lights, n, point_size=30.0):
    a = rgba_bottom_up
    assert a.dtype == np.uint8 and a.flags.c_contiguous and a.ndim == 3 and a.shape[2] == 4
    v = (C.c_float * 16)(*[float(x) for x in view])
    p = (C.c_float * 16)(*[float(x) for x in projection])
    load().rth_draw_light_markers(a.ctypes.data, a.shape[1], a.shape[0], v, p, lights.ctypes.data, n, point_size)
