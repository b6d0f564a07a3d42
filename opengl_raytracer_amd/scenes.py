"""Scene inputs: the reference's default camera and lights, and seeded synthetic
meshes and textures.  Everything here is plain numpy (input data, not compute).
"""
import math

import numpy as np

from ._native import LIGHT_DTYPE, VERTEX_DTYPE

f32 = np.float32


# ----------------------------------------------------------------------------
# camera  ("OpenGL Raytracer/camera.cpp":50-65)
# ----------------------------------------------------------------------------
def camera_basis(vertical_angle, horizontal_angle):
    """direction/right/up from the two angles (radians), in float32 like the reference.

    right uses 3.14f/2, not pi/2, exactly as camera.cpp:58-62 does.
    """
    v, h = f32(vertical_angle), f32(horizontal_angle)
    direction = np.array([np.cos(v) * np.sin(h), np.sin(v), np.cos(v) * np.cos(h)], dtype=f32)
    hr = f32(h - f32(3.14) / f32(2.0))
    right = np.array([np.sin(hr), f32(0), np.cos(hr)], dtype=f32)
    # glm::cross(right, direction)
    x, y = right, direction
    up = np.array([x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]], dtype=f32)
    return direction, up, right


def default_camera():
    """The pose main.cpp:402-406 sets: position, vertical 24.82 rad, horizontal -82.44 rad."""
    pos = np.array([2.77904, 0.594951, -1.95046], dtype=f32)
    direction, up, right = camera_basis(24.82, -82.44)
    return pos, direction, up, right


def look_at_camera(pos, target, world_up=(0.0, 1.0, 0.0)):
    """An orthonormal pinhole basis looking from pos at target (for the synthetic scenes)."""
    pos = np.asarray(pos, dtype=np.float64)
    d = np.asarray(target, dtype=np.float64) - pos
    d /= np.linalg.norm(d)
    r = np.cross(np.asarray(world_up, dtype=np.float64), d)
    r /= np.linalg.norm(r)
    u = np.cross(d, r)
    return pos.astype(f32), d.astype(f32), u.astype(f32), r.astype(f32)


# ----------------------------------------------------------------------------
# lights  (main.cpp:266-285, 378-386)
# ----------------------------------------------------------------------------
def msvc_rand(seed=1):
    """The MSVC C runtime rand(): x = x*214013 + 2531011; return (x >> 16) & 0x7fff."""
    x = seed & 0xFFFFFFFF
    while True:
        x = (x * 214013 + 2531011) & 0xFFFFFFFF
        yield (x >> 16) & 0x7FFF


def default_lights(angle_offset=0.0, seed=1):
    """Ten lights on the radius-2 circle at z = -2 (SetLightPosition, main.cpp:276-285)
    with Rand0To1() colours from an unseeded MSVC rand() (main.cpp:271-274, 383)."""
    lights = np.zeros(10, dtype=LIGHT_DTYPE)
    rnd = msvc_rand(seed)
    angle = f32(0.0)
    incr = f32(2.0 * math.pi / 10)
    for i in range(10):
        a = f32(angle + f32(angle_offset))
        lights[i]["pos"] = (f32(0) + f32(2) * np.cos(a), f32(0) + f32(2) * np.sin(a), f32(-2))
        angle = f32(angle + incr)
    for i in range(10):
        lights[i]["color"] = [f32(next(rnd) / 32767.0) for _ in range(3)]
        lights[i]["pad0"] = -1.0
        lights[i]["pad1"] = -1.0
    return lights


# ----------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------
def pack_vertices(points, texcoords):
    """(n,3) positions + (n,2) texture coordinates -> 32-byte records (main.cpp:150-163)."""
    points = np.asarray(points, dtype=f32).reshape(-1, 3)
    texcoords = np.asarray(texcoords, dtype=f32).reshape(-1, 2)
    v = np.zeros(points.shape[0], dtype=VERTEX_DTYPE)
    v["pos"] = points
    v["texCoord"] = texcoords
    return v


def triangle_soup(num_tris, seed=12345, extent=1.0, size=0.05):
    """Seeded random triangles: centroid uniform in [-extent,extent]^3, each vertex the
    centroid + uniform[-size,size]^3, texture coordinates uniform in [0,1)^2."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-extent, extent, size=(num_tris, 1, 3))
    p = c + rs.uniform(-size, size, size=(num_tris, 3, 3))
    uv = rs.uniform(0.0, 1.0, size=(num_tris * 3, 2))
    return pack_vertices(p.reshape(-1, 3), uv)


def quad_grid(nx, ny, origin=(-1.0, -1.0, 0.0), du=(2.0, 0.0, 0.0), dv=(0.0, 2.0, 0.0)):
    """A UV-mapped rectangle cut into nx*ny cells of two triangles each."""
    o, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
    pts, uvs = [], []
    for j in range(ny):
        for i in range(nx):
            u0, u1, v0, v1 = i / nx, (i + 1) / nx, j / ny, (j + 1) / ny
            corners = [(u0, v0), (u1, v0), (u1, v1), (u0, v1)]
            for tri in ((0, 1, 2), (0, 2, 3)):
                for k in tri:
                    u, v = corners[k]
                    pts.append(o + u * du + v * dv)
                    uvs.append((u, v))
    return pack_vertices(np.array(pts), np.array(uvs))


def blob(rows=94, cols=152, radius=0.8, center=(0.0, 0.0, 0.0), bumps=0.15, seed=7):
    """A closed, bumpy latitude/longitude sphere with 2*rows*cols triangles (the default
    94 x 152 gives 28 576, the triangle count of the reference's bunny.blend mesh).
    Texture coordinates are the planar (longitude, latitude) map."""
    rs = np.random.RandomState(seed)
    ph = rs.uniform(0, 2 * np.pi, size=6)
    th = np.linspace(0.0, np.pi, rows + 1)
    ps = np.linspace(0.0, 2 * np.pi, cols + 1)
    T, P = np.meshgrid(th, ps, indexing="ij")
    r = radius * (1.0 + bumps * (np.sin(3 * T + ph[0]) * np.cos(4 * P + ph[1]) * 0.5
                                 + np.sin(7 * T + ph[2]) * np.sin(5 * P + ph[3]) * 0.3
                                 + np.cos(11 * P + ph[4]) * np.sin(T) * 0.2))
    r[:, -1] = r[:, 0]
    x = r * np.sin(T) * np.cos(P) + center[0]
    y = r * np.cos(T) + center[1]
    z = r * np.sin(T) * np.sin(P) + center[2]
    pos = np.stack([x, y, z], axis=-1)
    uv = np.stack([P / (2 * np.pi), 1.0 - T / np.pi], axis=-1)
    i, j = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    i, j = i.ravel(), j.ravel()
    quads = [(i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)]
    idx = []
    for tri in ((0, 1, 2), (0, 2, 3)):
        idx.append(np.stack([np.stack(quads[k], axis=-1) for k in tri], axis=1))
    idx = np.stack(idx, axis=1).reshape(-1, 3, 2)  # (rows*cols*2, 3, 2)
    pts = pos[idx[..., 0], idx[..., 1]].reshape(-1, 3)
    uvs = uv[idx[..., 0], idx[..., 1]].reshape(-1, 2)
    return pack_vertices(pts, uvs)


def concat_meshes(*meshes):
    return np.concatenate([np.asarray(m, dtype=VERTEX_DTYPE) for m in meshes])


# ----------------------------------------------------------------------------
# textures (RGBA8, row 0 = bottom row)
# ----------------------------------------------------------------------------
def noise_texture(w=256, h=256, seed=99):
    rs = np.random.RandomState(seed)
    t = rs.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
    t[..., 3] = 255
    return t


def checker_texture(w=16, h=16, cell=4):
    """A 4-colour checker in the palette of the reference's 16x16 bth.tga."""
    pal = np.array([[0, 0, 0, 255], [16, 0, 255, 255], [29, 255, 0, 255], [255, 247, 250, 255]], dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return pal[((xx // cell) + 2 * (yy // cell)) % 4]
