"""Shared scene builders and comparison helpers for the tests."""
import numpy as np

from opengl_raytracer_amd import scenes

f32 = np.float32

FIELDS = ("origin", "dir", "color", "t", "primitiveID", "n")


def small_scene(kind="mixed", seed=3):
    """A few hundred triangles in front of a camera at z = -3, textured, two lights."""
    if kind == "mixed":
        mesh = scenes.concat_meshes(
            scenes.quad_grid(4, 4, origin=(-1.0, -1.0, 1.0)),
            scenes.triangle_soup(300, seed=seed, extent=0.8, size=0.2),
            scenes.quad_grid(2, 3, origin=(-0.3, -0.3, -0.5), du=(0.6, 0.0, 0.2), dv=(0.0, 0.6, 0.1)))
    elif kind == "soup":
        mesh = scenes.triangle_soup(1000, seed=seed, extent=1.0, size=0.15)
    elif kind == "blob":
        mesh = scenes.blob(rows=20, cols=30, radius=0.8)
    else:
        raise ValueError(kind)
    cam = scenes.look_at_camera((0.3, 0.2, -1.9), (0.0, 0.0, 0.0))
    return dict(mesh=mesh, tex=scenes.checker_texture(), lights=scenes.default_lights(), cam=cam)


def setup(obj, scene, num_lights=2):
    obj.set_mesh(scene["mesh"])
    obj.set_texture(scene["tex"])
    obj.set_lights(scene["lights"], num_lights)
    obj.set_camera(*scene["cam"])


def rays_equal_bits(a, b):
    """Field-by-field bitwise comparison of two ray dumps (padding ignored).
    Returns the per-ray mismatch mask."""
    bad = np.zeros(a.shape[0], dtype=bool)
    for f in FIELDS:
        x = np.ascontiguousarray(a[f]).view(np.uint32).reshape(a.shape[0], -1)
        y = np.ascontiguousarray(b[f]).view(np.uint32).reshape(b.shape[0], -1)
        diff = x != y
        if a[f].dtype.kind == "f":
            # a NaN equals a NaN: its sign and payload bits are not defined by IEEE arithmetic
            # (x86 produces 0xffc00000, the GPU 0x7fc00000)
            fa = np.ascontiguousarray(a[f]).reshape(a.shape[0], -1)
            fb = np.ascontiguousarray(b[f]).reshape(b.shape[0], -1)
            diff &= ~(np.isnan(fa) & np.isnan(fb))
        bad |= diff.any(axis=1)
    return bad


def describe_mismatch(a, b, bad, limit=5):
    idx = np.flatnonzero(bad)[:limit]
    lines = [f"{bad.sum()} of {bad.size} rays differ"]
    for i in idx:
        lines.append(f" ray {i}:")
        for f in FIELDS:
            lines.append(f"   {f}: {a[f][i]} vs {b[f][i]}")
    return "\n".join(lines)
