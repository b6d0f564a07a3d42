"""The C oracle against a second restatement of the shaders (tests/numpy_shaders.py, written from the
GLSL text separately), bit for bit: every field of every ray after every stage, and every pixel.
Neither is the reference -- its shaders cannot run here -- but two independent readings of the same
text agreeing on scenes that exercise the delicate statements (the origin mutated per light, the kill
rule, the re-shade of dead rays, out-of-range vertex and texel reads, misses that leave infinities and NaN behind)
leaves little room for a slip in either."""
import numpy as np
import pytest

from common import rays_equal_bits, describe_mismatch
import numpy_shaders as ns


def frame_rays(fr, dtype):
    n = fr.w * fr.h
    out = np.zeros(n, dtype=dtype)
    for name, src in (("origin", fr.origin), ("dir", fr.dir), ("color", fr.color), ("n", fr.n)):
        out[name] = np.stack(src, axis=1)
    out["t"] = fr.t
    out["primitiveID"] = fr.id
    return out


def scene(kind):
    from opengl_raytracer_amd import scenes
    if kind == "mixed":
        mesh = scenes.concat_meshes(scenes.quad_grid(2, 2, origin=(-1.0, -1.0, 1.0)),
                                    scenes.triangle_soup(40, seed=5, extent=0.8, size=0.3),
                                    scenes.quad_grid(1, 2, origin=(-0.3, -0.3, -0.5), du=(0.6, 0.0, 0.2), dv=(0.0, 0.6, 0.1)))
        cam = scenes.look_at_camera((0.3, 0.2, -1.9), (0.0, 0.0, 0.0))
    elif kind == "spheres_only":
        mesh = scenes.triangle_soup(0, seed=1)
        cam = scenes.look_at_camera((0.5, 0.1, -2.5), (-0.8, -0.3, 1.0))
    elif kind == "few_triangles":   # fewer than 13: nothing is ever killed, every hit stays a mirror
        mesh = scenes.quad_grid(2, 2, origin=(-1.5, -1.5, 2.5), du=(3.0, 0, 0), dv=(0, 3.0, 0))
        cam = scenes.default_camera()
    else:
        raise ValueError(kind)
    return mesh, cam


@pytest.mark.parametrize("kind,lights,passes", [("mixed", 2, 3), ("mixed", 0, 2), ("spheres_only", 1, 2),
                                                ("few_triangles", 3, 3)])
def test_numpy_restatement_equals_c_oracle(oracle_mod, kind, lights, passes):
    from opengl_raytracer_amd import scenes
    w, h = 40, 28
    mesh, cam = scene(kind)
    tex = scenes.noise_texture(16, 8)
    ls = scenes.default_lights()
    c = oracle_mod.Oracle(w, h, threads=4)
    c.set_mesh(mesh)
    c.set_texture(tex)
    c.set_lights(ls, lights)
    c.set_camera(*cam)
    fr = ns.Frame(w, h)
    verts = np.ascontiguousarray(mesh, dtype=oracle_mod.VERTEX_DTYPE)

    def same(tag):
        got, want = frame_rays(fr, oracle_mod.RAY_DTYPE), c.read_rays()
        bad = rays_equal_bits(got, want)
        assert not bad.any(), f"{kind} {tag}: " + describe_mismatch(got, want, bad)

    fr.raygen(cam)
    c.raygen()
    same("raygen")
    for p in range(passes):
        fr.intersect(verts)
        c.intersect()
        same(f"intersect {p}")
        fr.color_stage(verts, ls, lights, cam[0], tex)
        c.color()
        same(f"color {p}")
        assert np.array_equal(fr.pixels, c.read_pixels()), f"{kind} pass {p}: pixels differ"
    ids = c.read_rays()["primitiveID"]
    if kind == "mixed" and lights:
        assert (ids == -2).any() and (ids >= 0).any()      # killed or missed, and still alive
    assert not np.isfinite(c.read_rays()["origin"]).all()   # misses left their infinities behind
