"""GPU parity on the reference's own scenes (meshes rebuilt from its .blend sources, its texture,
its camera and light rig, all loaded through the host library), and the headless frame-loop
program end to end.  Same bar as test_gpu_parity.py: bit-exact rays and pixels against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from common import describe_mismatch, rays_equal_bits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGRAM = os.path.join(ROOT, "opengl_raytracer_amd", "rt_raytracer")


@pytest.fixture(scope="module")
def host():
    from opengl_raytracer_amd import host as h
    h.load()
    return h


def scene(host, name):
    from opengl_raytracer_amd import scenes
    sword = lambda: host.Mesh().LoadFromObjFile(os.path.join(ROOT, "assets", "sword") + os.sep, "sword_fixed.obj").packed(0)
    bunny = lambda: host.Mesh().LoadFromObjFile(os.path.join(ROOT, "assets", "bunny") + os.sep, "bunny.obj").packed(0)
    tex = host.load_image(os.path.join(ROOT, "assets", "sword", "sword.png"), invert_y=True)
    d, r, u = host.camera_basis(-82.44, 24.82)
    cam = (np.float32([2.77904, 0.594951, -1.95046]), d, u, r)  # main.cpp:402-406
    if name == "sword":
        return sword(), tex, cam
    if name == "bunny":
        return bunny(), tex, scenes.look_at_camera((0.8, 1.5, -1.9), (0.0, 1.0, -0.3))
    return scenes.concat_meshes(sword(), bunny()), tex, cam


@pytest.mark.parametrize("name,w,h,passes,nl", [
    ("sword", 200, 200, 3, 2),       # the reference's default scene (main.cpp:393-414), two bounces
    ("bunny", 128, 72, 1, 1),        # primary + shadow rays on the 28 576-triangle bunny
    ("sword+bunny", 96, 54, 3, 1),   # the combined scene
])
def test_reference_scenes_against_oracle(rt, oracle_mod, host, name, w, h, passes, nl):
    mesh, tex, cam = scene(host, name)
    lights = host.default_lights()
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(tex)
            o.set_lights(lights, nl)
            o.set_camera(*cam)
            o.render(passes)
        got, want = g.read_rays(), c.read_rays()
        bad = rays_equal_bits(got, want)
        assert not bad.any(), name + ": " + describe_mismatch(got, want, bad)
        assert np.array_equal(g.read_pixels(), c.read_pixels())
        grey = np.array([26, 26, 26], dtype=np.uint8)  # the background every missed ray keeps
        assert (g.read_pixels()[..., :3] != grey).any(-1).mean() > 0.1  # the scene is in the frame


def test_small_mesh_default_threshold(rt, oracle_mod, host):
    """With the library's own threshold (1024 triangles; the other tests run with 0) the bounce rays of
    the 165-triangle sword go through the per-pair filter (form 0 in the kernel log): same bits as the oracle."""
    mesh, tex, cam = scene(host, "sword")
    w, h = 160, 160
    g, c = rt.RayTracer(w, h, options={"cluster_min_tris": 1024}), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh); o.set_texture(tex); o.set_lights(host.default_lights(), 2); o.set_camera(*cam)
        g.kernel_log()
        g.render(2); c.render(2)
        got, want = g.read_rays(), c.read_rays()
        bad = rays_equal_bits(got, want)
        assert not bad.any(), describe_mismatch(got, want, bad)
        assert np.array_equal(g.read_pixels(), c.read_pixels())
        forms = [int(r["form"]) for r in g.kernel_log() if int(r["stage"]) == 1]
        assert forms == [3, 0], forms   # primary rays: direction cells; bounce rays: per-pair filter


def test_frame_loop_program(rt, oracle_mod, host, tmp_path):
    """rt_raytracer (the reference's RunRaytracer over the C ABI): its frame file equals the frame
    the Python mirror and the oracle produce from the same inputs; its CSV has the reference's
    nine integer columns (main.cpp:209-219); the marker overlay only adds yellow squares."""
    if not os.path.exists(PROGRAM):
        pytest.fail(f"{PROGRAM} is missing: build it with make -C opengl_raytracer_amd/csrc/demo")
    w = h = 160
    out, csv = str(tmp_path / "frame.png"), str(tmp_path / "results.csv")
    base = [PROGRAM, "--assets", os.path.join(ROOT, "assets"), "--scene", "sword", "--width", str(w), "--height", str(h),
            "--tgs", "32", "--passes", "2", "--lights", "2"]
    p = subprocess.run(base + ["--frames", "1", "--out", out, "--csv", csv], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    frame = host.load_image(out, invert_y=True)

    mesh, tex, cam = scene(host, "sword")
    lights = host.default_lights()
    host.set_light_positions(lights, 10, host.light_angle_offset(0.0))
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(tex)
            o.set_lights(lights, 2)
            o.set_camera(*cam)
            o.render(2)
        assert np.array_equal(g.read_pixels(), c.read_pixels())
        assert np.array_equal(frame, g.read_pixels())

    # three more frames: the lights move (main.cpp:449), the CSV grows by one line per frame
    out2 = str(tmp_path / "frame2.tga")
    p = subprocess.run(base + ["--frames", "3", "--out", out2, "--markers", "--csv", csv, "--time-step", "0.5"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(csv).read().splitlines()
    assert len(lines) == 4
    for line in lines:
        cols = [int(x) for x in line.split(",")]
        assert cols[:6] == [32, w, h, 2, 2, 165] and len(cols) == 9 and all(x >= 0 for x in cols[6:])
    marked = host.load_image(out2, invert_y=True)
    yellow = (marked == np.array([255, 255, 0, 255], dtype=np.uint8)).all(-1)
    host.set_light_positions(lights, 10, host.light_angle_offset(1.0))
    c.set_lights(lights, 2)
    c.render(2)
    want = c.read_pixels()
    assert np.array_equal(marked[~yellow], want[~yellow])
