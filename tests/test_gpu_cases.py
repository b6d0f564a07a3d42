"""GPU parity, continued: the rules and configurations that need scenes of their own.

  * ties  (trace.glsl:143, :169: strict `t < r.t`, so the first primitive in order keeps an equal t)
  * the 262 144-triangle, 64-view frame batch (the largest configuration of the benchmark)
  * the streaming band sweep that takes over when the band lists would outgrow 32-bit positions
  * a camera moved between raygen and intersect
  * a caller's stream and framebuffer bound to the context
  * two processes sharing one GPU, each rendering its row tiles

Same bar as test_gpu_parity.py: every field of every ray record and every pixel bit-equal to the oracle.
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from common import describe_mismatch, rays_equal_bits, setup, small_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32
MODES = [(2, "clustered"), (0, "filtered"), (1, "exact")]


def same(tag, gpu, cpu, pixels=True):
    got, want = gpu.read_rays(), cpu.read_rays()
    bad = rays_equal_bits(got, want)
    assert not bad.any(), tag + ": " + describe_mismatch(got, want, bad)
    if pixels:
        assert np.array_equal(gpu.read_pixels(), cpu.read_pixels()), tag + ": pixels differ"


# -- ties ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,name", MODES, ids=[m[1] for m in MODES])
@pytest.mark.parametrize("n,order", [(400, "blocks"), (3000, "blocks"), (1500, "interleaved")])
def test_identical_triangles_lowest_index_wins(rt, oracle_mod, mode, name, n, order):
    """Every triangle exists twice (k and k + n, or 2k and 2k + 1): every hit is a tie between two
    identical triangles and the reference's sequential loop keeps the first one."""
    from opengl_raytracer_amd import scenes
    base = scenes.concat_meshes(scenes.triangle_soup(n - 32, seed=11, extent=0.9, size=0.12),
                                scenes.quad_grid(4, 4, origin=(-1.2, -1.2, 1.1), du=(2.4, 0, 0), dv=(0, 2.4, 0)))
    assert base.shape[0] == 3 * n
    if order == "blocks":
        mesh = scenes.concat_meshes(base, base)
        first_copy = lambda ids: ids < n
    else:
        mesh = np.repeat(base.reshape(n, 3), 2, axis=0).reshape(-1)
        first_copy = lambda ids: ids % 2 == 0
    w, h = 96, 64
    g, c = rt.RayTracer(w, h, options={"sort_min_rays": 1024}), oracle_mod.Oracle(w, h, threads=8)
    with g:
        g.set_trace_mode(mode)
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.checker_texture())
            o.set_lights(scenes.default_lights(), 2)
            o.set_camera(*scenes.look_at_camera((0.2, 0.1, -2.4), (0, 0, 0)))
        g.raygen(); c.raygen()
        for p in range(3):
            g.intersect(); c.intersect()
            same(f"intersect {p}", g, c, pixels=False)
            ids = g.read_rays()["primitiveID"]
            tri = ids[(ids >= 0) & (ids < 2 * n)]
            assert (tri.size > 200 or p > 0) and first_copy(tri).all()
            g.color(); c.color()
            same(f"color {p}", g, c)


@pytest.mark.parametrize("mode,name", MODES, ids=[m[1] for m in MODES])
def test_triangle_and_sphere_at_equal_t(rt, oracle_mod, mode, name):
    """A triangle placed so that its computed t equals, bit for bit, the t of sphere 0 for the same ray:
    the triangle comes first in trace.glsl's order and the sphere's `t < r.t` is false (trace.glsl:169).
    The plane's position is found by trying neighbouring floats with the oracle's own arithmetic."""
    o = f32([0.0, -0.7, 0.0])
    d = f32([0.0, 0.0, 1.0])
    o_off = (o + d * f32(0.001)).astype(f32)                  # rayintersect_cs.glsl:12
    hit, ts = oracle_mod.ray_sphere(o_off, d, f32([0.0, -0.7, 2.0]), 0.5)
    assert hit and ts > 0
    z0 = f32(o_off[2] + ts)
    found = None
    for k in range(-64, 65):
        z = z0
        for _ in range(abs(k)):
            z = np.nextafter(z, f32(np.inf) if k > 0 else f32(-np.inf))
        tri = f32([[-0.4, -1.1, z], [0.5, -0.9, z], [0.0, 0.2, z]])
        ok, tt = oracle_mod.ray_vs_triangle(o_off, d, tri[0], tri[1], tri[2])
        if ok and tt.view(np.uint32) == ts.view(np.uint32):
            found = tri
            break
    assert found is not None, "no plane position reproduces the sphere's t"
    from opengl_raytracer_amd import scenes
    # the tying triangle sits in the middle of a mesh of far-away filler triangles
    filler = scenes.triangle_soup(300, seed=5, extent=0.3, size=0.05)
    filler["pos"] += f32([3.0, 3.0, -3.0])
    tie = scenes.pack_vertices(found, [[0, 0], [1, 0], [0, 1]])
    mesh = scenes.concat_meshes(filler[:450], tie, filler[450:])
    tie_id = 150
    w, h = 64, 2
    rays = np.zeros(w * h, dtype=rt.RAY_DTYPE)
    rays["origin"] = o
    rays["dir"] = d
    rays["primitiveID"] = -1
    rays["t"] = -1.0
    # neighbours of the tying ray: some hit the sphere first, some the triangle first
    rs = np.random.RandomState(2)
    rays["origin"][1:] += rs.uniform(-0.3, 0.3, (w * h - 1, 3)).astype(f32) * f32([1, 1, 0])
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=2)
    with g:
        g.set_trace_mode(mode)
        for x in (g, c):
            x.set_mesh(mesh)
            x.set_texture(scenes.checker_texture())
            x.set_lights(scenes.default_lights(), 1)
            x.set_camera((0, 0, -3), (0, 0, 1), (0, 1, 0), (1, 0, 0))
            x.write_rays(rays)
            x.intersect()
        same("intersect", g, c, pixels=False)
        got = g.read_rays()
        assert got["primitiveID"][0] == tie_id and got["t"][0].view(np.uint32) == ts.view(np.uint32)
        assert (got["primitiveID"] == 301).any()      # sphere 0 = numTris + 0 wins for other rays
        g.color(); c.color()
        same("color", g, c)


# -- 262 144 triangles, 64 views -------------------------------------------------------------------
def _views_workload():
    import bench
    return bench.workload("soup256k_64views_1080p")


def test_soup256k_views_frame_batch_against_oracle(rt, oracle_mod):
    """The benchmark's largest configuration at a size the oracle covers: the 262 144-triangle soup
    (seed 12346), views 0, 21 and 42 of the 64 cameras, 64 x 36 rays each, rendered through
    sharding.FrameBatch into a bound device framebuffer (what bench.py does), rays and pixels
    against the oracle."""
    import torch
    from opengl_raytracer_amd import scenes, sharding
    wl = _views_workload()
    assert wl["mesh"].shape[0] == 3 * 262144 and len(wl["cams"]) == 64
    w, h = 64, 36
    picks = [0, 21, 42]
    cams = [wl["cams"][k] for k in picks]
    dev = torch.device("cuda", 0)
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(wl["mesh"])
            o.set_texture(wl["tex"])
            o.set_lights(scenes.default_lights(), wl["lights"])
        job = sharding.FrameBatch(g, w, h, cams, dev)
        frames = job.render(wl["passes"])
        g.sync()
        torch.cuda.synchronize(dev)
        frames = frames.cpu().numpy()
        assert frames.shape == (3, h, w, 4)
        for j, cam in enumerate(cams):
            c.set_camera(*cam)
            c.render(wl["passes"])
            assert np.array_equal(frames[j], c.read_pixels()), f"view {picks[j]}: pixels differ"
        # the rays of the last view are still in the context
        same(f"view {picks[-1]}", g, c, pixels=False)
        assert np.isfinite(c.read_rays()["origin"]).all(axis=1).mean() > 0.15   # the soup fills about a quarter of the frame


def test_soup256k_1080p_modes_agree(rt):
    """The same configuration at its full size, through a size-independent property: one view at
    1920 x 1080, a subset of row tiles, clustered = filtered = all-pairs exact, bit for bit."""
    from opengl_raytracer_amd import scenes
    wl = _views_workload()
    w, h = wl["width"], wl["height"]
    with rt.RayTracer(w, h) as g:
        g.set_mesh(wl["mesh"])
        g.set_texture(wl["tex"])
        g.set_lights(scenes.default_lights(), 1)
        g.set_camera(*wl["cams"][5])
        g.render(2)
        full_rays, full_pix = g.read_rays().reshape(h, w), g.read_pixels()
        g.set_row_tiles(4, 3, 27)
        rows = g.row_map()
        out = {}
        for mode in (2, 0, 1):
            g.set_trace_mode(mode)
            g.render(2)
            out[mode] = (g.read_rays(), g.read_pixels())
        for mode in (0, 1):
            bad = rays_equal_bits(out[2][0], out[mode][0])
            assert not bad.any(), f"mode {mode}: " + describe_mismatch(out[2][0], out[mode][0], bad)
            assert np.array_equal(out[2][1], out[mode][1])
        assert not rays_equal_bits(out[2][0], full_rays[rows].reshape(-1)).any()
        assert np.array_equal(out[2][1], full_pix[rows])
        assert np.isfinite(full_rays["origin"]).all(axis=-1).mean() > 0.15


# -- band lists too long for 32-bit positions ----------------------------------------------------------
@pytest.mark.parametrize("sort_min", [2048, 1 << 30], ids=["sorted_rays", "unsorted_rays"])
def test_streaming_band_sweep_when_lists_are_not_built(rt, oracle_mod, sort_min):
    """RT_OPT_BAND_LIST_MAX = 0 is what a mesh whose band lists would pass 2^32 entries gets: no lists,
    and the bounce rays' band sweep streams the groups instead (over direction-sorted rays when the
    launch is large enough, inside the cluster kernel otherwise) -- with the direction cells of the
    shared origins still in use.  Against the oracle."""
    from opengl_raytracer_amd import scenes
    w, h = 320, 200
    mesh = scenes.concat_meshes(scenes.triangle_soup(7000, seed=21, extent=1.0, size=0.09),
                                scenes.quad_grid(12, 12, origin=(-1.4, -1.2, 1.3), du=(2.8, 0, 0), dv=(0, 2.4, 0.2)))
    opts = {"band_list_max": 0, "sort_min_rays": sort_min}
    g, c = rt.RayTracer(w, h, options=opts), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(64, 64))
            o.set_lights(scenes.default_lights(), 1)
            o.set_camera(*scenes.look_at_camera((0.1, 0.2, -2.5), (0, 0, 0)))
        g.kernel_log()
        g.render(3); c.render(3)
        same("no band lists", g, c)
        forms = [int(r["form"]) for r in g.kernel_log() if int(r["stage"]) == 1]
        assert forms[0] == 3 and 2 in forms[1:], forms


# -- the band lists split by the origin of the ray -----------------------------------------------------
@pytest.mark.parametrize("opts,built", [({"band_buckets": 1}, 1), ({"band_buckets": 2}, 2), ({"band_buckets": 16}, 16),
                                        ({"band_buckets": 16, "band_list_max": 40_000_000}, None)],
                         ids=["one_list_per_cell", "2x2", "16x16", "halved_to_fit"])
def test_band_lists_by_origin_bucket(rt, oracle_mod, monkeypatch, capfd, opts, built):
    """RT_OPT_BAND_BUCKETS: every bucket count renders the oracle's frame, rays that start off the mesh
    (on the scene's two spheres, which this camera sees) included; a limit on the lists' length halves
    the bucket count until they fit.  What was built is read from the library's report."""
    import re
    from opengl_raytracer_amd import scenes
    monkeypatch.setenv("RT_BAND_REPORT", "1")
    w, h = 320, 200
    mesh = scenes.concat_meshes(scenes.triangle_soup(7000, seed=23, extent=0.8, size=0.09),
                                scenes.quad_grid(12, 12, origin=(-1.4, -1.2, 2.8), du=(2.8, 0, 0), dv=(0, 2.4, 0.2)))
    g, c = rt.RayTracer(w, h, options=opts), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(64, 64))
            o.set_lights(scenes.default_lights(), 2)
            o.set_camera(*scenes.look_at_camera((-0.6, 0.1, -3.2), (-0.3, -0.1, 0.5)))
        g.render(4); c.render(4)
        same("band buckets %r" % (opts,), g, c)
        report = re.findall(r"band lists: (\d+) x \d+ buckets per cell, \d+ lists, (\d+) entries", capfd.readouterr().err)
        assert len(report) == 1, report
        got, entries = int(report[0][0]), int(report[0][1])
        if built is not None:
            assert got == built, report
        else:
            assert 1 <= got < 16 and entries <= opts["band_list_max"], report


def test_band_lists_shared_by_contexts_with_the_same_mesh(rt, oracle_mod, monkeypatch, capfd):
    """Contexts of one process that hold the same mesh on the same device (frames in flight) share one
    copy of the band lists: the first bounce pass of the first context builds them, the others' find them;
    a context with another mesh, or other options, builds its own; and once every holder is gone the next
    context builds them again.  Every frame equals the oracle's."""
    import re
    from opengl_raytracer_amd import scenes
    monkeypatch.setenv("RT_BAND_REPORT", "1")
    w, h = 256, 160
    mesh = scenes.triangle_soup(5000, seed=29, extent=0.9, size=0.08)
    other = scenes.triangle_soup(5000, seed=30, extent=0.9, size=0.08)

    def scene(o, m, eye):
        o.set_mesh(m)
        o.set_texture(scenes.noise_texture(32, 32))
        o.set_lights(scenes.default_lights(), 1)
        o.set_camera(*scenes.look_at_camera(eye, (0, 0, 0)))

    def reports():
        err = capfd.readouterr().err
        return len(re.findall(r"band lists: \d+ x \d+ buckets per cell, \d+ lists", err)), err.count("band lists: shared")

    eyes = [(0.2, 0.1, -2.6), (-0.4, 0.3, -2.4), (0.0, -0.5, -2.8)]
    c = oracle_mod.Oracle(w, h, threads=16)
    gs = [rt.RayTracer(w, h) for _ in eyes]
    try:
        for g, eye in zip(gs, eyes):
            scene(g, mesh, eye)
        for g in gs:                      # all three in flight before any is read
            g.render(3)
        for g, eye in zip(gs, eyes):
            scene(c, mesh, eye); c.render(3)
            same("shared band lists, eye %r" % (eye,), g, c)
        assert reports() == (1, 2)
        # another mesh in one of them: lists of its own; the other two keep theirs
        scene(gs[1], other, eyes[1]); gs[1].render(3)
        scene(c, other, eyes[1]); c.render(3)
        same("a context with another mesh", gs[1], c)
        assert reports() == (1, 0)
        gs[0].render(3); scene(c, mesh, eyes[0]); c.render(3)
        same("the first context after its neighbour changed meshes", gs[0], c)
        # other options: lists of its own
        with rt.RayTracer(w, h, options={"band_buckets": 2}) as g2:
            scene(g2, mesh, eyes[2]); g2.render(3)
            scene(c, mesh, eyes[2]); c.render(3)
            same("a context with other bucket counts", g2, c)
        assert reports() == (1, 0)
    finally:
        for g in gs:
            g.close()
    with rt.RayTracer(w, h) as g:         # every holder is gone: built again
        scene(g, mesh, eyes[0]); g.render(3)
        scene(c, mesh, eyes[0]); c.render(3)
        same("after the lists were freed", g, c)
    assert reports() == (1, 0)


@pytest.mark.parametrize("offset", [(30.0, -20.0, 15.0), (300.0, -200.0, 150.0)], ids=["near", "far"])
def test_mesh_far_from_the_world_origin(rt, oracle_mod, offset):
    """A mesh (and its camera and lights) translated away from the world's origin, where the scene's two
    fixed spheres stay: coordinates are coarser there, the scene's bounding sphere is mostly empty, the
    bands are wide and the band lists' origin buckets tile a box far from the centre they are measured
    from.  Three passes against the oracle."""
    from opengl_raytracer_amd import scenes
    w, h = 256, 160
    off = np.asarray(offset, dtype=np.float64)
    mesh = scenes.concat_meshes(scenes.triangle_soup(4000, seed=37, extent=0.9, size=0.09),
                                scenes.quad_grid(10, 10, origin=(-1.3, -1.1, 1.2), du=(2.6, 0, 0), dv=(0, 2.2, 0.3)))
    mesh = mesh.copy()
    mesh["pos"] = (mesh["pos"].astype(np.float64) + off).astype(np.float32)
    lights = scenes.default_lights()
    for i in range(2):
        lights[i]["pos"] = (np.asarray(lights[i]["pos"], dtype=np.float64) + off).astype(np.float32)
    cam = scenes.look_at_camera(tuple(off + np.array([0.2, 0.3, -2.7])), tuple(off))
    g, c = rt.RayTracer(w, h, options={"sort_min_rays": 4096}), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(32, 32))
            o.set_lights(lights, 2)
            o.set_camera(*cam)
        g.render(3); c.render(3)
        same("mesh at %r" % (offset,), g, c)


# -- the cells in space are built by the first bounce pass ---------------------------------------------
def test_cells_in_space_are_built_by_the_first_bounce_pass(rt, oracle_mod):
    """rt_set_mesh leaves the cells that bounce rays walk to the first pass that needs them: frames of one
    pass (camera and shadow rays only) never build them, a frame with bounces builds them once, another
    mesh on the same context starts over, and so does a context whose cell size was changed before its
    mesh was set again.  Every frame against the oracle."""
    from opengl_raytracer_amd import scenes
    w, h = 256, 160
    meshes = [scenes.concat_meshes(scenes.triangle_soup(9000, seed=71, extent=0.9, size=0.07),
                                   scenes.quad_grid(8, 8, origin=(-1.3, -1.0, -1.3), du=(2.6, 0, 0), dv=(0, 0, 2.6))),
              scenes.triangle_soup(12000, seed=72, extent=1.0, size=0.05)]
    cam = scenes.look_at_camera((0.3, 0.4, -2.8), (0.0, 0.0, 0.0))
    lights = scenes.default_lights()
    g, c = rt.RayTracer(w, h, options={"sort_min_rays": 4096}), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for k, mesh in enumerate(meshes):
            for o in (g, c):
                o.set_mesh(mesh)
                o.set_texture(scenes.noise_texture(32, 32))
                o.set_lights(lights, 2)
                o.set_camera(*cam)
            for passes in (1, 1, 4, 2, 1, 3):
                g.render(passes); c.render(passes)
                same("mesh %d, %d passes" % (k, passes), g, c)
        g.set_option("voxel_res", 20)   # (the mesh has to be set again after this option)
        g.set_mesh(meshes[1])
        g.render(3); c.render(3)
        same("coarser cells", g, c)


# -- a camera moved between raygen and intersect ------------------------------------------------------
@pytest.mark.parametrize("mode,name", MODES[:2], ids=[m[1] for m in MODES[:2]])
def test_camera_moved_after_raygen(rt, oracle_mod, mode, name):
    """raygen, then rt_set_camera with another position, then intersect and color: the rays still start
    at the old position (the reference's intersect reads each ray's own origin), while camera_pos --
    the depth plane's reference point and the view vector of the shading -- is the new one."""
    sc = small_scene("mixed")
    w, h = 96, 80
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=8)
    pos, d, u, r = sc["cam"]
    moved = (pos + f32([0.4, -0.3, 0.5]), d, u, r)
    with g:
        g.set_trace_mode(mode)
        setup(g, sc, 2); setup(c, sc, 2)
        g.render(1); c.render(1)            # lists of the first position exist
        for o in (g, c):
            o.raygen()
            o.set_camera(*moved)
            o.intersect()
        same("intersect after move", g, c, pixels=False)
        dg, dc = g.read_depth(), c.read_depth()
        assert ((dg.view(np.uint32) == dc.view(np.uint32)) | (np.isnan(dg) & np.isnan(dc))).all()
        for o in (g, c):
            o.color()
            o.intersect()
            o.color()
        same("second pass after move", g, c)
        # and a whole frame from the new position
        g.render(2); c.render(2)
        same("frame from the new position", g, c)


# -- a caller's stream and framebuffer ---------------------------------------------------------------
@pytest.mark.parametrize("in_turn", [0, 1], ids=["side_by_side", "in_turn"])
def test_bound_stream_and_framebuffer(rt, oracle_mod, in_turn):
    """rt_bind_stream with a torch stream and rt_bind_framebuffer with a torch tensor: the stage calls
    enqueue on the caller's stream (the library's internal side stream forks from it and joins it),
    pixels land in the caller's tensor; three passes of a scene whose bounce rays take the group
    sweeps, launches large enough to sort.  Then rt_unbind_stream / the own framebuffer again."""
    import torch
    from opengl_raytracer_amd import scenes
    w, h = 384, 256
    mesh = scenes.concat_meshes(scenes.triangle_soup(6000, seed=33, extent=1.0, size=0.1),
                                scenes.quad_grid(8, 8, origin=(-1.5, -1.3, 1.2), du=(3.0, 0, 0), dv=(0, 2.6, 0)))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    fb = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    g = rt.RayTracer(w, h, options={"sort_min_rays": 2048, "sweeps_in_turn": in_turn})
    c = oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(64, 64))
            o.set_lights(scenes.default_lights(), 2)
            o.set_camera(*scenes.look_at_camera((0.3, 0.1, -2.6), (0, 0, 0)))
        c.render(3)
        g.bind_stream(stream.cuda_stream)
        g.bind_framebuffer(fb.data_ptr())
        with torch.cuda.stream(stream):
            g.render(3)
            host = fb.cpu()                 # ordered after the render on the same stream
        stream.synchronize()
        assert np.array_equal(host.numpy(), c.read_pixels())
        same("bound stream", g, c, pixels=False)
        forms = [int(r["form"]) for r in g.kernel_log() if int(r["stage"]) == 1]
        assert forms[1:] == ([2, 4, 5] * 2 if in_turn else [2, 2]), forms
        # back to the context's own stream and framebuffer
        g.unbind_stream()
        g.bind_framebuffer(0)
        g.render(3)
        same("own stream again", g, c)
        assert np.array_equal(fb.cpu().numpy(), c.read_pixels())   # the caller's tensor was left alone


# -- two processes, one GPU, row tiles ---------------------------------------------------------------
def test_two_ranks_share_one_gpu_tiles(rt, tmp_path):
    """Two processes (gloo), each with its own context on cuda:0, render the interleaved row tiles of
    one frame through sharding.TileShardedFrame and gather them on rank 0: equal to the frame one
    process renders."""
    out = str(tmp_path / "frame.npy")
    worker = os.path.join(ROOT, "tests", "mp_tile_worker.py")
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))      # a port that is free now
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    # (each rank writes to a file of its own: a pipe nobody reads would block a rank that prints a lot)
    logs = [open(str(tmp_path / f"rank{r}.log"), "w+") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, out], env=dict(env, RANK=str(r)), stdout=logs[r],
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        deadline = time.monotonic() + 110
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.monotonic()))
    finally:
        for p in procs:        # a rank that hangs must not keep the GPU after the test has failed
            if p.poll() is None:
                p.kill()
                p.wait()
    texts = []
    for f in logs:
        f.seek(0)
        texts.append(f.read())
        f.close()
    assert all(p.returncode == 0 for p in procs), "\n".join(texts)
    got = np.load(out)
    import mp_tile_worker
    with rt.RayTracer(mp_tile_worker.W, mp_tile_worker.H) as g:
        mp_tile_worker.scene_setup(g)
        g.render(mp_tile_worker.PASSES)
        assert np.array_equal(got, g.read_pixels())


def test_torch_initialises_its_gpu_after_the_library_was_used():
    """A fresh process that renders through the library BEFORE torch is imported: torch must still
    find the GPU afterwards (one copy of the HIP runtime in the process, not two)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import opengl_raytracer_amd as rt\n"
        "from opengl_raytracer_amd import scenes\n"
        "assert 'torch' not in sys.modules\n"
        "with rt.RayTracer(64, 64) as g:\n"
        "    g.set_mesh(scenes.triangle_soup(100, seed=1)); g.set_lights(scenes.default_lights(), 1)\n"
        "    g.set_camera(*scenes.default_camera()); g.render(2); px = g.read_pixels()\n"
        "import torch\n"
        "t = torch.from_numpy(px).to('cuda:0')\n"
        "assert int(t.sum().item()) == int(px.sum())\n"
        "print('ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


@pytest.mark.parametrize("stage", ["intersect", "color"])
def test_stage_that_fails_after_it_began_leaves_timers_and_context_usable(rt, oracle_mod, stage):
    """A stage call that fails between its two timing events: rt_stage_ms still answers (the pair is
    closed on the way out), the kernel log still reads, and the next frame is bit-equal to the oracle's."""
    sc = small_scene("mixed")
    g, c = rt.RayTracer(96, 64), oracle_mod.Oracle(96, 64, threads=8)
    with g:
        for o in (g, c):
            setup(o, sc, 2)
        g.render(2)
        assert all(m > 0 for m in g.stage_ms())
        g.kernel_log()
        g.raygen()
        if stage == "color":
            g.intersect()
        g.set_option("fail_next_stage", 1)
        with pytest.raises(rt.RtError) as e:
            getattr(g, stage)()
        assert e.value.code == -2 and "RT_OPT_FAIL_NEXT_STAGE" in str(e.value)
        ms = g.stage_ms()                     # does not raise; the failed call's pair is closed
        assert all(np.isfinite(m) and m >= 0 for m in ms) and ms[0] > 0
        g.kernel_log()
        getattr(g, stage)()                   # the option is back to 0
        g.render(2); c.render(2)
        same("after a failed stage call", g, c)
        assert all(m > 0 for m in g.stage_ms())


def _sparse_scene():
    """A small soup in one corner of a large frame: most chunks of the frame never file a ray."""
    from opengl_raytracer_amd import scenes
    mesh = scenes.triangle_soup(3000, seed=77, extent=0.35, size=0.06)
    mesh["pos"][:, 0] += 0.9
    mesh["pos"][:, 1] -= 0.6
    return mesh, scenes.look_at_camera((0.0, 0.0, -3.0), (0.0, 0.0, 0.0))


def test_rays_rewritten_between_passes_are_all_visited(rt, oracle_mod):
    """Compaction skips the parts of the frame that filed nothing in the pass before; rays written by the
    caller in between (dead pixels revived, aimed at the mesh) must be found again."""
    from opengl_raytracer_amd import scenes
    mesh, cam = _sparse_scene()
    w, h = 512, 256
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(64, 64))
            o.set_lights(scenes.default_lights(), 2)
            o.set_camera(*cam)
        g.render(2); c.render(2)
        same("two passes", g, c)
        rays = c.read_rays()
        dead = rays["primitiveID"] == -2
        assert dead.mean() > 0.6                      # most of the frame
        # every dead ray comes back to life, from the camera towards the soup's corner
        rs = np.random.RandomState(5)
        k = int(dead.sum())
        rays["origin"][dead] = np.float32([0.0, 0.0, -3.0])
        d = np.float32([0.9, -0.6, 3.0]) + rs.uniform(-0.3, 0.3, size=(k, 3)).astype(np.float32)
        rays["dir"][dead] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        rays["primitiveID"][dead] = -1
        rays["t"][dead] = np.float32(np.inf)
        g.write_rays(rays); c.write_rays(rays)
        for o in (g, c):
            o.intersect(); o.color(); o.intersect(); o.color()
        same("after the rewrite", g, c)
        assert (c.read_rays()["primitiveID"][dead] >= 0).mean() > 0.2   # and many of them hit


def test_framebuffer_bound_between_passes_gets_every_pixel(rt, oracle_mod):
    """A framebuffer bound after the first pass has not seen the grey pixels that pass wrote: the next
    rt_color must write the whole frame into it, skipped parts included."""
    import torch
    from opengl_raytracer_amd import scenes
    mesh, cam = _sparse_scene()
    w, h = 512, 256
    dev = torch.device("cuda", 0)
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)
    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(64, 64))
            o.set_lights(scenes.default_lights(), 1)
            o.set_camera(*cam)
        g.render(2); c.render(2)
        late = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        g.bind_framebuffer(late.data_ptr())
        for o in (g, c):
            o.intersect(); o.color()
        g.sync()
        assert np.array_equal(late.cpu().numpy(), c.read_pixels())
        same("rays", g, c, pixels=False)
        g.bind_framebuffer(0)


# -- frames in flight ---------------------------------------------------------------------------------
def test_views_in_flight_on_two_contexts_against_oracle(rt, oracle_mod):
    """Five views dealt to two contexts in turn, each context on a stream of its own and writing
    into its slice of one device tensor with no wait between views (sharding.FrameBatch with a list
    of renderers): every frame equals the oracle's, whatever overlapped with it."""
    import torch
    from opengl_raytracer_amd import scenes, sharding
    sc = small_scene("mixed")
    w, h, passes = 160, 96, 3
    cams = [scenes.look_at_camera((2.6 * np.cos(a), 0.4, 2.6 * np.sin(a)), (0, 0, 0)) for a in (0.3, 1.1, 2.0, 3.3, 4.9)]
    dev = torch.device("cuda", 0)
    gs = [rt.RayTracer(w, h, options={"sort_min_rays": 1024}) for _ in range(2)]
    c = oracle_mod.Oracle(w, h, threads=8)
    try:
        for o in gs + [c]:
            setup(o, sc, 2)
        job = sharding.FrameBatch(gs, w, h, cams, dev)
        for _ in range(2):   # the second batch finds the contexts' lists and counters as the first left them
            frames = job.render(passes)
            job.wait()
            torch.cuda.synchronize(dev)
            frames = frames.cpu().numpy()
            for j, cam in enumerate(cams):
                c.set_camera(*cam)
                c.render(passes)
                assert np.array_equal(frames[j], c.read_pixels()), f"view {j}: pixels differ"
        # view 4 was the third view of context 0: its rays are still there
        same("last view of context 0", gs[0], c, pixels=False)
    finally:
        for g in gs:
            g.close()


def test_frames_in_flight_tiles_on_three_contexts(rt, oracle_mod):
    """sharding.TileShardedFrame with three renderers: consecutive frames go to the contexts in turn
    (streams of their own, framebuffers bound); every frame returned equals the oracle's frame."""
    import torch
    from opengl_raytracer_amd import sharding
    sc = small_scene("mixed")
    w, h, passes = 128, 80, 3
    dev = torch.device("cuda", 0)
    gs = [rt.RayTracer(w, h, options={"sort_min_rays": 1024}) for _ in range(3)]
    c = oracle_mod.Oracle(w, h, threads=8)
    try:
        for o in gs + [c]:
            setup(o, sc, 2)
        c.render(passes)
        want = c.read_pixels()
        job = sharding.TileShardedFrame(gs, w, h, dev)
        frames = [job.render(passes) for _ in range(7)]
        job.wait()
        torch.cuda.synchronize(dev)
        for k in (4, 5, 6):   # the last frame of each context
            assert np.array_equal(frames[k].cpu().numpy(), want), f"frame {k}: pixels differ"
        for g in gs:
            same("context", g, c, pixels=False)
    finally:
        for g in gs:
            g.close()


def test_lights_move_between_frames_in_flight(rt, oracle_mod):
    """The reference moves its lights every frame (main.cpp:449).  Three contexts take the frames in turn on
    streams of their own; before each frame its context gets the lights of that frame, so its lists of the
    light's direction cells are built again -- without a wait from their second build on.  Every frame equals
    the oracle's frame with the same lights."""
    import torch
    from opengl_raytracer_amd import scenes, sharding
    sc = small_scene("mixed")
    w, h, passes = 128, 80, 3
    dev = torch.device("cuda", 0)
    gs = [rt.RayTracer(w, h, options={"sort_min_rays": 1024}) for _ in range(3)]
    c = oracle_mod.Oracle(w, h, threads=8)
    try:
        for o in gs + [c]:
            setup(o, sc, 2)
        job = sharding.TileShardedFrame(gs, w, h, dev)
        angles = [0.05 * k for k in range(10)]
        frames = []
        for k, a in enumerate(angles):
            gs[k % 3].set_lights(scenes.default_lights(angle_offset=a), 2)
            f = job.render(passes)
            with job.streams.on(k % 3):   # a context's tensor is written again three frames on: keep a copy,
                frames.append(f.clone())  # made on the context's stream, behind the frame
        job.wait()
        torch.cuda.synchronize(dev)
        for k, a in enumerate(angles):
            c.set_lights(scenes.default_lights(angle_offset=a), 2)
            c.render(passes)
            assert np.array_equal(frames[k].cpu().numpy(), c.read_pixels()), f"frame {k}: pixels differ"
    finally:
        for g in gs:
            g.close()


def test_list_build_in_flight_runs_out_of_room(rt, oracle_mod, monkeypatch, capfd):
    """A light far from the mesh sees every triangle in a cell or two; moved into the mesh it sees each over
    thousands of cells.  The second build of the light's lists does not wait for its pair count, so it runs in
    arrays sized for the first: the slots it has no room for go on the list every ray reads, and the frame
    still equals the oracle's.  The third build finds the count of the second and waits once for larger arrays."""
    import re
    from opengl_raytracer_amd import scenes
    monkeypatch.setenv("RT_LIST_REPORT", "1")
    w, h, passes = 160, 100, 2
    mesh = scenes.triangle_soup(3000, seed=41, extent=0.9, size=0.12)
    cam = scenes.look_at_camera((0.0, 0.0, -60.0), (0, 0, 0))   # far away too: few pairs, little room
    g, c = rt.RayTracer(w, h), oracle_mod.Oracle(w, h, threads=16)

    def lights_at(pos):
        l = scenes.default_lights()
        l[0]["pos"] = pos
        return l

    with g:
        for o in (g, c):
            o.set_mesh(mesh)
            o.set_texture(scenes.noise_texture(32, 32))
            o.set_camera(*cam)
        seen = []
        for pos in ((0.0, 80.0, -20.0), (0.05, 0.1, -0.02), (0.06, 0.1, -0.02), (0.07, 0.1, -0.02)):
            for o in (g, c):
                o.set_lights(lights_at(pos), 1)
                o.render(passes)
            same("light at %r" % (pos,), g, c)
            for m in re.finditer(r"direction cells of origin 1: ([a-z ]+); last build counted (\d+) pairs, room for (\d+)",
                                 capfd.readouterr().err):
                seen.append((m.group(1), int(m.group(2)), int(m.group(3))))
    # first build waits; the second runs in flight; the third finds that the second counted more pairs than
    # there was room for, waits and enlarges; the fourth is in flight again, with room to spare
    assert [s[0] for s in seen] == ["waits for the count", "in flight", "waits for the count", "in flight"], seen
    assert seen[2][1] > seen[2][2], seen
    assert seen[3][1] <= seen[3][2], seen


# -- the benchmark program's collective path, rehearsed on one rank ----------------------------------
@pytest.mark.parametrize("workload", ["soup64k_4k_p6", "soup256k_64views_1080p"])
def test_bench_collectives_rehearsed_on_one_rank(workload):
    """bench.py with a process group of ONE rank over RCCL: every frame (or batch of views) goes
    through the gather, on the streams of the contexts in flight, and the timing through barrier
    and all-reduce, exactly as with N ranks.  The gathered pixels equal those of a plain run with
    one frame at a time (the checksum bench.py prints)."""
    import json

    def run(*flags):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-kernel-rooflines", *flags]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    plain = run("--frames-in-flight", "1")
    rehearsed = run("--rehearse-collectives")
    assert rehearsed["config"]["rehearse_collectives"] and rehearsed["config"]["frames_in_flight"] == 4
    assert plain["config"]["frame_crc32"] == rehearsed["config"]["frame_crc32"]
    assert rehearsed["config"]["rays_per_step"] == plain["config"]["rays_per_step"]
