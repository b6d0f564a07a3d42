"""Randomised comparison of the clustered (default) traversal with the all-pairs reference arithmetic
on the GPU: every ray record and pixel must agree bit for bit.  Scenes vary triangle size, mix
axis-aligned walls and floors (coplanar with many rays), put lights and cameras inside the geometry,
and use frames large enough for the sorted-ray kernels.
usage: python tools/fuzz_modes.py [first_seed] [count] [clustered|filtered]
The third argument picks the mode compared with the all-pairs arithmetic (default: clustered;
filtered = every pair through the per-pair filters, shared-origin and moment form)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import opengl_raytracer_amd as rt
from opengl_raytracer_amd import scenes
from common import describe_mismatch, rays_equal_bits


# FUZZ_OPTS="band_buckets=2,grid_bits=8": further options of the context (rt_set_option)
EXTRA_OPTIONS = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("FUZZ_OPTS", "").split(",") if kv))


def scene(seed):
    rs = np.random.RandomState(seed)
    parts = []
    n = int(rs.choice([3000, 9000, 20000, 40000]))
    size = float(rs.choice([0.004, 0.02, 0.05, 0.15, 0.5]))
    parts.append(scenes.triangle_soup(n, seed=seed, extent=1.0, size=size))
    if rs.rand() < 0.6:  # a floor and a wall of grid triangles (coplanar with shadow and bounce rays)
        k = int(rs.choice([4, 16, 40]))
        parts.append(scenes.quad_grid(k, k, origin=(-1.5, -1.0, -1.5), du=(3.0, 0, 0), dv=(0, 0, 3.0)))
        parts.append(scenes.quad_grid(k, k, origin=(-1.5, -1.0, 1.5), du=(3.0, 0, 0), dv=(0, 3.0, 0)))
    if rs.rand() < 0.3:
        parts.append(scenes.blob(rows=30, cols=40, radius=0.5))
    mesh = scenes.concat_meshes(*parts)
    eye = rs.uniform(-1.2, 1.2, 3) if rs.rand() < 0.5 else rs.uniform(-1, 1, 3) * 0.3 + np.array([0, 0, -3.0])
    cam = scenes.look_at_camera(tuple(eye), tuple(rs.uniform(-0.3, 0.3, 3)))
    nl = int(rs.choice([1, 2, 3]))
    lights = scenes.default_lights()
    for i in range(nl):
        lights[i]["pos"] = rs.uniform(-1.4, 1.4, 3) if rs.rand() < 0.7 else rs.uniform(-6, 6, 3)
    w, h = [(640, 360), (512, 512), (1024, 400)][int(rs.randint(3))]
    return mesh, cam, lights, nl, w, h, int(rs.choice([2, 3, 4]))


def run(first, count, log=print, mode=None):
    """Returns the number of seeds whose two modes disagree."""
    bad_total = 0
    mode = rt.RT_TRACE_CLUSTERED if mode is None else mode
    for seed in range(first, first + count):
        mesh, cam, lights, nl, w, h, passes = scene(seed)
        out = {}
        with rt.RayTracer(w, h, options=dict({"sort_min_rays": 16384, "cluster_min_tris": 0}, **EXTRA_OPTIONS)) as g:
            g.set_mesh(mesh); g.set_texture(scenes.noise_texture(64, 64)); g.set_lights(lights, nl); g.set_camera(*cam)
            for m in (mode, rt.RT_TRACE_EXACT):
                g.set_trace_mode(m)
                g.raygen()
                stages = []
                for p in range(passes):
                    g.intersect(); stages.append(g.read_rays())
                    g.color(); stages.append(g.read_rays())
                out[m] = (stages, g.read_pixels())
        a, b = out[mode], out[rt.RT_TRACE_EXACT]
        msg = "ok"
        for i, (x, y) in enumerate(zip(a[0], b[0])):
            bad = rays_equal_bits(x, y)
            if bad.any():
                msg = "MISMATCH after stage %d: %s" % (i, describe_mismatch(x, y, bad)); bad_total += 1
                break
        if msg == "ok" and not np.array_equal(a[1], b[1]):
            msg = "PIXEL MISMATCH"; bad_total += 1
        live = int((a[0][-1]["primitiveID"] >= 0).sum())
        log("seed %d: %d triangles, %dx%d, %d passes, %d lights, %d live rays at the end: %s" %
              (seed, mesh.shape[0] // 3, w, h, passes, nl, live, msg))

    return bad_total


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    which = sys.argv[3] if len(sys.argv) > 3 else "clustered"
    sys.exit(1 if run(first, count, log=lambda m: print(m, flush=True),
                      mode={"clustered": rt.RT_TRACE_CLUSTERED, "filtered": rt.RT_TRACE_FILTERED}[which]) else 0)
