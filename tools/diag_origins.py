"""Where the bounce rays of the default bench frame start: on the mesh (inside its bounding box, where the
band sweep's origin buckets are) or off it (on the scene's two spheres; those read whole cell lists)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
g = rt.RayTracer(wl["width"], wl["height"])
setup_scene(g, wl)
mesh = wl["mesh"]
g.raygen()
for p in range(wl["passes"] - 1):
    g.intersect(); g.color()
    r = g.read_rays()
    live = r["primitiveID"] != -2          # rays that bounce on
    o = r["origin"][live]
    inside = np.ones(len(o), bool)
    if mesh is not None:
        lo, hi = mesh["pos"].min(0) - 0.01, mesh["pos"].max(0) + 0.01
        inside = ((o >= lo) & (o <= hi)).all(1)
    print("after pass %d: %d rays bounce on, %.1f %% of them start outside the mesh's box" %
          (p, live.sum(), 100.0 * (1 - inside.mean()) if len(o) else 0.0), flush=True)
g.close()
