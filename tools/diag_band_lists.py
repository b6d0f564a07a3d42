"""Memory and build time of the band sweep's lists against the number of origin buckets per cell
(usage: BUCKETS=1,8,16 python tools/diag_band_lists.py [workload])."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
for G in [int(x) for x in os.environ.get("BUCKETS", "1,8,16").split(",")]:
    g = rt.RayTracer(wl["width"], wl["height"], options={"band_buckets": G})
    setup_scene(g, wl)
    g.render(1); g.sync()                     # camera and shadow rays only: no lists yet
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    g.render(wl["passes"]); g.sync()          # the first bounce pass builds them
    t1 = time.perf_counter()
    free1 = torch.cuda.mem_get_info()[0]
    g.render(wl["passes"]); g.sync()
    t2 = time.perf_counter()
    for _ in range(5):
        g.render(wl["passes"])
    g.sync()
    t3 = time.perf_counter()
    print("buckets %2d: lists %.2f GB, first frame %.1f ms, next %.2f ms, then %.2f ms per frame"
          % (G, (free0 - free1) / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) / 5 * 1e3), flush=True)
    g.close()
