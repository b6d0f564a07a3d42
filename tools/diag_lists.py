"""How the direction-cell lists of a workload's camera and light are built (diagnostic): the library's own
report (RT_LIST_REPORT) for a few frames with the lights moving, and the time of those frames.
usage: python tools/diag_lists.py [workload]"""
import math
import os
import sys
import time

os.environ["RT_LIST_REPORT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opengl_raytracer_amd as rt
from opengl_raytracer_amd import scenes
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "swordbunny_4k_p4")
g = rt.RayTracer(wl["width"], wl["height"])
setup_scene(g, wl)
g.render(wl["passes"]); g.sync()
for k in range(4):
    g.set_lights(scenes.default_lights(angle_offset=(2 * math.pi / 10) * math.sin((k + 1) / 60.0)), wl["lights"])
    t0 = time.perf_counter()
    g.render(wl["passes"]); g.sync()
    print("frame with moved lights: %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
g.render(wl["passes"]); g.sync()
print("frame with lights at rest: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
