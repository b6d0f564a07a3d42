"""Occupancy of the walk's steps (a library built with -DRT_WALK_STATS=1 or =2; RT_HIP_LIB names it):
=1: lane-steps with a walk under way, triangle slots that were really tested;
=2: lane-steps in all (64 per wavefront step), triangle slots that held a triangle before the seen-table."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
g = rt.RayTracer(wl["width"], wl["height"], options={"sweeps_in_turn": 1})
setup_scene(g, wl)
g.render(wl["passes"]); g.sync(); g.counters()
g.render(wl["passes"]); g.sync()
c = (C.c_uint64 * 12)()
g._chk(g._lib.rt_get_counters(g._ctx, c))
print("closest rays", c[0], "candidates", c[4], "slow steps", c[6], "stats a", c[10], "stats b", c[11])
