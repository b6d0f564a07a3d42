import sys, os
sys.path.insert(0, os.getcwd())
import opengl_raytracer_amd as rt
from bench import workload, setup_scene
wl = workload("soup64k_4k_p6")
g = rt.RayTracer(wl["width"], wl["height"]); setup_scene(g, wl)
g.render(2); g.sync(); g.counters()
g.raygen()
for p in range(3):
    g.intersect(); c = g.counters(); k = list(c.values())
    if p == 0: print("pass0 intersect(FORM1 closest): waves?  tile %.3g longest wave %.3g slow %.3g total %.3g cycles" % (k[4], k[5], k[6], k[7]))
    g.color(); c = g.counters(); k = list(c.values())
    print("pass%d color (FORM1 anyhit): tile %.3g longest wave %.3g slow(incl drain) %.3g total %.3g cycles; shadow rays %d" % (p, k[4], k[5], k[6], k[7], k[1]))
