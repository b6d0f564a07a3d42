"""Frames in flight on one GPU: K contexts (each with its own stream) take the frames of a sequence in
turn, so that the dependent kernel chains of consecutive frames overlap.  Prints the time per frame for
K = 1, 2, 3, ... for the whole frame and for one rank's share of a tile split
(usage: WORLD=8 RANK=0 INFLIGHT=1,2,3 python tools/diag_inflight.py [workload])."""
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
world = int(os.environ.get("WORLD", "8"))
rank = int(os.environ.get("RANK", "0"))
ks = [int(x) for x in os.environ.get("INFLIGHT", "1,2,3").split(",")]
frames = int(os.environ.get("FRAMES", "12"))
threads = int(os.environ.get("THREADS", "0"))

# OPTS="sweeps_in_turn=1": options of the contexts (rt_set_option)
opts = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv))
ctxs = []
for _ in range(max(ks)):
    g = rt.RayTracer(wl["width"], wl["height"], options=opts)
    setup_scene(g, wl)
    ctxs.append(g)


def run(k, tile_rows, r, w):
    use = ctxs[:k]
    for g in use:
        g.set_row_tiles(tile_rows, r, w)
        g.render(wl["passes"])
    for g in use:
        g.sync()
    t0 = time.perf_counter()
    if threads:  # one host thread per context: the enqueue calls of different frames run side by side
        def work(i):
            for f in range(i, frames, k):
                use[i].render(wl["passes"])
            use[i].sync()
        ts = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        host = 0.0
    else:
        for f in range(frames):
            use[f % k].render(wl["passes"])
        host = (time.perf_counter() - t0) / frames * 1e3
        for g in use:
            g.sync()
    return (time.perf_counter() - t0) / frames * 1e3, host


for k in ks:
    print("frames in flight %d: whole frame %.3f ms per frame (host enqueue %.3f)" % ((k,) + run(k, wl["height"], 0, 1)), flush=True)
for k in ks:
    print("frames in flight %d: share of rank %d of %d: %.3f ms per frame (host enqueue %.3f)"
          % ((k, rank, world) + run(k, 15, rank, world)), flush=True)
for g in ctxs:
    g.close()
