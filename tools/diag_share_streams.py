"""One rank's share of a tile split, rendered the way bench.py renders it (TileShardedFrame: contexts bound to
torch streams and device framebuffers, no gather), for several numbers of frames in flight
(usage: WORLD=8 INFLIGHT=3,4 python tools/diag_share_streams.py [workload])."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
import opengl_raytracer_amd as rt
from opengl_raytracer_amd import sharding
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
world = int(os.environ.get("WORLD", "8"))
ks = [int(x) for x in os.environ.get("INFLIGHT", "3,4").split(",")]
frames = int(os.environ.get("FRAMES", "24"))
dev = torch.device("cuda:0")
ctxs = []
for _ in range(max(ks)):
    # OPTS="walk_budget=24,...": further options of the contexts (rt_set_option)
    opts = {"sweeps_in_turn": 1}
    opts.update((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv))
    g = rt.RayTracer(wl["width"], wl["height"], options=opts)
    setup_scene(g, wl)
    ctxs.append(g)
for w in (1, world):
    sharding._world = lambda w=w: (0, w)
    for k in ks:
        job = sharding.TileShardedFrame(ctxs[:k], wl["width"], wl["height"], dev,
                                        tile_rows=int(os.environ["TILE_ROWS"]) if w > 1 and os.environ.get("TILE_ROWS") else None)
        job.gather = False
        for _ in range(k + 1):
            job.render(wl["passes"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            job.render(wl["passes"])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"share 1/{w}, {k} frames in flight: {(t2 - t0) / frames * 1e3:.3f} ms per frame "
              f"(host enqueue {(t1 - t0) / frames * 1e3:.3f})", flush=True)
