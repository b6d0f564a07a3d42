"""Times one rank's share of the default bench frame for world sizes 1, 2, 4, 8 on one GPU:
the ceiling of strong-scaling efficiency before the gather (usage: python tools/diag_shard.py)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import opengl_raytracer_amd as rt
from opengl_raytracer_amd import sharding
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
base = None
for world in [int(x) for x in os.environ.get('WORLDS', '1,2,4,8').split(',')]:
    g = rt.RayTracer(wl["width"], wl["height"], options={"kernel_timing": int(os.environ.get("KERNEL_TIMING", "0"))})
    setup_scene(g, wl)
    tr = int(os.environ.get('TILE_ROWS', '0')) or sharding.choose_tile_rows(wl["height"], world)
    g.set_row_tiles(tr, int(os.environ.get('RANK', world - 1)) % world, world)
    g.render(wl["passes"]); g.sync()
    g.stage_ms(); g.kernel_log()
    t0 = time.perf_counter()
    for _ in range(3):
        g.render(wl["passes"])
    g.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    base = base or ms
    st = g.stage_ms()
    print("world %d: rank %d share %.2f ms/frame (ideal %.2f, efficiency ceiling %.0f %%)  stages %s" %
          (world, world - 1, ms, base / world, 100 * base / world / ms, [round(x / 3, 2) for x in st]), flush=True)
    agg = {}
    for r in g.kernel_log():
        k = (int(r["stage"]), int(r["form"]), int(r["rays_per_lane"]))
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["ms"])
    print("   ", {k: (v[0] // 3, round(v[1] / 3, 2)) for k, v in sorted(agg.items())}, flush=True)
    g.close()
