"""Times every rank's share of a bench frame on one GPU for several row-tile heights: the slowest
rank bounds the strong-scaling efficiency (usage: WORLD=8 TILE_ROWS=16,32,64 python tools/diag_tiles.py)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

wl = workload(sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6")
world = int(os.environ.get("WORLD", "8"))
# OPTS="walk_budget=24,voxel_res=32": options of the context (rt_set_option)
opts = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv))
g = rt.RayTracer(wl["width"], wl["height"], options=opts)
setup_scene(g, wl)


def share(tile_rows, rank, world):
    g.set_row_tiles(tile_rows, rank, world)
    g.render(wl["passes"]); g.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        g.render(wl["passes"])
    g.sync()
    return (time.perf_counter() - t0) / 2 * 1e3


if os.environ.get("ONLY_RANK"):
    # one rank's share, several frames (for a profiler run around this program)
    tr = int(os.environ.get("TILE_ROWS", "15").split(",")[0])
    for _ in range(3):
        print("share of rank %s: %.3f ms" % (os.environ["ONLY_RANK"], share(tr, int(os.environ["ONLY_RANK"]), world)), flush=True)
    print([round(x, 3) for x in g.stage_ms()])
    g.close()
    sys.exit(0)
base = share(wl["height"], 0, 1)
print("whole frame %.2f ms  %s" % (base, opts), flush=True)
if os.environ.get("RANKS"):  # only these ranks' shares
    tr = int(os.environ.get("TILE_ROWS", "15").split(",")[0])
    print("world %d tile_rows %d:" % (world, tr), ["%.3f" % share(tr, int(r), world) for r in os.environ["RANKS"].split(",")], flush=True)
    g.close()
    sys.exit(0)
for tr in [int(x) for x in os.environ.get("TILE_ROWS", "16,32,64,128").split(",")]:
    ms = [share(tr, r, world) for r in range(world)]
    print("world %d tile_rows %d: max %.2f mean %.2f ms -> speedup %.2fx  %s" %
          (world, tr, max(ms), sum(ms) / len(ms), base / max(ms), [round(m, 1) for m in ms]), flush=True)
g.close()
