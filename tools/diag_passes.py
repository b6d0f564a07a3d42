"""Per-pass work and time of one frame of a bench workload (diagnostic)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opengl_raytracer_amd as rt
from bench import workload, setup_scene

name = sys.argv[1] if len(sys.argv) > 1 else "soup64k_4k_p6"
wl = workload(name)
# RT_DIAG_OPTS="voxel_res=32,sweeps_in_turn=1": options of the context (rt_set_option)
opts = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("RT_DIAG_OPTS", "").split(",") if kv))
opts.setdefault("kernel_timing", 1)
g = rt.RayTracer(wl["width"], wl["height"], options=opts)
setup_scene(g, wl)
if os.environ.get("RT_DIAG_WORLD"):  # one rank's share of the frame (interleaved tiles of 16 rows)
    g.set_row_tiles(16, 0, int(os.environ["RT_DIAG_WORLD"]))
if os.environ.get('RT_TRACE_MODE'):
    g.set_trace_mode(int(os.environ['RT_TRACE_MODE']))
g.render(wl["passes"]); g.sync(); g.counters(); g.stage_ms(); g.kernel_log()
g.raygen()
for p in range(wl["passes"]):
    for stage in ("intersect", "color"):
        getattr(g, stage)()
        ms = g.stage_ms(); c = g.counters(); log = g.kernel_log()
        kms = float(log["ms"].sum())
        if os.environ.get("RT_DIAG_LAUNCHES"):
            print("      launches:", [(int(r["form"]), int(r["rays_per_lane"]), round(float(r["ms"]), 3)) for r in log])
        if stage == "intersect":
            print(f"pass {p} intersect: {ms[1]:8.2f} ms (traverse {kms:8.2f})  rays {c['closest_rays']:9d}  pairs {c['closest_pairs']:.3e}"
                  f"  rate {c['closest_pairs']/max(kms,1e-9)/1e9:7.2f} Tpairs/s  slow steps {c['closest_slow_steps']:.3e}  cand {c['closest_candidates']:.3e}")
        else:
            print(f"pass {p} color    : {ms[2]:8.2f} ms (traverse {kms:8.2f})  rays {c['shadow_rays']:9d}  pairs {c['shadow_pairs']:.3e}"
                  f"  rate {c['shadow_pairs']/max(kms,1e-9)/1e9:7.2f} Tpairs/s  slow steps {c['shadow_slow_steps']:.3e}  cand {c['shadow_candidates']:.3e}")
