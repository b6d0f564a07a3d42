"""Prints the figures of a bench.py line that are looked at most."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s   ms_per_step %.3f   crc %s" % (d["value"], d["unit"], d["ms_per_step"], d["config"].get("frame_crc32")))
for key in ("stage_ms_per_step_rank0", "stage_ms_per_step_one_frame_at_a_time"):
    if key in d:
        print(key, {k: round(v, 3) for k, v in d[key].items()})
for key in ("roofline", "roofline_raygen", "roofline_color", "roofline_frame"):
    e = d.get(key)
    if e:
        print(key, e.get("kernel"), "achieved %s peak %s frac %s avg_ms %s" % (e.get("achieved"), e.get("peak"), e.get("frac"), e.get("avg_launch_ms")))
for e in d.get("roofline_kernels") or []:
    print("  kernel %-48s launches %2d avg_ms %.4f frac %s" % (e["kernel"], e["launches"], e["avg_launch_ms"], e.get("frac")))
for e in d.get("roofline_exact_mode") or []:
    print("  exact %s: %.1f %s frac %.3f avg_ms %.3f" % (e["kernel"], e["achieved"], e["unit"], e["frac"], e["avg_launch_ms"]))
for key in ("ms_per_step_moving_lights", "setup_ms", "device_bytes", "cpu_baseline"):
    if key in d:
        print(key, d[key])
