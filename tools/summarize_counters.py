"""Summarises the rocprofv3 outputs of tools/prof_counters.sh into one text file.
usage: python tools/summarize_counters.py gpurun_out/prof_<tag> profiles/<name>.txt [workload]
Also writes profiles/hbm_traffic_per_launch.json (kernel -> HBM bytes per launch from FETCH_SIZE and
WRITE_SIZE with the gfx950 corrections), which bench.py quotes as roofline.traffic; for a workload other
than the default one the file is hbm_traffic_per_launch_<workload>.json."""
import collections
import csv
import json
import os
import sys

base, out = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "soup64k_4k_p6"


def short(n):
    return n.split("(")[0].replace("void ", "")


lines = []
stats = list(csv.DictReader(open(f"{base}/trace/t_kernel_stats.csv")))
lines.append(f"== rocprofv3 --kernel-trace --stats (python3 bench.py --workload {workload} --steps 3 --warmup 1 --no-cpu-baseline --sweeps-in-turn --frames-in-flight 1) ==")
traffic = {}
lines.append(f"{'kernel':70s} {'calls':>6s} {'avg_us':>12s} {'total_ms':>10s} {'%':>6s}")
for r in stats[:24]:
    lines.append(f"{short(r['Name'])[:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:12.1f} "
                 f"{float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")
dur = {short(r["Name"]): (float(r["AverageNs"]) * 1e-9, int(r["Calls"])) for r in stats}

# raygen also runs on the 1/32 tile share that the all-pairs measurement renders: only its whole-frame launches
# (the largest grid) are summarised
FULL_FRAME_ONLY = ("rtk::raygen_kernel",)
trace = list(csv.DictReader(open(f"{base}/trace/t_kernel_trace.csv")))
for name in FULL_FRAME_ONLY:
    rows = [r for r in trace if short(r["Kernel_Name"]) == name]
    if rows:
        gmax = max(int(r["Grid_Size_X"]) for r in rows)
        full = [r for r in rows if int(r["Grid_Size_X"]) == gmax]
        dur[name] = (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in full) / len(full) * 1e-9, len(full))

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in ("pmc_sq/sq", "pmc_fetch/f", "pmc_write/w", "pmc_misc/m"):
    rows = list(csv.DictReader(open(f"{base}/{f}_counter_collection.csv")))
    gmax = {name: max([int(r["Grid_Size"]) for r in rows if short(r["Kernel_Name"]) == name] or [0]) for name in FULL_FRAME_ONLY}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k in gmax and int(r["Grid_Size"]) != gmax[k]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, f)].add(r["Dispatch_Id"])
# one whole frame of the counter run: the dispatches from one whole-frame primary traversal up to the next
# (rt_render leaves raygen to that kernel, so it is the first kernel of a frame)
frame = None
sq_rows = list(csv.DictReader(open(f"{base}/pmc_sq/sq_counter_collection.csv")))
FIRST = "rtk::traverse_grid_kernel<false>"
gen = sorted({int(r["Dispatch_Id"]) for r in sq_rows if short(r["Kernel_Name"]) == FIRST
              and int(r["Grid_Size"]) == max(int(q["Grid_Size"]) for q in sq_rows if short(q["Kernel_Name"]) == FIRST)})
if len(gen) >= 4:
    lo, hi = gen[2], gen[3]
    sel = [r for r in sq_rows if lo <= int(r["Dispatch_Id"]) < hi]
    frame = {"valu_wave_insts": sum(float(r["Counter_Value"]) for r in sel if r["Counter_Name"] == "SQ_INSTS_VALU"),
             "salu_wave_insts": sum(float(r["Counter_Value"]) for r in sel if r["Counter_Name"] == "SQ_INSTS_SALU"),
             "dispatches": len({r["Dispatch_Id"] for r in sel})}
    lines.append("")
    lines.append(f"== one whole frame (dispatches {lo}..{hi - 1} of the SQ counter run): {frame['dispatches']} kernels, "
                 f"SQ_INSTS_VALU {frame['valu_wave_insts']:.4g}, SQ_INSTS_SALU {frame['salu_wave_insts']:.4g} ==")
lines.append("")
lines.append("== hardware counters, per launch (each --pmc set collected in a run of its own) ==")
for k, v in agg.items():
    if not any(s in k for s in ("traverse", "raygen", "finish", "setup", "cluster", "build", "compact", "grid_list", "bin_")):
        continue
    n = max(len(calls[(k, f)]) for f in ("pmc_sq/sq", "pmc_fetch/f", "pmc_write/w", "pmc_misc/m"))
    t, _ = dur.get(k, (float("nan"), 0))
    lines.append(f"{k}   launches={n}  avg duration {t*1e3:.3f} ms")
    per = {a: b / n for a, b in v.items()}
    valu = per.get("SQ_INSTS_VALU", 0.0)
    lines.append(f"   SQ_INSTS_VALU {valu:.4g}  SQ_INSTS_SALU {per.get('SQ_INSTS_SALU',0):.4g}  SQ_INSTS_LDS {per.get('SQ_INSTS_LDS',0):.4g}"
                 f"  SQ_WAVES {per.get('SQ_WAVES',0):.4g}  SQ_INSTS_VALU_MFMA_F32 {per.get('SQ_INSTS_VALU_MFMA_F32',0):.4g}")
    clk = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t if t == t and t > 0 else float("nan")
    lines.append(f"   VALU wave-instructions/s {valu/t/1e9:.1f} G   effective clock (GRBM_GUI_ACTIVE/8/t) {clk/1e9:.2f} GHz"
                 f"   VALU issue utilisation {valu/t/(1024*0.25*clk)*100 if clk==clk and clk>0 else float('nan'):.1f} % of 1024 SIMDs x 0.25 inst/clk")
    # FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-B reads as 64 B on gfx950 (x2)
    lines.append(f"   FETCH_SIZE {per.get('FETCH_SIZE',0)*1024/1e6:.1f} MB raw (x2 for wide coalesced reads on gfx950 = {per.get('FETCH_SIZE',0)*2048/1e6:.1f} MB)"
                 f"   WRITE_SIZE {per.get('WRITE_SIZE',0)*1024/1e6:.1f} MB   -> {(per.get('FETCH_SIZE',0)*2048+per.get('WRITE_SIZE',0)*1024)/t/1e12:.3f} TB/s HBM")
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        traffic[k] = {"hbm_bytes_per_launch": per["FETCH_SIZE"] * 2048 + per["WRITE_SIZE"] * 1024,
                      "launches": n, "avg_ms": t * 1e3, "valu_wave_insts_per_launch": valu,
                      "valu_issue_frac": (valu / t / (1024 * 0.25 * clk)) if clk == clk and clk > 0 else None,
                      "valu_busy_quadcycles_per_launch": per.get("SQ_ACTIVE_INST_VALU")}
    lines.append(f"   SQ_ACTIVE_INST_VALU {per.get('SQ_ACTIVE_INST_VALU',0):.4g} quad-cycles = {per.get('SQ_ACTIVE_INST_VALU',0)/valu if valu else 0:.2f} per VALU instruction"
                 f"   SQ_WAVE_CYCLES {per.get('SQ_WAVE_CYCLES',0):.4g}  SQ_WAIT_INST_ANY {per.get('SQ_WAIT_INST_ANY',0):.4g}")
    lines.append(f"   SQ_LDS_IDX_ACTIVE {per.get('SQ_LDS_IDX_ACTIVE',0):.4g}  SQ_LDS_BANK_CONFLICT {per.get('SQ_LDS_BANK_CONFLICT',0):.4g}")
open(out, "w").write("\n".join(lines) + "\n")
json.dump({"source": os.path.basename(out), "workload": workload,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each; FETCH_SIZE x 2 KiB "
                   "(128-byte reads count as 64 on gfx950), WRITE_SIZE x 1 KiB; per launch",
           "frame": frame, "kernels": traffic},
          open(os.path.join(os.path.dirname(out), "hbm_traffic_per_launch.json" if workload == "soup64k_4k_p6"
                            else f"hbm_traffic_per_launch_{workload}.json"), "w"), indent=1)
print("\n".join(lines))
