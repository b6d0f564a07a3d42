"""Per kernel: the counters tools/prof_memory_path.sh collected, per launch
(python tools/memory_path_summary.py <dir> [out.json [workload]]).  With out.json: the derived figures bench.py
quotes beside a kernel's roofline -- share of the launch the address units are busy, L1 hit rate, cache-line
accesses per wave-level load, L2 round trip in cycles."""
import collections
import csv
import glob
import json
import sys

base = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(f"{base}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0)):
    if not k.startswith("rtk::traverse") and "compact" not in k and "color" not in k:
        continue
    print(k)
    for name, v in sorted(agg[k].items()):
        n = len(launches[(k, name)])
        print(f"   {name:44s} {v / n:14.4g} per launch ({n} launches)")

if len(sys.argv) > 2:
    CUS, XCDS = 256, 8
    derived = {}
    for k, c in agg.items():
        per = {name: v / len(launches[(k, name)]) for name, v in c.items()}
        need = ("GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum",
                "TCP_TCC_READ_REQ_LATENCY_sum", "TA_FLAT_READ_WAVEFRONTS_sum")
        if not k.startswith("rtk::traverse") or any(per.get(n, 0) <= 0 for n in need):
            continue
        cycles = per["GRBM_GUI_ACTIVE"] / XCDS
        derived[k] = {
            "address_units_busy": per["TA_TA_BUSY_sum"] / CUS / cycles,
            "l1_hit_rate": 1.0 - per["TCP_TCC_READ_REQ_sum"] / per["TCP_TOTAL_CACHE_ACCESSES_sum"],
            "cache_accesses_per_wave_load": per["TCP_TOTAL_CACHE_ACCESSES_sum"] / per["TA_FLAT_READ_WAVEFRONTS_sum"],
            "l2_round_trip_cycles": per["TCP_TCC_READ_REQ_LATENCY_sum"] / per["TCP_TCC_READ_REQ_sum"],
            "wave_loads_per_launch": per["TA_FLAT_READ_WAVEFRONTS_sum"],
        }
    json.dump({"workload": sys.argv[3] if len(sys.argv) > 3 else "soup64k_4k_p6",
               "source": "tools/prof_memory_path.sh (rocprofv3 --pmc TA_* TCP_* GRBM_GUI_ACTIVE, one run per set)",
               "kernels": derived}, open(sys.argv[2], "w"), indent=1)
