"""Per kernel: the counters tools/prof_memory_path.sh collected, per launch (python tools/memory_path_summary.py <dir>)."""
import collections
import csv
import glob
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
