"""Per microbenchmark kernel: SQ_ACTIVE_INST_VALU per SQ_INSTS_VALU (the largest launch of each kind)."""
import collections
import csv
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from asm_counts import NAMES

rows = list(csv.DictReader(open(sys.argv[1])))
best = {}
for r in rows:
    m = re.search(r"k<(\d+)>", r["Kernel_Name"])
    if not m:
        continue
    kind = int(m.group(1))
    d = best.setdefault((kind, r["Dispatch_Id"]), dict(grid=int(r["Grid_Size"])))
    d[r["Counter_Name"]] = float(r["Counter_Value"])
per_kind = collections.defaultdict(list)
for (kind, _), d in best.items():
    per_kind[kind].append(d)
print(f"{'row':<24s} {'SQ_INSTS_VALU':>14s} {'SQ_ACTIVE_INST_VALU':>20s} {'active/inst':>12s} {'GRBM_GUI_ACTIVE':>16s} {'inst/clk/CU':>12s}")
for kind in sorted(per_kind):
    d = max(per_kind[kind], key=lambda e: e.get("SQ_INSTS_VALU", 0))
    iv, av, gui = d.get("SQ_INSTS_VALU", 0), d.get("SQ_ACTIVE_INST_VALU", 0), d.get("GRBM_GUI_ACTIVE", 0)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    clk = gui / 8.0 if gui else float("nan")
    print(f"{NAMES[kind]:<24s} {iv:14.4g} {av:20.4g} {av / iv if iv else float('nan'):12.3f} {gui:16.4g} {iv / clk / 256 if gui else float('nan'):12.3f}")
