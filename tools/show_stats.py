"""Prints the head of a rocprofv3 kernel stats table: python tools/show_stats.py <t_kernel_stats.csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(r["Name"].split("(")[0].replace("void ", "")[:60].ljust(60), r["Calls"].rjust(6),
          "%10.1f us" % (float(r["AverageNs"]) / 1e3), "%9.2f ms" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
