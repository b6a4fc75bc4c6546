"""Group a rocprofv3 kernel_trace.csv by (kernel, grid size): per-workload launch counts and average durations.
usage: python tools/summarize_trace.py <kernel_trace.csv> [out.csv]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    rows[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = [("kernel", "grid_threads", "calls", "avg_us", "min_us", "max_us", "total_ms")]
for (name, grid), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    out.append((name, grid, len(d), round(sum(d) / len(d) / 1e3, 3), round(min(d) / 1e3, 3), round(max(d) / 1e3, 3), round(sum(d) / 1e6, 3)))
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
