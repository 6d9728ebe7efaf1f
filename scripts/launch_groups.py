#!/usr/bin/env python3
"""Launches of a rocprofv3 kernel trace grouped by kernel / grid / duration class: one kernel name
covers several legs of bench.py, so a per-leg mean is read from its group.
usage: launch_groups.py <kernel_trace.csv> > groups.csv"""
import csv
import sys
from collections import defaultdict

groups = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "rag::" not in name:
            continue
        name = name.replace("void ", "").split("(")[0]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cls = ">=1ms" if dur >= 1_000_000 else ">=100us" if dur >= 100_000 else ">=10us" if dur >= 10_000 else "<10us"
        groups[(name, r["Grid_Size_X"], cls)].append(dur)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_size_x", "duration_class", "launches", "mean_ns", "min_ns", "max_ns"])
for (name, grid, cls), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, cls, len(d), sum(d) // len(d), min(d), max(d)])
