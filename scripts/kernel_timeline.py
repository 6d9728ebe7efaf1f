#!/usr/bin/env python3
"""Compact timeline of this library's kernels from a rocprofv3 --kernel-trace CSV:
usage kernel_timeline.py TRACE.csv [first_prep_index] [count]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rag::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "prep_queries" in r["Kernel_Name"]]
first = starts[int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 12
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("rag::")[1].split("(")[0][:44]
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):>6} x {r['Workgroup_Size_X']:>4}  {name}")
