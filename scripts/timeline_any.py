#!/usr/bin/env python3
"""Timeline of the last `count` kernels of this library in a rocprofv3 --kernel-trace CSV, with the
gap to the previous kernel: usage timeline_any.py TRACE.csv [count]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rag::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("rag::")[1].split("(")[0][:48]
    gap = 0 if prev_end is None else s - prev_end
    print(f"{(s - t0) / 1e3:10.1f} us  gap {gap / 1e3:6.1f}  +{(e - s) / 1e3:8.1f} us  grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):>6} x {r['Workgroup_Size_X']:>4}  {name}")
    prev_end = e
