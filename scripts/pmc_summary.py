#!/usr/bin/env python3
"""Mean counter value per (kernel, counter) from rocprofv3 counter_collection CSVs.

usage: pmc_summary.py LABEL DIR [DIR ...]   -> CSV rows on stdout
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; the bytes column
doubles FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-byte-per-lane
streams at 64 bytes) and leaves WRITE_SIZE as is."""
import collections
import csv
import glob
import os
import sys

label = sys.argv[1]
acc = collections.OrderedDict()
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Kernel_Name"], row["Counter_Name"])
                # one row per (dispatch, counter): sum the per-dimension rows of one dispatch first
                disp = row["Dispatch_Id"]
                acc.setdefault(key, collections.OrderedDict())
                acc[key][disp] = acc[key].get(disp, 0.0) + float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["label", "kernel", "counter", "launches", "mean_value", "bytes_per_launch"])
for (kern, ctr), per in acc.items():
    vals = list(per.values())
    mean = sum(vals) / len(vals)
    b = ""
    if ctr == "FETCH_SIZE":
        b = f"{mean * 1024 * 2:.0f}"
    elif ctr == "WRITE_SIZE":
        b = f"{mean * 1024:.0f}"
    w.writerow([label, kern.split("(")[0], ctr, len(vals), f"{mean:.3f}", b])
