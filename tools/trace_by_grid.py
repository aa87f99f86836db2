#!/usr/bin/env python
"""Per-kernel, per-launch-shape durations out of a rocprofv3 --kernel-trace csv: groups the dispatches of every kernel whose name
contains one of the given substrings by (name, grid, workgroup) and prints count / median / min in microseconds.
Usage: trace_by_grid.py <dir with *kernel_trace.csv> substr [substr ...]"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict


def main(d, subs):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return
    groups = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r.get("Kernel_Name", "")
        if not any(s in name for s in subs):
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (name.replace("fhip::", "").replace("void ", "")[:90], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
        groups[key].append(dur)
    for (name, grid, wg), v in sorted(groups.items()):
        print(f"{name:92s} grid {grid:>9s} wg {wg:>4s}  n {len(v):4d}  median {statistics.median(v):8.2f} us  min {min(v):8.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
