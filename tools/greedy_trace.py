#!/usr/bin/env python3
"""Per-launch durations of the greedy-NMS kernels from a rocprofv3 --kernel-trace csv (last call of tools/bench_greedy.py):
python tools/greedy_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "greedy" in r["Kernel_Name"] or "topk_select" in r["Kernel_Name"] or "subpixel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last call = everything after the last topk_select but one
ends = [i for i, r in enumerate(rows) if "subpixel" in r[2]]
last = rows[ends[-2] + 1: ends[-1] + 1] if len(ends) > 1 else rows
t0 = last[0][0]
for s, e, n in last:
    short = n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {short}")
print(f"total span {(last[-1][1] - t0) / 1e3:.1f} us, kernel time {sum(e - s for s, e, _ in last) / 1e3:.1f} us")
