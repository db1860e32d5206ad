#!/usr/bin/env python3
"""Print the kernel sequence between the last two launches whose name contains <marker> in a rocprofv3 kernel trace:
python3 tools/step_sequence.py <trace.csv> <marker> [skip_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
ends = [i for k, i in enumerate(idx[:-1]) if idx[k + 1] - i > 3] + [idx[-1]]
a, b = ends[-skip - 2], ends[-skip - 1]
t0 = int(rows[a]["End_Timestamp"])
tot = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f}  {r['Kernel_Name'][:100]}")
print(f"{b - a} kernels, {tot:.1f} us of kernel time, span {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us")
