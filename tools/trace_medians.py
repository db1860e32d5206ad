#!/usr/bin/env python3
"""Median device time per (kernel, grid) from a rocprofv3 --kernel-trace CSV:  trace_medians.py <kernel_trace.csv> [substr]"""
import collections, csv, statistics, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    key = (n[:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:70s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):5d} median={statistics.median(v):8.1f} us")
