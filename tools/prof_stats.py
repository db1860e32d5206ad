#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv:  python3 tools/prof_stats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:86]:86s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e6:9.2f} ms {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s}%')
