"""Per (kernel, grid) table of a rocprofv3 kernel trace: calls, average and total duration -- the grid tells the shapes apart.
    python3 tools/kernel_table.py <trace dir> [steps] [min share %]
`steps` (optional) divides the totals into per-step figures."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
min_share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
f = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = defaultdict(lambda: [0, 0])
for r in rows:
    grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
    wg = "x".join(str(r.get(k, "")) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")) if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "")
    a = agg[(r["Kernel_Name"].split("(")[0][-70:], grid, wg)]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"{len(rows)} launches, {tot / 1e6:.2f} ms of kernel time" + (f", {tot / steps / 1e3:.1f} us per step" if steps else ""))
for (name, grid, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if 100.0 * t / tot < min_share:
        continue
    per = f"  {n / steps:5.2f}/step {t / steps / 1e3:8.1f} us/step" if steps else ""
    print(f"{100.0 * t / tot:5.1f}%  n={n:6d}  avg {t / n / 1e3:8.1f} us{per}  grid {grid:>14s} wg {wg:>9s}  {name}")
