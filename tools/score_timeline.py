"""Kernel timeline of the carried-threshold gene_ranklist calls of tools/score_profile.py (ONLY_CARRIED=1) from a
rocprofv3 kernel trace: start offset and duration of every kernel of the last calls, and the gaps between them.

    python3 tools/score_timeline.py [train_steps] [ONLY_CARRIED|EPOCH_APART]         (on the GPU box)
"""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "300"
out = "/tmp/score_timeline"
shutil.rmtree(out, ignore_errors=True)
mode = sys.argv[2] if len(sys.argv) > 2 else "ONLY_CARRIED"          # or EPOCH_APART
env = dict(os.environ, TMPDIR="/tmp", **{mode: "1"})
r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
                    os.path.join(ROOT, "tools", "score_profile.py"), steps], cwd="/tmp", env=env, capture_output=True, text=True)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not f:
    sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
r_out = r.stdout
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "pack_items" in n and cur:
        calls.append(cur)
        cur = []
    if cur or "pack_items" in n:
        cur.append(r)
calls.append(cur)
print(len(calls), "calls");  print("\n".join(l for l in r_out.splitlines() if "us" in l))
for c in calls[-4:-1]:
    t0 = int(c[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in c:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void chaorec::", "").replace("chaorec::", "")
        print("   %8.1f +%7.1f us  gap %5.1f  grid %-9s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), n[:64]))
        prev_end = e
        busy += e - s
    print("   -> span %.1f us, kernel time %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
