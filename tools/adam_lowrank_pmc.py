#!/usr/bin/env python3
"""HBM-side traffic of chaorec_adam_lowrank_f32 (mode 0) and of the plain chaorec_adam_step_f32 at FREEDOM's image table:
one rocprofv3 --pmc pass per counter over tools/adam_lowrank_bench.py (program directly after `--`), per-kernel means.
FETCH_SIZE is reported raw (KB as the counter gives it) and with the streaming-read factor the microarch guide
prescribes for gfx950 is NOT applied here: the two kernels are compared with each other and with their element counts."""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_lines = []
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out = f"/tmp/adam_pmc_{counter}"
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "c", "--",
                        sys.executable, os.path.join(ROOT, "tools", "adam_lowrank_bench.py")], cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
    if not f:
        sys.exit(r.stdout[-1500:] + r.stderr[-1500:])
    acc = {}
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    for k, (n, s) in sorted(acc.items()):
        if "adam" in k:
            out_lines.append(f"{counter:11s} {k[:60]:60s} launches {n:4d}  mean {s / n / 1024:10.1f} MB")
I, K = 11384, 4096
out_lines.append(f"table: {I} x {K} fp32 = {I * K * 4 / 1e6:.1f} MB per array; mode 0 reads 3 and writes 3 arrays (+ gy, W: 3.9 MB); "
                 f"adam_step reads 4 and writes 3")
print("\n".join(out_lines))
