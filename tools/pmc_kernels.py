"""Per-kernel SQ counter means for a command:  python3 tools/pmc_kernels.py <kernel-substring> -- python3 tools/x.py
Four rocprofv3 --pmc passes (4 counters each, kernel trace only), grouped by kernel name and grid size."""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"],
    ["SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU"],
    ["SQ_ACTIVE_INST_VMEM", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"],
]
sep = sys.argv.index("--")
filt, cmd = sys.argv[1:sep], sys.argv[sep + 1:]
agg = {}
for i, ctrs in enumerate(PASSES):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_pass{i}")
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", out, "-o", "p", "--"] + cmd,
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=150)
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        sys.exit(r.stdout[-1500:] + r.stderr[-1500:])
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if filt and not any(s in k for s in filt):
            continue
        key = (k.split("(")[0][-60:], row["Grid_Size"])
        d = agg.setdefault(key, {})
        c = d.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
for key, d in agg.items():
    print(key)
    for name, (n, tot) in sorted(d.items()):
        print(f"   {name:28s} {tot / n:12.4g}   (n={n})")
