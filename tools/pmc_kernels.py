"""Per-kernel SQ counter means for a command:  python3 tools/pmc_kernels.py <kernel-substring> -- python3 tools/x.py
Four rocprofv3 --pmc passes (4 counters each, kernel trace only), grouped by kernel name and grid size."""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"],
    ["SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU"],
    ["SQ_ACTIVE_INST_VMEM", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"],
    ["SQ_INST_CYCLES_VMEM_RD", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_INST_LEVEL_VMEM"],
    ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TOTAL_WAVEFRONTS_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"],
]
sep = sys.argv.index("--")
filt, cmd = sys.argv[1:sep], sys.argv[sep + 1:]
agg = {}
only = [int(x) for x in os.environ.get("PMC_PASSES", "0,1,2,3").split(",")]
for i, ctrs in enumerate(PASSES):
    if i not in only:
        continue
    out = os.path.join(ROOT, "gpurun_out", f"pmc_pass{i}")
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", out, "-o", "p", "--"] + cmd,
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=int(os.environ.get("PMC_TIMEOUT", "150")))
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        sys.exit(r.stdout[-1500:] + r.stderr[-1500:])
    rows = [row for row in csv.DictReader(open(f[0])) if not filt or any(s in row["Kernel_Name"] for s in filt)]
    last_n = int(os.environ.get("LAST_N", "0"))          # only the last N launches of every kernel (e.g. the timed calls)
    if last_n:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        seen, keep = {}, []
        for row in reversed(rows):
            kk = (row["Kernel_Name"], row["Grid_Size"], row["Counter_Name"])
            seen[kk] = seen.get(kk, 0) + 1
            if seen[kk] <= last_n:
                keep.append(row)
        rows = keep
    for row in rows:
        k = row["Kernel_Name"]
        key = (k.split("(")[0][-60:], row["Grid_Size"])
        d = agg.setdefault(key, {})
        c = d.setdefault(row["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(row["Counter_Value"])
for key, d in agg.items():
    print(key)
    for name, (n, tot) in sorted(d.items()):
        print(f"   {name:28s} {tot / n:12.4g}   (n={n})")
