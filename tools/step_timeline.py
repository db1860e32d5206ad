"""Timeline of the captured training step from a rocprofv3 kernel trace: per kernel of the repeating step sequence the
average duration and the average gap to its predecessor (device timestamps).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -o trace -- python3 <repo>/bench.py --steps 60 \
        --warmup 5 --no-cpu-baseline --no-trained-state --no-hbm-regime
    python3 tools/step_timeline.py <out> [first_kernel_substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "spmm_csr_ordered_kernel"
    f = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the step = the most frequent gap-free window starting at an acc_init SpMM is hard to tell from names alone:
    # take windows between consecutive occurrences of the (L SpMM, bpr_fwd_bwd) pattern instead
    idx = [i for i, n in enumerate(names) if "bpr_fwd_bwd" in n or "bpr_fwd_terms_drawn" in n]
    if len(idx) < 12:
        raise SystemExit("no step pattern found")
    period = idx[6] - idx[5]
    starts = [i for a, i in zip(idx, idx[1:]) if i - a == period]
    # align the window so that it begins `k` kernels before the BPR kernel, k = number of forward SpMMs
    k = 0
    while k < period and first in names[starts[0] - k - 1]:
        k += 1
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    tot = []
    for s in starts[2:-2]:
        w = rows[s - k:s - k + period]
        if [r["Kernel_Name"] for r in w] != [r["Kernel_Name"] for r in rows[starts[2] - k:starts[2] - k + period]]:
            continue
        prev_end = int(rows[s - k - 1]["End_Timestamp"])
        t_begin = int(w[0]["Start_Timestamp"])
        for j, r in enumerate(w):
            a = agg[j]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a[2] += int(r["Start_Timestamp"]) - prev_end
            prev_end = max(prev_end, int(r["End_Timestamp"]))
        tot.append(int(rows[s - k + period]["Start_Timestamp"]) - t_begin)
    ref = rows[starts[2] - k:starts[2] - k + period]
    print(f"{len(tot)} steps, period {period} kernels, mean step {sum(tot) / len(tot) / 1e3:.2f} us")
    for j, r in enumerate(ref):
        n, d, g = agg[j]
        print(f"{j:2d} dur {d / n / 1e3:7.2f} us  gap {g / n / 1e3:6.2f} us  {r['Kernel_Name'][:110]}")


if __name__ == "__main__":
    main()
