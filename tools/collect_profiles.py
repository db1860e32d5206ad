"""Collect the rocprofv3 evidence bench.py's roofline object refers to, on the GPU box.

    python3 tools/collect_profiles.py <tag>        e.g.  r01_d

Three separate rocprofv3 runs of the SAME bench command (the pool refuses --pmc together with trace domains
other than the kernel trace, and the microarch guide prescribes one counter per pass):
  1. --kernel-trace --stats                    -> profiles/<tag>_bench_kernel_stats.csv
  2. --pmc FETCH_SIZE  --kernel-trace          -> profiles/<tag>_pmc_FETCH_SIZE.csv  (per-kernel means)
  3. --pmc WRITE_SIZE  --kernel-trace          -> profiles/<tag>_pmc_WRITE_SIZE.csv
and profiles/spmm_traffic.json (HBM bytes per SpMM launch, corrected with the calibrated FETCH_SIZE factor for the
row-gather pattern -- see DESIGN.md "PMC calibration").  Output goes through gpurun_out/ (scratch) and the
summaries are copied to profiles/ by the caller's commit.

rocprofv3 is started with the program directly after `--` (python3 bench.py ...), never through a shell.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPMM = "spmm_csr_ordered_kernel<16, 1>"
FETCH_FACTOR_D64 = 1.478        # profiles/r01_c_spmm_hbm_scale.json, permutation-matrix calibration at D=64


def run(tag, name, extra):
    out = os.path.join(ROOT, "gpurun_out", f"{tag}_{name}")
    shutil.rmtree(out, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", name] + extra + \
          ["--", "python3", os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
           "--no-graph", "--no-trained-state"]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        raise SystemExit(f"{name}: rocprofv3 run failed rc={r.returncode}")
    return out, json.loads(line[-1])


def counter_means(out, counter):
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {out}")
    agg = {}
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"]
        s = agg.setdefault(k, [0, 0.0])
        s[0] += 1
        s[1] += float(row["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_x"
    prof = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    os.makedirs(prof, exist_ok=True)

    out, line = run(tag, "stats", ["--stats"])
    f = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(f, os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    stats = {r["Name"]: r for r in csv.DictReader(open(f))}
    spmm_avg_ns = [float(r["AverageNs"]) for n, r in stats.items() if SPMM in n]
    json.dump(line, open(os.path.join(prof, f"{tag}_bench_line_eager_under_rocprof.json"), "w"), indent=1)

    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out, _ = run(tag, counter, ["--pmc", counter])
        m = counter_means(out, counter)
        means[counter] = m
        with open(os.path.join(prof, f"{tag}_pmc_{counter}.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Launches", f"{counter}_KB_mean"])
            for k, (n, v) in sorted(m.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
                w.writerow([k, n, f"{v:.3f}"])

    def pick(m):
        for k, v in m.items():
            if SPMM in k:
                return v[1]
        raise SystemExit("SpMM kernel not in counter output: " + ", ".join(m))

    fetch_kb, write_kb = pick(means["FETCH_SIZE"]), pick(means["WRITE_SIZE"])
    traffic = {
        "kernel": "spmm_csr_ordered_kernel<16,1>",
        "workload": "bench.py sports-shaped LightGCN step, D=64 (eager launches, 35 steps): mean over the 3 forward + 3 "
                    "backward launches of a step",
        "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
        "correction": f"FETCH_SIZE scaled by the calibrated factor for this gather pattern at D=64 ({FETCH_FACTOR_D64}: "
                      "permutation-matrix run, profiles/r01_c_spmm_hbm_scale.json); WRITE_SIZE exact; separate --pmc "
                      "passes",
        "hbm_bytes_per_launch": (fetch_kb * FETCH_FACTOR_D64 + write_kb) * 1024.0,
        "kernel_avg_us_rocprofv3": spmm_avg_ns[0] / 1e3 if spmm_avg_ns else None,
        "note": "counts L2 fabric-side requests; Infinity-Cache hits are included, so at this cache-resident size this "
                "is L2-miss traffic, not DRAM traffic",
        "source": [f"profiles/{tag}_pmc_FETCH_SIZE.csv", f"profiles/{tag}_pmc_WRITE_SIZE.csv",
                   f"profiles/{tag}_bench_kernel_stats.csv"],
    }
    json.dump(traffic, open(os.path.join(prof, "spmm_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    for n, r in list(stats.items())[:12]:
        print(f"{n[:70]:70s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.1f}us  {r['Percentage']}%")


if __name__ == "__main__":
    main()
