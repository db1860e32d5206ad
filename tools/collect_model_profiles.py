#!/usr/bin/env python3
"""Kernel times of the MMGCN / FREEDOM captured steps for the bench's `models.*.roofline`: rocprofv3 kernel
stats of `bench.py --model X` on ONE stream (CHAOREC_MMGCN_STREAMS=0: with the two modality branches side by side a kernel's
begin-to-end time contains the other branch's kernels), the CSV kept as gpurun_out/r06_<X>_kernel_stats.csv and the per-step
kernels as gpurun_out/model_kernel_times.json (copied to profiles/ by hand: the bench quotes it only when the sources it was
measured on are the ones it runs).  Run on the GPU box:  python3 tools/collect_model_profiles.py"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
STEPS, WARMUP = 100, 10


def sha(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()


def main():
    result = {"sources": {f: sha(f) for f in ("chaorec_amd/csrc/gemm_bf16x3.hip", "chaorec_amd/csrc/feature_adam.hip",
                                               "chaorec_amd/csrc/spmm.hip", "chaorec_amd/csrc/bpr.hip")}, "models": {}}
    for name in ("MMGCN", "FREEDOM"):
        d = os.path.join(OUT, f"modelprof_{name}")
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp", CHAOREC_MMGCN_STREAMS="0")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "m", "--",
                            "python3", os.path.join(ROOT, "bench.py"), "--model", name, "--steps", str(STEPS), "--warmup",
                            str(WARMUP), "--no-cpu-baseline"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else None
        f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not f:
            print(name, "no kernel trace", r.stderr[-800:])
            continue
        st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        keep = os.path.join(OUT, f"r06_{name}_one_stream_kernel_stats.csv")
        if st:
            shutil.copy(st[0], keep)
        # per (kernel, grid): the stats file pools every shape a kernel runs at, the grid tells them apart.  Kernels of the
        # STEP: at least one launch per timed step; one-off kernels (graph build, kNN, ranking) drop out
        agg = {}
        for row in csv.DictReader(open(f[0])):
            grid = "x".join(str(row.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in row \
                else str(row.get("Grid_Size", ""))
            a = agg.setdefault((row["Kernel_Name"].split("(")[0], grid), [0, 0])
            a[0] += 1
            a[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        kernels = [{"name": k[0], "grid": k[1], "calls": n, "avg_us": t / n / 1e3, "total_us": t / 1e3}
                   for k, (n, t) in agg.items() if n >= STEPS]
        kernels.sort(key=lambda k: -k["total_us"])
        result["models"][name] = {"ms_per_step_under_rocprof_one_stream": ms, "csv": os.path.relpath(keep, ROOT).replace("gpurun_out", "profiles"),
                                  "kernels": kernels[:12]}
        shutil.rmtree(d, ignore_errors=True)
        print(name, ms, [(k["name"][-60:], round(k["avg_us"], 1), k["calls"]) for k in kernels[:4]])
    json.dump(result, open(os.path.join(OUT, "model_kernel_times.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
