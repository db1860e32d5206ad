"""Rebuild libchaorec_hip.so with experiment macros and report the duration of the selection kernel in the cold and
the carried-threshold call (compare variants only within ONE run of this script: the kernels' times depend on the
training state, TRAIN_STEPS, default 3000 in the EPOCH_APART mode) (rocprofv3 kernel trace of tools/score_profile.py) per variant."""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or ["", "-DCHAOREC_SEL_EXP=5", "-DCHAOREC_SEL_EXP=6", "-DCHAOREC_SEL_EXP=1"]
for v in variants:
    env = dict(os.environ, CHAOREC_EXTRA_HIPCC_FLAGS=v, TMPDIR="/tmp")
    subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT, env=env)
    out = "/tmp/selvar"
    shutil.rmtree(out, ignore_errors=True)
    if os.environ.get("EPOCH_APART"):       # bench.py's steady state only: the timeline of the last calls
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_timeline.py"), os.environ.get("TRAIN_STEPS", "3000"), "EPOCH_APART"], cwd=ROOT, env=env,
                           capture_output=True, text=True)
        print(f"== {v or 'baseline'}\n" + "\n".join(r.stdout.splitlines()[-6:]), flush=True)
        continue
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
                    os.path.join(ROOT, "tools", "score_profile.py"), "1500"], cwd="/tmp", env=env, capture_output=True)
    rows = list(csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], []
    for r in rows:
        n = r["Kernel_Name"]
        if "pack_items" in n and cur:
            calls.append(cur)
            cur = []
        if "score_" in n or "pack_items" in n:
            cur.append(r)
    calls.append(cur)
    def dur(c, key, which=0):
        k = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c if key in r["Kernel_Name"]]
        return k[which] / 1e3 if len(k) > which else float("nan")
    print(f"== {v or 'baseline':24s} cold: sample {dur(calls[3], 'sample'):6.1f} sweep {dur(calls[3], 'sweep'):6.1f} "
          f"select {dur(calls[3], 'select_kernel_pf<64, 512'):6.1f} | carried: sweep {dur(calls[10], 'sweep'):6.1f} "
          f"select {dur(calls[10], 'select_kernel_pf<64, 512'):6.1f}", flush=True)
subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT)
