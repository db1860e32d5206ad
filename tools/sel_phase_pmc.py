"""VALU / SALU / VMEM instruction counts of the selection kernel by phase: builds with the stage-cut experiment macros
(CHAOREC_SEL_EXP = 5 expansion only, 6 + exact scores and keys, 9 = whole kernel; the cuts act on calls with
hint_rank >= 1000, so the earlier calls of the script leave valid thresholds) under rocprofv3 --pmc."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1] if len(sys.argv) > 1 else ""
for v in ["-DCHAOREC_SEL_EXP=5", "-DCHAOREC_SEL_EXP=6", "-DCHAOREC_SEL_EXP=9"]:
    env = dict(os.environ, CHAOREC_EXTRA_HIPCC_FLAGS=(v + " " + extra).strip(), TMPDIR="/tmp", EPOCH_APART="1", LAST_N="4",
               PMC_PASSES="0,2", TIMED_HINT_RANK="1100" if v else "100")
    subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT, env=env,
                          stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_kernels.py"), "score_select_kernel_pf<64, 512", "--",
                        sys.executable, os.path.join(ROOT, "tools", "score_profile.py"), "300"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    keep = [l for l in r.stdout.splitlines() if any(k in l for k in ("INSTS_VALU", "INSTS_SALU", "INSTS_VMEM_RD", "WAVE_CYCLES", "ACTIVE_INST_VALU"))]
    print("==", v, "\n" + "\n".join(keep) if keep else r.stdout[-800:] + r.stderr[-800:], flush=True)
subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT, stdout=subprocess.DEVNULL)
