"""Experiment builds of the scoring kernels (CHAOREC_EXTRA_HIPCC_FLAGS -> their own library under csrc/exp/, never the product
one) timed on one shape:  python3 tools/sweep_variants.py [build|run] U I D -- "-DCHAOREC_PF_UB128=3" "-DCHAOREC_SWEEP_STAGE=1" ...
`build` cross-compiles the variants (no GPU needed: do it before gpurun), `run` times cold / hinted calls per variant."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1]
sep = sys.argv.index("--")
shape = sys.argv[2:sep]
variants = [""] + sys.argv[sep + 1:]
for v in variants:
    env = dict(os.environ)
    if v:
        env["CHAOREC_EXTRA_HIPCC_FLAGS"] = v
    else:
        env.pop("CHAOREC_EXTRA_HIPCC_FLAGS", None)
    if mode == "build":
        subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; print(_lib.build())"], cwd=ROOT, env=env)
    else:
        env["TIMES"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_case.py")] + shape, cwd=ROOT, env=env,
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("times")]
        lib = [l for l in r.stdout.splitlines() if l.startswith("lib ")]
        print(f"{v or '(product)':50s} {lib[-1] if lib else ''} {line[-1] if line else r.stderr[-300:]}", flush=True)
