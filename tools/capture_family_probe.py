"""Which of the eagerly-run family models also train as a captured hipGraph step?  For each model: two epochs through
chaorec_amd.main at baby size (synthetic), eager (the shipped setting) and with CHAOREC_TRY_CAPTURE naming it; prints
seconds per run and the best Recall@20 of each.     python3 tools/capture_family_probe.py [Model ...]"""
import logging
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import main as cmain, dataload  # noqa: E402

models = sys.argv[1:] or ["SimGCL", "XSimGCL", "SelfCF", "SLMRec", "SGL", "MMGCL"]
dataload.SYNTHETIC_FEATURE_DIMS["default"] = (96, 64)
real = cmain.load_yaml_config
cmain.load_yaml_config = lambda name: {k: (v if k == "hyper_parameters" else v[:1]) for k, v in real(name).items()}
os.chdir(tempfile.mkdtemp())
for m in models:
    for mode in ("eager", "capture"):
        os.environ["CHAOREC_TRY_CAPTURE"] = m if mode == "capture" else ""
        logging.getLogger().handlers.clear()
        t0 = time.perf_counter()
        try:
            best = cmain.main(["--Model", m, "--data_path", "baby", "--synthetic", "--num_epoch", "3"])
            print(f"{m:10s} {mode:8s} {time.perf_counter() - t0:7.1f} s  recall@20 {best[20]['recall']:.4f}", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{m:10s} {mode:8s} FAILED after {time.perf_counter() - t0:.1f} s: {exc!r}"[:300], flush=True)
