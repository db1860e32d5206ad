#!/usr/bin/env python3
"""Train every model for a few epochs through the real entry point (captured steps, device evaluation) on a synthetic
dataset-shaped graph and print the per-epoch losses: they must stay finite and go down."""
import logging, os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import main as cmain, dataload
dataload.SYNTHETIC_FEATURE_DIMS["default"] = (96, 64)
models = sys.argv[1:] or ["LightGCN", "NGCF", "LayerGCN", "FREEDOM", "MGCN", "MMGCN"]
for m in models:
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        logging.getLogger().handlers.clear()
        best = cmain.main(["--Model", m, "--data_path", "baby", "--synthetic", "--num_epoch", "6"])
        log = open(os.path.join(d, "log", f"{m}_baby.log")).read()
        losses = [float(x) for x in re.findall(r"Epoch \d+, Loss: ([0-9.eE+-]+|nan|inf)", log)]
        ok = all(l == l and l < 1e6 for l in losses) and losses[-1] < losses[0]
        print(f"{m:9s} losses/epoch {['%.3f' % l for l in losses]}  recall@20 {best[20]['recall']:.4f}  {'OK' if ok else 'SUSPECT'}")
