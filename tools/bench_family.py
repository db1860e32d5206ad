"""Epoch times of the widened models on the MI355X: for every model of chaorec_amd.main's table, the first point of its
YAML grid on the REAL sports interactions (tests/golden/; synthetic modality features of the real widths where the model
reads any), two warm epochs then the median of three: seconds per training epoch (155 batches of 1024; captured hipGraph
step or eager, as train_and_evaluate decides), milliseconds per batch, and the full-rank evaluation (gene_ranklist + device
metrics).      python3 tools/bench_family.py [Model ...]  ->  one line per model + a JSON summary"""
import json
import logging
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload, main as cmain, train_and_evaluate as te  # noqa: E402
from chaorec_amd.arg_parser import load_yaml_config, parse_args  # noqa: E402
from chaorec_amd.optim import FusedAdam  # noqa: E402

ALL = ["LightGCN", "NGCF", "LayerGCN", "BPR", "VBPR", "MGCN", "FREEDOM", "MMGCN", "SimGCL", "XSimGCL", "NCL", "SelfCF", "SLMRec", "MCLN",
       "DHCF", "LGMRec", "POWERec", "SMORE", "GUME", "MMGCL", "FKAN_GCF", "VGCL", "DDRec", "DCCF", "MICRO", "LATTICE", "MENTOR", "HCCF",
       "LightGCL", "SGL", "BM3", "MGCL", "MMSSL", "GRCN", "MGAT"]
models = sys.argv[1:] or ALL
os.chdir(tempfile.mkdtemp())
logging.disable(logging.CRITICAL)
dev = torch.device("cuda:0")
out = {}
for name in models:
    try:
        args = parse_args(["--Model", name, "--data_path", "sports"])
        cfg = load_yaml_config(name)
        for p in cfg["hyper_parameters"]:
            setattr(args, p, cfg[p][0])
        cmain.setup_seed(args.seed)
        needs = name in ("MMGCN", "FREEDOM", "MGCN", "VBPR", "SLMRec", "MCLN", "POWERec", "LGMRec", "SMORE", "MMGCL", "LightGT", "GUME",
                         "DDRec", "MICRO", "MENTOR", "BM3", "MGCL", "LATTICE", "MMSSL", "GRCN", "MGAT")
        train, val, test, uid, U, I, v, t = dataload.data_load("sports", has_v=needs, has_t=needs, data_root=args.data_root, synthetic=False)
        loader = dataload.DeviceBatchSampler(U, I, uid, train, args.batch_size, dev, name, args.seed)
        args.num_user, args.num_item = U, I
        model = cmain.build_model(args, U, I, train, uid, v, t, dev).to(dev)
        opt = FusedAdam([{"params": model.parameters(), "lr": args.learning_rate}])
        if name in te.PRE_EPOCH:
            model.pre_epoch_processing()
        graphed = te._capture_step(model, loader, opt, name)
        ep, ev = [], []
        val_l = te.EvalLists(val, dev)
        for k in range(5):
            if name in te.PRE_EPOCH:
                model.pre_epoch_processing()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            te.train(model, loader, opt, name, graphed)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.eval()
            te.evaluate(model, val_l, model.gene_ranklist(to_cpu=False), [5, 10, 20])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if k >= 2:
                ep.append(t1 - t0)
                ev.append(t2 - t1)
        nb = len(loader)
        out[name] = {"epoch_s": float(np.median(ep)), "ms_per_batch": float(np.median(ep)) / nb * 1e3, "eval_ms": float(np.median(ev)) * 1e3,
                     "captured": graphed is not None, "batches": nb}
        print(f"{name:10s} epoch {out[name]['epoch_s']:7.3f} s  {out[name]['ms_per_batch']:7.3f} ms/batch  eval {out[name]['eval_ms']:7.2f} ms  "
              f"{'captured' if graphed is not None else 'eager'}", flush=True)
        del model, opt, graphed, loader
        torch.cuda.empty_cache()
    except Exception as exc:  # noqa: BLE001
        out[name] = {"error": repr(exc)[:200]}
        print(f"{name:10s} FAILED {exc!r}"[:250], flush=True)
print(json.dumps(out))
