"""FREEDOM on baby: epoch-1 / epoch-2 test Recall@20 of the product for many seeds, beside the reference's ten (the epoch-parity
golden): is the product's first epoch distributed like the reference's?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_interactions
from chaorec_amd import dataload, graph
from chaorec_amd.Model import FREEDOM
from chaorec_amd.optim import FusedAdam
from chaorec_amd.train_and_evaluate import train_and_evaluate
from chaorec_amd.utils import setup_seed
import logging; logging.disable(logging.CRITICAL)
g = load_golden("freedom_epochs_baby.npz"); d = load_interactions("baby")
U, I, train = d["U"], d["I"], d["train"]; dev = torch.device("cuda:0")
val = np.array(d["val"], dtype=object); test = np.array(d["test"], dtype=object)
uid = graph.user_item_dict_from_edges(train)
fg = torch.Generator().manual_seed(int(g["feat_seed"]))
v_feat = torch.randn(I, int(g["dv"]), generator=fg); t_feat = torch.randn(I, int(g["dt"]), generator=fg)
print("reference epoch 1:", np.sort(g["test_recall"][:, 0]).round(4).tolist())
print("reference epoch 2:", np.sort(g["test_recall"][:, 1]).round(4).tolist())
print("reference loss 1:", np.sort(g["loss"][:, 0]).round(3).tolist())
graphed = os.environ.get("GRAPH", "1") == "1"
out = []
for seed in [int(s) for s in os.environ.get("SEEDS", "1,2,3,7,42,99,5,6,8,9,10,11,12,13,14,15").split(",")]:
    setup_seed(seed)
    model = FREEDOM(U, I, train, uid, v_feat.clone(), t_feat.clone(), 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev).to(dev)
    loader = dataload.DeviceBatchSampler(U, I, uid, train, 1024, dev, "FREEDOM", seed)
    opt = FusedAdam([{"params": model.parameters(), "lr": 1e-3}])
    hist = []
    train_and_evaluate(model, loader, val, test, opt, 2, model_name="FREEDOM", topk=(5, 10, 20), patience=10 ** 6, history=hist, graph=graphed)
    out.append((seed, round(hist[0]["test"][20]["recall"], 4), round(hist[1]["test"][20]["recall"], 4), round(hist[0]["loss"], 3)))
    print(out[-1], flush=True)
print("product epoch 1:", sorted(o[1] for o in out))
print("product epoch 2:", sorted(o[2] for o in out))
