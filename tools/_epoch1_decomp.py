"""FREEDOM / baby, epoch 1: where the spread of the product's test Recall@20 comes from -- model initialisation or sampling."""
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


def run(init_seed, samp_seed, lazy=None, prune=True, prune_seed=None):
    setup_seed(init_seed)
    model = FREEDOM(U, I, train, uid, v_feat.clone(), t_feat.clone(), 64, 64, 1e-3, 0.1 if prune else 0.0, 2, 1, 10, 0.8, dev).to(dev)
    setup_seed(samp_seed)
    model._prune_seed = samp_seed if prune_seed is None else prune_seed
    loader = dataload.DeviceBatchSampler(U, I, uid, train, 1024, dev, "FREEDOM", samp_seed)
    opt = FusedAdam([{"params": model.parameters(), "lr": 1e-3}], lazy_rows=lazy)
    hist = []
    train_and_evaluate(model, loader, val, test, opt, 1, model_name="FREEDOM", topk=(5, 10, 20), patience=10 ** 6, history=hist, graph=False)
    return hist[0]["test"][20]["recall"]


a = [run(1, 100 + s, prune_seed=7) for s in range(8)]
print("product fixed init, FIXED pruning, varying batches:", np.round(sorted(a), 4), "std", np.std(a, ddof=1).round(5), flush=True)
b = [run(1, 100, prune_seed=200 + s) for s in range(8)]
print("product fixed init, fixed batches, VARYING pruning:", np.round(sorted(b), 4), "std", np.std(b, ddof=1).round(5), flush=True)
