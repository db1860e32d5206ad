#!/usr/bin/env python3
"""Epoch-level golden of the reference's sampler + loop: the REFERENCE LightGCN trained on Data/baby by the
REFERENCE's own TrainingDataset (dataload.py:61-106: random.sample rejection sampler) under DataLoader(shuffle=True)
(main.py:194-195) and its own train() / gene_ranklist / gene_metrics (train_and_evaluate.py:39-48, 655-659), several seeds;
stored: per seed and epoch the summed batch loss and Recall / NDCG @ 20 on val and test.  The fixed-batch trajectory goldens
cannot see a bias of the product's in-kernel counter sampler, its attempt cap or its per-epoch permutation; this one can:
tests/test_gpu_epoch_parity.py trains chaorec_amd's loop for the same epochs and asks every epoch's numbers to lie inside
the spread of the reference's seeds.

Runs only in the build container (needs /root/reference):    python tests/golden/gen_epoch_parity.py
LightGCN imports torch_geometric, which is not installed: oracle/pyg_standin.py provides the restated propagate (its
docstring), as for the other LightGCN goldens.  Nothing of the reference is copied: inputs and outputs only."""
import os
import random
import sys
import time
import warnings

import numpy as np
import torch

REF = os.environ.get("CHAOREC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyg_standin  # noqa: E402

pyg_standin.install()
sys.path.insert(0, REF)
_argv = sys.argv
sys.argv = ["main.py", "--Model", "LightGCN", "--data_path", "baby"]  # parse_args() runs at import
warnings.filterwarnings("ignore")
from Model.LightGCN import LightGCN  # noqa: E402
import utils as ref_utils  # noqa: E402
import dataload as ref_dataload  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

sys.argv = _argv
torch.set_num_threads(8)
SEEDS = [int(s) for s in os.environ.get("SEEDS", "0,1,2,3,4").split(",")]
EPOCHS = int(os.environ.get("EPOCHS", "12"))
D, L, REG, LR, B, K = 64, 2, 1e-3, 1e-3, 1024, 20          # BASELINE configs[0]; Model_YAML/LightGCN.yaml: lr, reg


def main():
    data = os.path.join(REF, "Data", "baby")
    train = np.load(os.path.join(data, "train.npy"), allow_pickle=True)
    val = np.load(os.path.join(data, "val.npy"), allow_pickle=True)
    test = np.load(os.path.join(data, "test.npy"), allow_pickle=True)
    U, I = 12351, 4794                                       # dataload.py:36-38
    uid = {}
    for u, i in train.tolist():
        uid.setdefault(u, []).append(i)
    names = ["loss", "val_recall", "val_ndcg", "test_recall", "test_ndcg"]
    out = {n: np.zeros((len(SEEDS), EPOCHS)) for n in names}
    for si, seed in enumerate(SEEDS):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        model = LightGCN(U, I, train, uid, D, REG, L, "add", torch.device("cpu"))
        loader = DataLoader(ref_dataload.TrainingDataset(U, I, uid, train), B, shuffle=True, num_workers=0)
        opt = torch.optim.Adam([{"params": model.parameters(), "lr": LR}])
        for ep in range(EPOCHS):
            t0 = time.time()
            model.train()
            s = 0.0
            for users, pos, neg in loader:                   # train_and_evaluate.py:43-48
                opt.zero_grad()
                loss = model.loss(users, pos, neg)
                loss.backward()
                opt.step()
                s += loss.item()
            model.eval()
            with torch.no_grad():
                rank = model.gene_ranklist()
                mv = ref_utils.gene_metrics(val, rank, [K])
                mt = ref_utils.gene_metrics(test, rank, [K])
            row = [s, mv[K]["recall"], mv[K]["ndcg"], mt[K]["recall"], mt[K]["ndcg"]]
            for n, v in zip(names, row):
                out[n][si, ep] = v
            print(f"seed {seed} epoch {ep + 1}: loss {s:.4f} val R@20 {row[1]:.5f} N@20 {row[2]:.5f} "
                  f"test R@20 {row[3]:.5f} ({time.time() - t0:.1f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "lightgcn_epochs_baby.npz"), seeds=np.array(SEEDS), epochs=EPOCHS, D=D, L=L, reg=REG,
                        lr=LR, batch=B, K=K, **out)
    print("wrote lightgcn_epochs_baby.npz")


if __name__ == "__main__":
    main()
