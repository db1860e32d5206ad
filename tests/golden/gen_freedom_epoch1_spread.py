#!/usr/bin/env python3
"""How widely the REFERENCE FREEDOM's metrics are spread after the FIRST epoch (companion of gen_epoch_parity_mm.py).

After one epoch Recall / NDCG @ 20 sit on the steepest part of the learning curve (0.005 at initialisation, 0.054 after epoch 1,
0.067 after epoch 2) and their distribution over the sampling seeds is left-skewed: the ten seeds of freedom_epochs_baby.npz
give a standard deviation of 0.0011 for the test recall, thirty give 0.0017 with a minimum at 0.050.  The epoch-parity test
takes the first epoch's spread from HERE: the reference class with one initialisation (seed 1) under thirty sampling seeds
(DataLoader shuffle, negative sampler, pruning draw), one epoch each, the reference's own gene_ranklist / gene_metrics.

Runs only in the build container (needs /root/reference):    python tests/golden/gen_freedom_epoch1_spread.py     (~5 min)
Nothing of the reference is copied: inputs and outputs only."""
import os
import random
import sys
import warnings

import numpy as np
import torch

REF = os.environ.get("CHAOREC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
_argv = sys.argv
sys.argv = ["main.py", "--Model", "FREEDOM", "--data_path", "baby"]
warnings.filterwarnings("ignore")
from Model.FREEDOM import FREEDOM  # noqa: E402
import utils as ref_utils  # noqa: E402
import dataload as ref_dataload  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

sys.argv = _argv
torch.set_num_threads(int(os.environ.get("THREADS", "6")))
N = int(os.environ.get("N", "30"))


def main():
    data = os.path.join(REF, "Data", "baby")
    train = np.load(os.path.join(data, "train.npy"), allow_pickle=True)
    val = np.load(os.path.join(data, "val.npy"), allow_pickle=True)
    test = np.load(os.path.join(data, "test.npy"), allow_pickle=True)
    U, I = 12351, 4794
    uid = {}
    for u, i in train.tolist():
        uid.setdefault(u, []).append(i)
    g = torch.Generator().manual_seed(5)
    v, t = torch.randn(I, 128, generator=g), torch.randn(I, 64, generator=g)
    names = ["val_recall", "val_ndcg", "test_recall", "test_ndcg"]
    out = {n: np.zeros(N) for n in names}
    for s in range(N):
        random.seed(1)
        np.random.seed(1)
        torch.manual_seed(1)
        m = FREEDOM(U, I, train, uid, v.clone(), t.clone(), 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, torch.device("cpu"))
        random.seed(100 + s)
        np.random.seed(100 + s)
        torch.manual_seed(100 + s)
        loader = DataLoader(ref_dataload.TrainingDataset(U, I, uid, train), 1024, shuffle=True, num_workers=0)
        opt = torch.optim.Adam([{"params": m.parameters(), "lr": 1e-3}])
        m.pre_epoch_processing()
        m.train()
        for b in loader:
            opt.zero_grad()
            loss = m.loss(*b)
            loss.backward()
            opt.step()
        m.eval()
        with torch.no_grad():
            r = m.gene_ranklist()
            mv, mt = ref_utils.gene_metrics(val, r, [20]), ref_utils.gene_metrics(test, r, [20])
        row = [mv[20]["recall"], mv[20]["ndcg"], mt[20]["recall"], mt[20]["ndcg"]]
        for n, x in zip(names, row):
            out[n][s] = x
        print(s, [round(x, 5) for x in row], flush=True)
    np.savez_compressed(os.path.join(HERE, "freedom_epoch1_spread_baby.npz"), n=N, init_seed=1, sampling_seeds=100 + np.arange(N), **out)
    print({n: (round(float(a.mean()), 5), round(float(a.std(ddof=1)), 5), round(float(a.min()), 4)) for n, a in out.items()})


if __name__ == "__main__":
    main()
