#!/usr/bin/env python3
"""Epoch-level goldens of the reference's OTHER two loop branches, the companions of gen_epoch_parity.py:

  MODEL=FREEDOM   the reference FREEDOM (Model/FREEDOM.py, pure torch: no stand-in of any kind) on Data/baby -- every epoch its own
                  pre_epoch_processing() (train_and_evaluate.py:555-557 -> Model/FREEDOM.py:143-162: torch.multinomial edge
                  pruning, the second place besides the sampler where the product's RNG parity is distributional by design), then
                  the generic (users, pos, neg) branch of train() over the reference's TrainingDataset / DataLoader(shuffle=True);
  MODEL=MMGCN     the reference MMGCN on Data/baby through the [B, 2] branch of train() (train_and_evaluate.py:32-38) with
                  TrainingDataset's (LongTensor([u, u]), LongTensor([pos, neg])) items (dataload.py:86-88).  MMGCN imports
                  torch_geometric: oracle/pyg_standin.py provides the restated propagate, as for the other MMGCN goldens.

Both with seeded synthetic modality features (the reference's feature blobs are not in the mount) of SMALL widths, so that a
CPU epoch of the reference takes seconds: v_feat [I, 128], t_feat [I, 64] = torch.randn under Generator().manual_seed(5) --
the GPU test regenerates the same tensors.  Stored per seed and epoch: summed batch loss, Recall / NDCG @ 20 on val and test
(the reference's own gene_ranklist + utils.gene_metrics).  Hyper-parameters: Model_YAML/{FREEDOM,MMGCN}.yaml.

Runs only in the build container (needs /root/reference):    MODEL=FREEDOM python tests/golden/gen_epoch_parity_mm.py
(freedom_epochs_baby.npz: SEEDS=0,1,2,3,4,5 then APPEND=1 SEEDS=6,7,8,9 -- ten seeds; mmgcn_epochs_baby.npz: the default six,
 a CPU epoch of the reference MMGCN takes 35 s)
Nothing of the reference is copied: inputs and outputs only."""
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
MODEL = os.environ.get("MODEL", "FREEDOM")
sys.path.insert(0, ROOT)
if MODEL == "MMGCN":
    from oracle import pyg_standin  # noqa: E402
    pyg_standin.install()
sys.path.insert(0, REF)
_argv = sys.argv
sys.argv = ["main.py", "--Model", MODEL, "--data_path", "baby"]  # parse_args() runs at import (dataload.py reads args.Model)
warnings.filterwarnings("ignore")
if MODEL == "MMGCN":
    from Model.MMGCN import MMGCN as RefModel  # noqa: E402
else:
    from Model.FREEDOM import FREEDOM as RefModel  # noqa: E402
import utils as ref_utils  # noqa: E402
import dataload as ref_dataload  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

sys.argv = _argv
torch.set_num_threads(int(os.environ.get("THREADS", "8")))
SEEDS = [int(s) for s in os.environ.get("SEEDS", "0,1,2,3,4,5").split(",")]
EPOCHS = int(os.environ.get("EPOCHS", "12"))
D, LR, B, K = 64, 1e-3, 1024, 20
DV, DT, FEAT_SEED = 128, 64, 5


def features(I):
    g = torch.Generator().manual_seed(FEAT_SEED)
    return torch.randn(I, DV, generator=g), torch.randn(I, DT, generator=g)


def main():
    data = os.path.join(REF, "Data", "baby")
    train = np.load(os.path.join(data, "train.npy"), allow_pickle=True)
    val = np.load(os.path.join(data, "val.npy"), allow_pickle=True)
    test = np.load(os.path.join(data, "test.npy"), allow_pickle=True)
    U, I = 12351, 4794                                       # dataload.py:36-38
    uid = {}
    for u, i in train.tolist():
        uid.setdefault(u, []).append(i)
    v_feat, t_feat = features(I)
    names = ["loss", "val_recall", "val_ndcg", "test_recall", "test_ndcg"]
    out = {n: np.zeros((len(SEEDS), EPOCHS)) for n in names}
    dev = torch.device("cpu")
    for si, seed in enumerate(SEEDS):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if MODEL == "MMGCN":                                  # main.py:261-263, Model_YAML/MMGCN.yaml
            reg = 1e-4
            model = RefModel(U, I, train, uid, v_feat.clone(), t_feat.clone(), D, reg, "add", "False", True, dev)
        else:                                                 # main.py:287-289, Model_YAML/FREEDOM.yaml
            reg = 1e-3
            model = RefModel(U, I, train, uid, v_feat.clone(), t_feat.clone(), D, 64, reg, 0.1, 2, 1, 10, 0.8, dev)
        loader = DataLoader(ref_dataload.TrainingDataset(U, I, uid, train), B, shuffle=True, num_workers=0)
        opt = torch.optim.Adam([{"params": model.parameters(), "lr": LR}])
        for ep in range(EPOCHS):
            t0 = time.time()
            if MODEL == "FREEDOM":
                model.pre_epoch_processing()                 # train_and_evaluate.py:555-557
            model.train()
            s = 0.0
            for batch in loader:                             # train_and_evaluate.py:32-48
                opt.zero_grad()
                loss = model.loss(*batch)
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
            print(f"{MODEL} seed {seed} epoch {ep + 1}: loss {s:.4f} val R@20 {row[1]:.5f} N@20 {row[2]:.5f} "
                  f"test R@20 {row[3]:.5f} ({time.time() - t0:.1f} s)", flush=True)
    path = os.path.join(HERE, f"{MODEL.lower()}_epochs_baby.npz")
    seeds = np.array(SEEDS)
    if os.environ.get("APPEND") == "1" and os.path.exists(path):     # more seeds for an existing file (same settings)
        old = np.load(path)
        assert int(old["epochs"]) == EPOCHS and int(old["dv"]) == DV and int(old["feat_seed"]) == FEAT_SEED
        seeds = np.concatenate([old["seeds"], seeds])
        out = {n: np.concatenate([old[n], out[n]], 0) for n in names}
    np.savez_compressed(path, seeds=seeds, epochs=EPOCHS, D=D, reg=reg, lr=LR, batch=B, K=K, dv=DV, dt=DT,
                        feat_seed=FEAT_SEED, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
