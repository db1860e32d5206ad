"""FREEDOM / baby, epoch 1: the PRODUCT model on the GPU fed with batches made the reference's way on the host (torch.randperm
over the edges, python `random` rejection sampling per edge) against the same model fed by dataload.DeviceBatchSampler: which
side of the loop carries the wider spread of the first epoch's recall?"""
import os, random, sys
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
U, I, train = d["U"], d["I"], np.asarray(d["train"]); dev = torch.device("cuda:0")
val = np.array(d["val"], dtype=object); test = np.array(d["test"], dtype=object)
uid = graph.user_item_dict_from_edges(train)
uset = {u: set(v) for u, v in uid.items()}
fg = torch.Generator().manual_seed(int(g["feat_seed"]))
v_feat = torch.randn(I, int(g["dv"]), generator=fg); t_feat = torch.randn(I, int(g["dt"]), generator=fg)
all_items = list(range(U, U + I))


from chaorec_amd import ops


class MixedLoader:
    """host permutation + device negatives (mode "hp_dn") or device permutation + host negatives ("dp_hn")."""
    def __init__(self, seed, mode):
        self.seed, self.mode = seed, mode
        rowptr, col = graph.user_hist_csr(uid, U)
        self.hist = (rowptr.to(dev), col.to(dev))
        self.edges = torch.from_numpy(train.astype(np.int64)).to(dev)
    def __len__(self):
        return (len(train) + 1023) // 1024
    def __iter__(self):
        if self.mode == "hp_dn":
            perm = torch.randperm(len(train), generator=torch.Generator().manual_seed(self.seed)).to(dev)
        else:
            gen = torch.Generator(device=dev); gen.manual_seed(self.seed)
            perm = torch.randperm(len(train), device=dev, generator=gen)
        rnd = random.Random(self.seed)
        step = 0
        for s in range(0, perm.numel(), 1024):
            e = self.edges[perm[s:s + 1024]]
            users, pos = e[:, 0].contiguous(), e[:, 1].contiguous()
            if self.mode == "hp_dn":
                neg = ops.sample_negatives(self.hist, users, I, self.seed, step, U)
            else:
                neg = []
                for u in users.tolist():
                    while True:
                        n = all_items[rnd.randrange(I)]
                        if n not in uset[int(u)]:
                            break
                    neg.append(n)
                neg = torch.tensor(neg, dtype=torch.int64, device=dev)
            step += 1
            yield users, pos, neg


class HostLoader:
    """dataload.py:61-106 + DataLoader(shuffle=True), batch 1024: one pass over a permutation of the edges, negatives by rejection."""
    def __init__(self, seed):
        self.seed = seed
    def __len__(self):
        return (len(train) + 1023) // 1024
    def __iter__(self):
        perm = torch.randperm(len(train), generator=torch.Generator().manual_seed(self.seed)).numpy()
        rnd = random.Random(self.seed)
        for s in range(0, len(perm), 1024):
            e = train[perm[s:s + 1024]]
            neg = []
            for u in e[:, 0]:
                while True:
                    n = all_items[rnd.randrange(I)]
                    if n not in uset[int(u)]:
                        break
                neg.append(n)
            yield (torch.from_numpy(e[:, 0].astype(np.int64)).to(dev), torch.from_numpy(e[:, 1].astype(np.int64)).to(dev),
                   torch.tensor(neg, dtype=torch.int64, device=dev))


def run(samp_seed, host, mode=None):
    setup_seed(1)
    model = FREEDOM(U, I, train, uid, v_feat.clone(), t_feat.clone(), 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev).to(dev)
    model._prune_seed = 7
    loader = MixedLoader(samp_seed, mode) if mode else (HostLoader(samp_seed) if host else dataload.DeviceBatchSampler(U, I, uid, train, 1024, dev, "FREEDOM", samp_seed))
    opt = FusedAdam([{"params": model.parameters(), "lr": 1e-3}])
    hist = []
    train_and_evaluate(model, loader, val, test, opt, 1, model_name="FREEDOM", topk=(5, 10, 20), patience=10 ** 6, history=hist, graph=False)
    return hist[0]["test"][20]["recall"]


N = int(os.environ.get("N", "30"))
for base, stride in ((100, 1), (1, 1), (1000, 7)):
    for name, host, mode in (("host perm + host negs", True, None), ("DeviceBatchSampler", False, None)):
        a = np.array([run(base + stride * s, host, mode) for s in range(N)])
        print(f"seeds {base}+{stride}s  {name:24s} n={N} mean {a.mean():.5f} std {a.std(ddof=1):.5f}  min {a.min():.4f} max {a.max():.4f}", flush=True)
