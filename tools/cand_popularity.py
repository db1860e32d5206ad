"""How concentrated the ranking's candidates are: share of the users' top-100 slots taken by the most frequent items
(LightGCN/sports after training) -- the case for an LDS-resident subset of the item table in the exact re-score."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402
dev = torch.device("cuda:0")
d = dataload.packed_interactions("sports")
U, I, edges = d["num_user"], d["num_item"], d["train"]
torch.manual_seed(42)
m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
step = FusedLightGCNStep(m, opt, batch_size=1024, edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42, step_dev=cnt, steps_per_replay=5)
for steps in (300, 3000):
    step.run(steps - int(cnt))
    res = m.result.detach()
    S = res[:U] @ res[U:U + I].T
    top = torch.topk(S, 100, dim=1).indices.flatten()
    c = torch.bincount(top, minlength=I).sort(descending=True).values.double()
    tot = float(c.sum())
    print(f"after {steps} steps: share of the top-100 slots held by the most frequent items: " +
          ", ".join(f"{n}: {float(c[:n].sum()) / tot:.3f}" for n in (64, 128, 256, 512, 1024, 2048, 4096)))
    cnt_item = torch.bincount(top, minlength=I).double()
    for name, key in (("norm", res[U:U + I].norm(dim=1)), ("item degree", torch.bincount(torch.from_numpy(edges[:, 1].astype(np.int64) - U).to(dev), minlength=I).double()),
                      ("mean score", S.mean(0))):
        order = torch.argsort(key, descending=True)
        print(f"    items ranked by {name}: " + ", ".join(f"{n}: {float(cnt_item[order[:n]].sum()) / tot:.3f}" for n in (128, 256, 384, 512, 1024)))
