"""How many more candidates would a GLOBAL item-norm bound (c ||u|| max||i||) admit than the per-item bound (c ||u|| ||i_j||)
the sweep uses today?  Trained LightGCN/sports tables (5000 steps), thresholds = exact score of rank 110 (a carried hint)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
d = dataload.packed_interactions("sports")
U, I, edges = d["num_user"], d["num_item"], d["train"]
torch.manual_seed(42)
m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
step = FusedLightGCNStep(m, opt, batch_size=1024, edges=edges_dev, seed=42, step_dev=cnt, steps_per_replay=5)
step.run(steps)
res = m.result.detach()
ue, ie = res[:U], res[U:U + I]
c = 1.05 / 256
nu, ni = ue.norm(dim=1), ie.norm(dim=1)
print("item norms: min %.4f median %.4f p99 %.4f max %.4f" % (ni.min(), ni.median(), ni.quantile(0.99), ni.max()))
tot = {"per_item": 0, "global_max": 0, "global_p999": 0, "none": 0}
for u0 in range(0, U, 4096):
    s = ue[u0:u0 + 4096] @ ie.T
    T = s.topk(110, dim=1).values[:, -1:]
    band = c * nu[u0:u0 + 4096, None]
    tot["none"] += int((s > T).sum())
    tot["per_item"] += int((s + band * ni[None, :] > T).sum())
    tot["global_max"] += int((s + band * ni.max() > T).sum())
    tot["global_p999"] += int((s + band * ni.quantile(0.999) > T).sum())
print({k: v / U for k, v in tot.items()})
