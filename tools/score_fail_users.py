"""Which users fail pass A one epoch after their thresholds were taken, and why the exact route takes what it takes for
them: history length, number of unmasked items above the old threshold.  (LightGCN/sports, torch arithmetic.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload, ops  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402

dev = torch.device("cuda:0")
d = dataload.packed_interactions("sports")
U, I, edges = d["num_user"], d["num_item"], d["train"]
torch.manual_seed(42)
m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
step = FusedLightGCNStep(m, opt, batch_size=1024, edges=edges_dev, seed=42, step_dev=cnt, steps_per_replay=5)
step.run(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
res = m.result.detach().clone()
old = torch.empty(U, dtype=torch.float32, device=dev)
ops.score_topk(res[:U], res[U:U + I], m.hist, 1e-6, 50, id_offset=U, hint=old, hint_valid=False, hint_rank=100)
step.run(155)
res = m.result.detach().clone()
ue, ie = res[:U], res[U:U + I]
rowptr, col = m.hist
deg = (rowptr[1:] - rowptr[:-1]).cpu()
print("history length: max", int(deg.max()), "99.9 %", int(torch.quantile(deg.float(), 0.999)), "users > 128:", int((deg > 128).sum()),
      "> 1024:", int((deg > 1024).sum()))
S = ue @ ie.T
rows = torch.repeat_interleave(torch.arange(U, device=dev), rowptr[1:] - rowptr[:-1])
S[rows, col.long()] = -1e30
above = (S > old[:, None]).sum(1).cpu()
bad = torch.nonzero(above < 50).flatten()
print("users with fewer than 50 unmasked items above the epoch-old rank-100 threshold:", bad.tolist())
for u in bad.tolist():
    print("  user", u, "history", int(deg[u]), "items above", int(above[u]))
counters = torch.zeros(4, dtype=torch.int32, device=dev)
hint = old.clone()
for light in (True, False):
    ts = []
    for _ in range(6):
        hint.copy_(old)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=100, light=light, counters=counters)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    print("light" if light else "full ", "%.1f us" % float(np.median(ts)), counters.tolist())
