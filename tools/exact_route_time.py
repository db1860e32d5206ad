"""Cost of the exact per-user route against the number of users on it: LightGCN/sports, carried thresholds on the same
tables, the first n users' thresholds set to +inf (no candidates -> they fail pass A and, in light mode, take the
exact route).  Whole-call event times; the difference to n = 0 is the route's cost.
Stage cuts (experiment builds, wrong results): CHAOREC_EXTRA_HIPCC_FLAGS=-DCHAOREC_EX_EXP=k with k = 1 scores only,
2 + per-wave selection, 3 + block merge (no slice merge), 4 scores without the history search, 5 history search without
scores.  Round 3, one user: scores 22 of the route's 30-35 us (a chain of dependent gathers: id -> row pointer -> rows,
then 8 passes of 16 rows per wave), block merge 6.7, slice merge 6.4."""
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
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
step = FusedLightGCNStep(m, opt, batch_size=1024, edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42, step_dev=cnt,
                         steps_per_replay=5)
step.run(int(sys.argv[1]) if len(sys.argv) > 1 else 500)
res = m.result.detach().clone()
ue, ie = res[:U], res[U:U + I]
old = torch.empty(U, dtype=torch.float32, device=dev)
ref_idx, ref_val = ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=old, hint_valid=False, hint_rank=100)
counters = torch.zeros(4, dtype=torch.int32, device=dev)
hint = old.clone()
base = None
for n in (0, 1, 3, 8, 16, 64, 256):
    ts = []
    for rep in range(7):
        hint.copy_(old)
        hint[:n] = float("inf")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        idx, val = ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=100, light=True,
                                  counters=counters)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    t = float(np.median(ts))
    base = t if base is None else base
    ok = torch.equal(idx, ref_idx) and torch.equal(val, ref_val)
    print(f"users on the exact route {n:4d} (queues {counters.tolist()}): call {t:7.1f} us, route {t - base:6.1f} us, result {'identical' if ok else 'DIFFERS'}")
