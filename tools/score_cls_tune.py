"""The cold ranking call on the config-5 shard's PROPAGATED tables (model.result after a few training steps: norms that fall
with the degree) under variants of the sorted layout's sampler (CHAOREC_PF_CLS_STRIDE / _RANK; 'own' = the table's own
order): one model build, every variant timed in the same process.  python3 tools/score_cls_tune.py [dataset] [dim]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chaorec_amd import ops  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402

ds = sys.argv[1] if len(sys.argv) > 1 else "config5_shard"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
edges, U, I, _ = bench.load_graph(ds, False)
torch.manual_seed(42)
model = LightGCN(U, I, edges, None, D, 1e-3, 3, "add", dev).to(dev)
model.graph.schedule(D)
opt = FusedAdam(model.parameters(), lr=1e-3)
import numpy as np  # noqa: E402
edges_dev = edges.to(torch.int64) if torch.is_tensor(edges) else torch.from_numpy(edges.astype(np.int64)).to(dev)
stepper = FusedLightGCNStep(model, opt, batch_size=1024, edges=edges_dev, seed=42, capture=False)
stepper.run(4, full_last=True)
res = model.result.detach()
ue, ie = res[:U], res[U:U + I]
nrm = ie.norm(dim=1)
q = torch.quantile(nrm[::16].float(), torch.tensor([0.0, 0.01, 0.5, 0.99, 1.0], device=dev))
print("item norms min/1%/median/99%/max", [f"{x:.4g}" for x in q.tolist()], flush=True)
variants = [("own", None, None)] + [("sorted", s, r) for s, r in ((16, 7), (16, 5), (32, 5), (32, 4), (32, 3), (24, 5), (48, 3), (64, 3))]
base = None
for name, s, r in variants:
    os.environ["CHAOREC_PF_CLS_MIN_ITEMS"] = "0" if name == "own" else "131072"
    for k, v in (("CHAOREC_PF_CLS_STRIDE", s), ("CHAOREC_PF_CLS_RANK", r)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    st = {}
    with torch.no_grad():
        i0, v0 = ops.score_topk(ue, ie, model.hist, 1e-6, 50, id_offset=U, stats=st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.score_topk(ue, ie, model.hist, 1e-6, 50, id_offset=U)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
    if base is None:
        base = (i0, v0)
    same = torch.equal(i0, base[0]) and torch.equal(v0, base[1])
    ms = min(ts)
    print(f"{name:6s} stride {s} rank {r}: {ms:8.2f} ms  frac {2.0 * U * I * D / ms / 1e9 / 2500:.4f}  cand/user {st['candidates'] / U:6.1f} "
          f"reth {st['rethreshold_users']} fallback {st['fallback_users']} {st['fallback_reasons']} same_as_own {same}", flush=True)
    del i0, v0
