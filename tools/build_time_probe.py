"""Where the host-side build of BASELINE configs[4] whole goes: edge list, CSR, history, tables, schedule."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import graph  # noqa: E402
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions_torch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
dev = torch.device("cuda:0")
U, I, E = DATASET_SHAPES[name]


def timed(what, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{what:48s} {time.perf_counter() - t0:7.2f} s", flush=True)
    return out


edges = timed("edge list (device)", lambda: synthetic_interactions_torch(U, I, E, seed=42, device="cuda"))
csr = timed("lightgcn_csr (device)", lambda: graph.lightgcn_csr(edges, U + I).to(dev))
hist = timed("user_hist_csr_from_edges (device)", lambda: graph.user_hist_csr_from_edges(edges, U))
torch.manual_seed(42)
emb = timed("nn.Embedding x2 + xavier on the HOST", lambda: [torch.nn.init.xavier_uniform_(torch.nn.Embedding(n, 128).weight) for n in (U, I)])
emb_d = timed("... copied to the device", lambda: [e.to(dev) for e in emb])
del emb, emb_d
emb2 = timed("nn.Embedding x2 + xavier ON THE DEVICE", lambda: [torch.nn.init.xavier_uniform_(torch.nn.Embedding(n, 128, device=dev).weight) for n in (U, I)])
del emb2
for flag in ("1", "0"):
    graph.SCHEDULE_ON_DEVICE = flag == "1"
    csr._orders.clear()
    timed(f"schedule(128), SCHEDULE_ON_DEVICE={flag}", lambda: csr.schedule(128))
