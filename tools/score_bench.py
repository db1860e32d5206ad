#!/usr/bin/env python3
"""Time gene_ranklist's kernel sequence on a dataset-shaped random problem (run under rocprofv3 for a breakdown)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph, ops
from chaorec_amd.synthetic import synthetic_interactions, DATASET_SHAPES
ds = os.environ.get("DATASET", "sports")
U, I, E = DATASET_SHAPES[ds]
dev = torch.device("cuda:0")
edges = synthetic_interactions(U, I, E, seed=42)
hist = tuple(t.to(dev) for t in graph.user_hist_csr_from_edges(edges, U))
if os.environ.get("HIST", "1") == "0":
    hist = None
torch.manual_seed(0)
emb = torch.randn(U + I, 64, device=dev) * 0.1
for prec in [int(x) for x in os.environ.get("PREC", "0,1").split(",")]:
    for _ in range(2):
        ops.score_topk(emb[:U], emb[U:], hist, 1e-6, 50, id_offset=U, precision=prec)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.score_topk(emb[:U], emb[U:], hist, 1e-6, 50, id_offset=U, precision=prec)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    st = {}
    ops.score_topk(emb[:U], emb[U:], hist, 1e-6, 50, id_offset=U, precision=prec, stats=st)
    if st.get("prefilter_users"):
        print("   prefilter:", st, "candidates/user %.1f" % (st["candidates"] / st["prefilter_users"]))
    print(f"{ds} precision={prec}: {ms:.3f} ms  {U / ms * 1e3 / 1e6:.1f} M users/s  {2 * U * I * 64 / ms / 1e9:.1f} TF")
