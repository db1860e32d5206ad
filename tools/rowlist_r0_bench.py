#!/usr/bin/env python3
"""The light step's launch over a BATCH's rows (forward layer L over R0's list + layer mean) alone, on the synthetic
config-5 graphs, for several stripe thresholds (CHAOREC_ROWLIST_STRIPE_T; 0 = no column stripes): what the launch costs,
how long its longest rows are.  Measurement only.    python tools/rowlist_r0_bench.py config5_shard|config5 [D]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.common import load_graph  # noqa: E402
from chaorec_amd import _lib, graph, ops  # noqa: E402

_lib.ensure_built()
dev = torch.device("cuda:0")
dataset = sys.argv[1] if len(sys.argv) > 1 else "config5_shard"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = 1024
edges, U, I, _ = load_graph(dataset, True)
N = U + I
ed = edges if torch.is_tensor(edges) else torch.from_numpy(edges).to(dev)
csr = graph.lightgcn_csr(edges, N)
csr = csr.to(dev) if not csr.rowptr.is_cuda else csr
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g) * 0.1
t0 = torch.randn(N, D, device=dev, generator=g) * 0.1
out = torch.empty(N, D, device=dev)
idx = torch.randint(0, ed.shape[0], (B,), device=dev, generator=g)
ids = (ed[idx, 0].long(), ed[idx, 1].long() - U, torch.randint(0, I, (B,), device=dev, generator=g))
bits = ops.row_bitmap(N, dev)
list0, n0 = torch.empty(3 * B, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
ops.batch_rows(ids, bits, U, list0, n0)
rows = list0[:int(n0)].long()
deg = (csr.rowptr[rows + 1] - csr.rowptr[rows]).cpu().numpy()
print(f"{dataset} D={D}: {len(rows)} rows, {int(deg.sum())} entries; rows > 256: {(deg > 256).sum()} ({int(deg[deg > 256].sum())} entries), "
      f"> 8192: {(deg > 8192).sum()} ({int(deg[deg > 8192].sum())} entries); longest {np.sort(deg)[-6:][::-1].tolist()}")
long_rows = ops.long_row_buffers(csr)
for st in (0, 2048, 8192, 32768):
    os.environ["CHAOREC_ROWLIST_STRIPE_T"] = str(st)

    def once():
        ops.spmm_rowlist_raw(csr, x, None, list0, n0, mean_out=out, mean_terms=[t0, x], mean_w=1.0 / 3.0, long_rows=long_rows)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in ev:
        s.record()
        once()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    by = float(deg.sum()) * (4 * D + 8) + len(rows) * (4 * D + 8)
    print(f"   stripe_t {st:6d}: {ts[len(ts) // 2] * 1e3:8.1f} us (min {ts[0] * 1e3:.1f})   {by / ts[len(ts) // 2] / 1e6 / 8000:.3f} of 8 TB/s")
