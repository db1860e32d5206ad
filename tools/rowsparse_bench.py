#!/usr/bin/env python3
"""Time the row-sparse backward launches against the dense ones on a config-5-shard-sized graph:
    python3 tools/rowsparse_bench.py [dataset] [D]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import _lib, graph, ops  # noqa: E402
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions  # noqa: E402

_lib.ensure_built()
_lib.load()
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "config5_shard"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
U, I, E = DATASET_SHAPES[name]
edges = synthetic_interactions(U, I, E, seed=42)
N = U + I
csr = graph.lightgcn_csr(edges, N).to(dev)
csr.schedule(D)
gen = torch.Generator(device=dev).manual_seed(1)
B = 1024
# a REAL BPR batch: B training edges picked uniformly (users and positives in proportion to their degree -- popular items),
# one uniform negative each (chaorec_draw_batch)
hist = graph.user_hist_csr_from_edges(edges, U)
hist = (hist[0].to(dev), hist[1].to(dev))
edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
bu, bp, bn = ops.draw_batch(edges_dev, hist, B, U, I, 42, 3)
rows = torch.unique(torch.cat((bu, U + bp, U + bn)))
G = torch.zeros(N, D, device=dev)
G[rows] = torch.randn(rows.numel(), D, device=dev, generator=gen)
bits = [ops.row_bitmap(N, dev) for _ in range(3)]
w = np.zeros(bits[0].numel(), np.uint32)
for r in rows.cpu().tolist():
    w[r >> 5] |= np.uint32(1 << (r & 31))
bits[0].copy_(torch.from_numpy(w.view(np.int32)))
y1, y2, yd = torch.empty_like(G), torch.empty_like(G), torch.empty_like(G)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def pop(b):
    return int(np.unpackbits(b.cpu().numpy().view(np.uint8)).sum())


print(f"{name}: N={N} nnz={csr.nnz} D={D}; |R0|={rows.numel()}")
print(f"dense  y = A G + G          {timed(lambda: ops.spmm_raw(csr, G, y=yd, alpha=0.25, z=G, beta=0.25)):9.3f} ms")
print(f"expand R0 -> N1             {timed(lambda: ops.expand_row_bits(csr, bits[0], bits[1])):9.3f} ms   |N1|={pop(bits[1])}")
print(f"expand N1 -> N2             {timed(lambda: ops.expand_row_bits(csr, bits[1], bits[2])):9.3f} ms   |N2|={pop(bits[2])}")
print(f"sparse #1 entry bits only   {timed(lambda: ops.spmm_rowsparse_raw(csr, G, y1, alpha=0.25, z=G, beta=0.25, src_bits=bits[0], z_bits=bits[0])):9.3f} ms")
print(f"sparse #1 row mask, no zero {timed(lambda: ops.spmm_rowsparse_raw(csr, G, y1, alpha=0.25, z=G, beta=0.25, src_bits=bits[0], z_bits=bits[0], row_bits=bits[1], write_zeros=False)):9.3f} ms")
print(f"sparse #1 row mask + zeros  {timed(lambda: ops.spmm_rowsparse_raw(csr, G, y2, alpha=0.25, z=G, beta=0.25, src_bits=bits[0], z_bits=bits[0], row_bits=bits[1], write_zeros=True)):9.3f} ms")
print("  equal to dense:", bool(torch.equal(y2, yd)))
lst, ln = torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
b1 = ops.row_bitmap(N, dev)
ops.expand_row_bits(csr, bits[0], b1, lst, ln)
yl = torch.full_like(G, 3.0)
print(f"row list  (|list|={int(ln)})     {timed(lambda: ops.spmm_rowlist_raw(csr, G, yl, lst, ln, alpha=0.25, z=G, beta=0.25, src_bits=bits[0], z_bits=bits[0])):9.3f} ms")
sel = lst[:int(ln)].long()
print("  listed rows equal to dense:", bool(torch.equal(yl[sel], yd[sel])), " others untouched:", int((yl != 3.0).any(1).sum()) == int(ln))


def both():
    ln.zero_(); b1.zero_()
    ops.expand_row_bits(csr, bits[0], b1, lst, ln)
    ops.spmm_rowlist_raw(csr, G, yl, lst, ln, alpha=0.25, z=G, beta=0.25, src_bits=bits[0], z_bits=bits[0])


print(f"clear + expand + row list   {timed(both):9.3f} ms")
d2 = ops.spmm_raw(csr, yd, z=G, beta=0.25)
print(f"dense  y = A y1 + G         {timed(lambda: ops.spmm_raw(csr, yd, y=d2, z=G, beta=0.25)):9.3f} ms")
print(f"sparse #2 row mask + zeros  {timed(lambda: ops.spmm_rowsparse_raw(csr, y1, y2, z=G, beta=0.25, src_bits=bits[1], z_bits=bits[0], row_bits=bits[2], write_zeros=True)):9.3f} ms")
print("  equal to dense:", bool(torch.equal(y2, d2)))
print(f"sparse #2 entry bits only   {timed(lambda: ops.spmm_rowsparse_raw(csr, yl, y2, z=G, beta=0.25, src_bits=bits[1], z_bits=bits[0])):9.3f} ms   (source = the row-list launch's output: unlisted rows hold stale values)")
print("  equal to dense:", bool(torch.equal(y2, d2)))
