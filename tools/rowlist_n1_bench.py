#!/usr/bin/env python3
"""The light step's two launches over N1's row list alone, on the synthetic config-5 graphs: the forward layer L-1 (every entry of a
listed row gathered) and the backward's first propagate (gated by R0's bitmap: a handful of flagged entries per row).
Measurement only.    python tools/rowlist_n1_bench.py config5_shard|config5 [D]"""
import os
import sys
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
z = torch.randn(N, D, device=dev, generator=g) * 0.1
idx = torch.randint(0, ed.shape[0], (B,), device=dev, generator=g)
ids = (ed[idx, 0].long(), ed[idx, 1].long() - U, torch.randint(0, I, (B,), device=dev, generator=g))
bits0, bits1 = ops.row_bitmap(N, dev), ops.row_bitmap(N, dev)
list0, n0 = torch.empty(3 * B, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
ops.batch_rows(ids, bits0, U, list0, n0)
list1, n1 = torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
ops.expand_row_bits(csr, bits0, bits1, list1, n1)
rows = list1[:int(n1)].long()
deg = csr.rowptr[rows + 1] - csr.rowptr[rows]
print(f"{dataset} D={D}: N1 = {int(n1)} rows, {int(deg.sum())} entries; rows > 256: {int((deg > 256).sum())} with {int(deg[deg > 256].sum())} entries")
long_rows = ops.long_row_buffers(csr)
y = torch.empty(N, D, device=dev)


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[reps // 2]


def expand():
    bits1.zero_()
    n1.zero_()
    ops.expand_row_bits(csr, bits0, bits1, list1, n1)


t = timed(expand)
print(f"   expand_row_bits alone (+ two clears)      {t:8.3f} ms")
t = timed(lambda: ops.spmm_rowlist_raw(csr, x, y, list1, n1, long_rows=long_rows))
print(f"   forward over N1's list (ungated)          {t:8.3f} ms")
t = timed(lambda: ops.spmm_rowlist_raw(csr, x, y, list1, n1, alpha=0.25, z=z, beta=0.25, src_bits=bits0, z_bits=bits0, long_rows=long_rows))
print(f"   backward 1 over N1's list (gated by R0)   {t:8.3f} ms")
