#!/usr/bin/env python3
"""The light step's gated backward launch (every row computed, gathers gated by N1's bitmap) alone, on the synthetic config-5
graphs, beside the dense launch -- and, with KNOCKOUT=1, with every / no / a random half of the sources flagged: where its
time is (profiles/r06_exp_entry_flags.txt).  Measurement only.    python tools/gated_bench.py config5_shard|config5 [D]"""
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
nnz = csr.col.numel()
deg = (csr.rowptr[1:] - csr.rowptr[:-1])
flagged = torch.zeros(N, dtype=torch.bool, device=dev)
flagged[list1[:int(n1)].long()] = True
gathers = int(flagged[csr.col.long()].sum())
print(f"{dataset} D={D}: {N} rows, {nnz} entries, N1 = {int(n1)} rows, {gathers} flagged entries")
y0, y1 = (torch.empty(N, D, device=dev) for _ in range(2))



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


by = gathers * (4.0 * D) + nnz * 8.0 + N * (4.0 * D + 8)
t = timed(lambda: ops.spmm_raw(csr, x, y=y0, z=z, beta=0.5))
print(f"   dense launch                  {t:8.3f} ms   {(nnz * (4.0 * D + 8) + N * (4.0 * D + 8)) / t / 1e6 / 8000:.3f} of 8 TB/s")
t = timed(lambda: ops.spmm_rowsparse_raw(csr, x, y1, z=z, beta=0.5, src_bits=bits1, z_bits=bits0))
print(f"   gated, bitmap probes          {t:8.3f} ms   {by / t / 1e6 / 8000:.3f}")
if os.environ.get("ADAM"):
    p_, m_, v_ = (torch.zeros(N, D, device=dev) for _ in range(3))
    bc = torch.tensor([0.1, 0.001], device=dev)
    t = timed(lambda: ops.spmm_adam_raw(csr, x, p_, m_, v_, bc, 1e-3, (0.9, 0.999), 1e-8, 0.0, alpha=1.0, z=z, beta=0.5))
    print(f"   dense launch + Adam epilogue  {t:8.3f} ms   {(nnz * (4.0 * D + 8) + N * (4.0 * D + 8) + N * 4.0 * D * 7) / t / 1e6 / 8000:.3f}")
if os.environ.get("KNOCKOUT"):
    ones = torch.full_like(bits1, -1)
    zeros = torch.zeros_like(bits1)
    for name, b in (("all sources flagged", ones), ("no source flagged", zeros)):
        t = timed(lambda: ops.spmm_rowsparse_raw(csr, x, y1, z=z, beta=0.5, src_bits=b, z_bits=bits0))
        print(f"   {name:22s}: {t:8.3f} ms")
    # every other entry's source: random half
    half = torch.randint(-2 ** 31, 2 ** 31 - 1, bits1.shape, device=dev, dtype=torch.int64).to(torch.int32)
    t = timed(lambda: ops.spmm_rowsparse_raw(csr, x, y1, z=z, beta=0.5, src_bits=half, z_bits=bits0))
    print(f"   random half of the rows flagged: {t:8.3f} ms")
