#!/usr/bin/env python3
"""The wide projections of MMGCN's first layer (Model/MMGCN.py:97-131 on microlens: N = 115 k rows, K = 768 / 128 features,
64 / 128 outputs) through the product's split-bf16 GEMMs, beside torch's fp32 matmul of the same shapes (hipBLASLt): what
the step pays for them and what the library gets out of the same hardware.  Prints a table; measurement only."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import _lib, ops  # noqa: E402

_lib.ensure_built()
_lib.load()
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[reps // 2] * 1e3


# (60499, 772, 768): MMGCN's textual first convolution on microlens -- [A x | A 1] [W | b]^T and its weight gradient, the step's two
# largest launches (round 6: 563 / 732 us in the one-stream kernel trace before the slab-count and tile-order changes)
SHAPES = ((60499, 772, 768), (115000, 768, 64), (115000, 768, 128), (115000, 128, 128), (115000, 64, 128))
if len(sys.argv) > 1 and sys.argv[1] == "mmgcn":
    SHAPES = SHAPES[:1]
for N, K, M in SHAPES:
    x = torch.randn(N, K, device=dev)
    w = torch.randn(M, K, device=dev) * 0.05
    gy = torch.randn(N, M, device=dev)
    fl = 2.0 * N * K * M
    rows = []
    y = torch.empty(N, M, device=dev)
    rows.append(("NT  y = x W^T        product", timed(lambda: ops.gemm_nt_bf16x3(x, w, out=y) if hasattr(ops, "gemm_nt_bf16x3") else ops.linear(x, w))))
    rows.append(("NT  y = x W^T        torch  ", timed(lambda: torch.mm(x, w.t(), out=y))))
    gw = torch.empty(M, K, device=dev)
    if hasattr(ops, "gemm_tn_bf16x3"):
        rows.append(("TN  gW = gy^T x      product", timed(lambda: ops.gemm_tn_bf16x3(gy, x, out=gw))))
    rows.append(("TN  gW = gy^T x      torch  ", timed(lambda: torch.mm(gy.t(), x, out=gw))))
    gx = torch.empty(N, K, device=dev)
    if hasattr(ops, "gemm_nn_bf16x3"):
        rows.append(("NN  gx = gy W        product", timed(lambda: ops.gemm_nn_bf16x3(gy, w, out=gx))))
    rows.append(("NN  gx = gy W        torch  ", timed(lambda: torch.mm(gy, w, out=gx))))
    print(f"N={N} K={K} M={M}  ({fl / 1e9:.1f} GFLOP; x {N * K * 4 / 1e6:.0f} MB)")
    for name, us in rows:
        print(f"   {name}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
