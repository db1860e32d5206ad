"""The skinny GEMMs of a 64-wide Linear over all graph nodes (MMGCN layers 2-4): both pipes, every direction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (44147, 60499):
    for K, N in ((64, 64), (128, 64), (64, 128)):
        x, w, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
        b = torch.randn(N, device=dev)
        print(f"M={M} in={K} out={N}:  x W^T+b act: f32 {timed(lambda: ops.gemm_raw(x, w, transB=True, bias=b, act=1)):6.1f} bf16x3 {timed(lambda: ops.gemm_nt_bf16x3(x, w, bias=b, act=1)):6.1f} |"
              f"  g W: f32 {timed(lambda: ops.gemm_raw(g, w)):6.1f} bf16x3 {timed(lambda: ops.gemm_nn_bf16x3(g, w)):6.1f} |"
              f"  g^T x: f32 {timed(lambda: ops.gemm_raw(g, x, transA=True)):6.1f} bf16x3 {timed(lambda: ops.gemm_tn_bf16x3(g, x)):6.1f}"
              f" swapped {timed(lambda: ops.gemm_tn_bf16x3(x, g)):6.1f} us")
