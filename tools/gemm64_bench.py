import os, sys, torch
sys.path.insert(0, "/root/repo")
from chaorec_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (44147, 60499):
    x, w, g = torch.randn(M, 64, device=dev), torch.randn(64, 64, device=dev), torch.randn(M, 64, device=dev)
    b = torch.randn(64, device=dev)
    print(f"M={M}: x W^T+b act {timed(lambda: ops.gemm_raw(x, w, transB=True, bias=b, act=1)):6.1f}  g W {timed(lambda: ops.gemm_raw(g, w)):6.1f}  g^T x {timed(lambda: ops.gemm_raw(g, x, transA=True)):6.1f}  bf16x3 NT {timed(lambda: ops.gemm_nt_bf16x3(x, w, bias=b, act=1)):6.1f} us")
