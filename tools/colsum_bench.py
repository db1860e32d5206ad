import sys, torch
sys.path.insert(0, "/root/repo")
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
for M, N in ((60499, 64), (60499, 768), (44147, 64), (2048, 64), (60499, 256)):
    x = torch.randn(M, N, device=dev)
    print(f"col_sum [{M},{N}]: {timed(lambda: ops.col_sum(x)):6.1f} us   err {float((ops.col_sum(x).double() - x.double().sum(0)).abs().max()):.2e}")
