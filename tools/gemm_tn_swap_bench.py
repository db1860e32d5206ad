import sys, torch
sys.path.insert(0, "/root/repo")
from chaorec_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
K = 60499
for N in (64, 128, 256, 320, 768, 832):
    gy, x = torch.randn(K, 64, device=dev), torch.randn(K, N, device=dev)
    t1 = timed(lambda: ops.gemm_raw(gy, x, transA=True))
    t2 = timed(lambda: ops.gemm_tn_bf16x3(gy, x))
    t3 = timed(lambda: ops.gemm_tn_bf16x3(x, gy).t().contiguous())
    ok = torch.allclose(ops.gemm_tn_bf16x3(x, gy).t(), ops.gemm_raw(gy, x, transA=True), rtol=1e-4, atol=1e-2)
    print(f"dW[64,{N}] over {K} rows: f32 {t1:7.1f}  bf16x3 TN {t2:7.1f}  swapped TN + transpose {t3:7.1f} us  close={ok}")
