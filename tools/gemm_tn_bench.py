import os, sys, torch
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
for K, M, N in ((60499, 768, 768), (60499, 64, 64), (60499, 64, 128), (60499, 256, 256), (60499, 64, 832), (60499, 64, 768), (14079, 256, 128), (60499, 64, 320)):
    gy, x = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    t1 = timed(lambda: ops.gemm_raw(gy, x, transA=True)); t2 = timed(lambda: ops.gemm_tn_bf16x3(gy, x))
    print(f"gy^T x  K={K} M={M} N={N}: f32 {t1:8.1f} us   bf16x3 TN {t2:8.1f} us   ({2*K*M*N/t2/1e6:6.1f} TF/s-equivalent)")
print("NT (forward / input gradient):")
for M, N, K in ((60499, 768, 768), (60499, 256, 256), (60499, 128, 64), (60499, 64, 128), (60499, 832, 64), (60499, 64, 832), (14079, 256, 128), (11384, 64, 4096)):
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    t1 = timed(lambda: ops.gemm_raw(x, w, transB=True)); t2 = timed(lambda: ops.gemm_nt_bf16x3(x, w))
    print(f"x W^T   M={M} N={N} K={K}: f32 {t1:8.1f} us   bf16x3 NT {t2:8.1f} us   ({2*K*M*N/t2/1e6:6.1f} TF/s-equivalent)")
