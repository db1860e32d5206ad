#!/usr/bin/env python3
"""Time chaorec_gemm_f32 on the MMGCN / FREEDOM shapes (forward NT, input-gradient NN, weight-gradient TN)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import ops
dev = torch.device("cuda:0")
shapes = [  # (M, N, K, transA, transB)  -- C[M,N] = op(A) op(B)
    (60499, 256, 256, False, True), (60499, 64, 320, False, True), (60499, 64, 64, False, True),
    (60499, 256, 64, False, False), (256, 256, 60499, True, False), (64, 320, 60499, True, False),
    (11384, 64, 4096, False, True), (64, 4096, 11384, True, False), (14079, 256, 128, False, True),
    # NGCF at sports size: forward / input-gradient / weight-gradient of a 64 -> 64 layer over all 47 k nodes
    (47297, 64, 64, False, True), (47297, 64, 64, False, False), (64, 64, 47297, True, False), (47297, 64, 128, False, True),
]
for M, N, K, tA, tB in shapes:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    for _ in range(3):
        ops.gemm_raw(A, B, transA=tA, transB=tB)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm_raw(A, B, transA=tA, transB=tB)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"M={M:6d} N={N:5d} K={K:6d} tA={int(tA)} tB={int(tB)}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s")

print("-- split-bf16 NT (chaorec_gemm_nt_bf16x3) vs f32 MFMA on the forward shapes")
for M, N, K in [(11384, 64, 4096), (11384, 64, 384), (14079, 256, 128), (60499, 256, 256), (60499, 64, 320), (60499, 768, 768),
                (47297, 64, 64)]:
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K), device=dev)
    for name, fn in (("f32   ", lambda: ops.gemm_raw(A, B, transB=True)), ("bf16x3", lambda: ops.gemm_nt_bf16x3(A, B))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"{name} M={M:6d} N={N:5d} K={K:6d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  A-stream {M*K*4/ms/1e9:7.2f} TB/s")
