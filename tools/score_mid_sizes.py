"""Own-order against norm-sorted packed tables at MID item counts (17 k - 130 k: own order by default), log-normal item norms:
cold and carried-threshold call times.  python3 tools/score_mid_sizes.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
K = 50


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for D in (64, 128):
    for U, I in ((65536, 20000), (65536, 40000), (65536, 80000), (28940, 120000)):
        g = torch.Generator(device=dev).manual_seed(3)
        ue = torch.randn(U, D, generator=g, device=dev) * 0.1
        ie = torch.randn(I, D, generator=g, device=dev) * 0.1 * torch.exp2(torch.randn(I, 1, generator=g, device=dev) * 0.6)
        rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 8
        col = (torch.arange(U * 8, device=dev) % 8 * (I // 8) + torch.arange(U * 8, device=dev) // 8 % (I // 8)).to(torch.int32)
        hist = (rowptr, col)
        row = []
        for name, v in (("own", 0), ("sorted", 1)):
            os.environ["CHAOREC_PF_CLS_MIN_ITEMS"] = str(v)
            hint = torch.empty(U, device=dev)
            st = {}
            ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=False, stats=st)
            cold = t(lambda: ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U))
            hot = t(lambda: ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=True))
            row.append(f"{name}: cold {cold:7.3f} ms  carried {hot:7.3f} ms  cand/user {st['candidates'] / U:6.1f} fb {st['fallback_users']}")
        print(f"D={D} U={U} I={I}:  " + "   |   ".join(row), flush=True)
