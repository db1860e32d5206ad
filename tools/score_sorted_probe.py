"""Own-order against norm-sorted packed tables (CHAOREC_PF_CLS_MIN_ITEMS) on tables with several norm laws: statistics of a
cold and a carried-threshold call, and whether the results agree.  python3 tools/score_sorted_probe.py [U I]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import ops  # noqa: E402

U, I = (int(x) for x in (sys.argv[1:3] + ["4096", "200000"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
K = 50
for D in (64, 128):
    for law in ("level", "lognormal", "outliers"):
        g = torch.Generator(device=dev).manual_seed(7 + D)
        ue = torch.randn(U, D, generator=g, device=dev) * 0.1
        ie = torch.randn(I, D, generator=g, device=dev) * 0.1
        if law == "lognormal":
            ie *= torch.exp2(torch.randn(I, 1, generator=g, device=dev) * 0.7)
        elif law == "outliers":
            ie[torch.randint(0, I, (300,), generator=g, device=dev)] *= 30.0
            ie[torch.randint(0, I, (5,), generator=g, device=dev)] *= 1e6
        rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 6
        col = ((torch.arange(U * 6, device=dev) % 6) * 30011 + torch.arange(U * 6, device=dev) // 6 * 7 % 30011).to(torch.int32)
        hist = (rowptr, col)
        res = {}
        for name, v in (("own", 0), ("sorted", 1)):
            os.environ["CHAOREC_PF_CLS_MIN_ITEMS"] = str(v)
            hint = torch.empty(U, device=dev)
            st, st2 = {}, {}
            i0, v0 = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=False, stats=st)
            i1, v1 = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=True, stats=st2)
            torch.cuda.synchronize()
            res[name] = (i0, v0, i1, v1)
            for tag, s in (("cold", st), ("hinted", st2)):
                print(f"D={D} {law:9s} {name:6s} {tag:6s} cand/user {s['candidates'] / U:7.1f} longest {s['longest_list']:4d} "
                      f"fallback {s['fallback_users']:5d} {s['fallback_reasons']} reth {s.get('rethreshold_users')}", flush=True)
        print("   equal:", all(torch.equal(a, b) for a, b in zip(res["own"], res["sorted"])), flush=True)
