"""Does a ranking call depend on what its workspace held before?  The call in one piece and as FRONT + BACK phases, unhinted and
hinted, over workspaces pre-filled with 0x00 / 0xFF / 0x5A bytes: every variant must give the same [U, K] lists.
    python3 tools/score_phase_dirty_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(3)
for (U, I, D) in ((8192, 6000, 64), (4096, 20000, 128)):
    K = 50
    ue = torch.randn(U, D, generator=g, device=dev) * 0.2
    ie = torch.randn(I, D, generator=g, device=dev) * 0.2
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 3
    col = ((torch.arange(U * 3, device=dev) % 3) * 1000 + torch.arange(U * 3, device=dev) // 3 % 997).to(torch.int32)
    hist = (rowptr, col)
    nb = lib.chaorec_score_topk_workspace_bytes(U, I, K, D)
    ref_i, ref_v = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U)
    h0 = torch.empty(U, device=dev)
    ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=h0, hint_valid=False)
    bad = 0
    for fill in (0x00, 0xFF, 0x5A):
        for hinted in (False, True):
            for phased in (False, True):
                ws = torch.full((nb,), fill, dtype=torch.uint8, device=dev)
                idx = torch.empty((U, K), dtype=torch.int64, device=dev)
                val = torch.empty((U, K), dtype=torch.float32, device=dev)
                hint = h0.clone() if hinted else None
                cnt = torch.zeros(4, dtype=torch.int32, device=dev)
                a = (lib, ue, ie, hist, 1e-6, K, U, 0, hint, hinted, 0, False, cnt, idx, val, ws, nb)
                if phased:
                    ops._score_call(*a, phase=ops.SCORE_FRONT)
                    ops._score_call(*a, phase=ops.SCORE_BACK)
                else:
                    ops._score_call(*a)
                torch.cuda.synchronize()
                ok = torch.equal(idx, ref_i) and torch.equal(val, ref_v)
                bad += not ok
                print(f"U={U} I={I} D={D} fill={fill:#04x} hinted={hinted!s:5} phased={phased!s:5} -> {'same' if ok else 'DIFFERENT'}"
                      + ("" if ok else f" ({int((idx != ref_i).any(1).sum())} users)"), flush=True)
    print("mismatching variants:", bad)
