"""One cold + one carried-threshold ranking call on random tables of a given shape (the command of the scoring --pmc passes:
python3 tools/pmc_kernels.py score_ -- python3 tools/score_case.py 262144 262144 128)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import ops  # noqa: E402

U, I, D = (int(x) for x in (sys.argv[1:4] + ["262144", "262144", "128"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
a = (6.0 / (U + I + D)) ** 0.5
ue = (torch.rand(U, D, generator=g, device=dev) * 2 - 1) * a
ie = (torch.rand(I, D, generator=g, device=dev) * 2 - 1) * a
rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 8
col = (torch.arange(U * 8, device=dev) % 8 * (I // 8) + torch.arange(U * 8, device=dev) // 8 % (I // 8)).to(torch.int32)
hint = torch.empty(U, device=dev)
st = {}
if "EXP_CUT" not in os.environ.get("CHAOREC_EXTRA_HIPCC_FLAGS", ""):
    ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=False)
else:      # (a stage-cut build cannot rank: thresholds that look like the real ones)
    hint.copy_((ue.norm(dim=1) * ie.norm(dim=1).mean()) * 0.31)
for rep in range(0 if "EXP_CUT" in os.environ.get("CHAOREC_EXTRA_HIPCC_FLAGS", "") else int(os.environ.get("REPS", "2"))):
    ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=False, stats=st)
    ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=True)
torch.cuda.synchronize()
print(st)
if os.environ.get("TIMES"):
    out = []
    for valid in (False, True):
        ts = []
        for _ in range(0 if "EXP_CUT" in os.environ.get("CHAOREC_EXTRA_HIPCC_FLAGS", "") else 3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=valid)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        out.append(min(ts) if ts else float("nan"))
    # the hinted call's FRONT phase alone = pack + the sweep over all users (stage-cut builds break the back phase)
    from chaorec_amd import _lib as L
    lib = L.load()
    nb = lib.chaorec_score_topk_workspace_bytes(U, I, 50, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    idx = torch.empty((U, 50), dtype=torch.int64, device=dev)
    val = torch.empty((U, 50), device=dev)
    href = hint.clone()
    ts = []
    for _ in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops._score_call(lib, ue, ie, (rowptr, col), 1e-6, 50, U, 0, href, True, 80, False, None, idx, val, ws, nb, phase=ops.SCORE_FRONT)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    out.append(min(ts))
    fl = 2.0 * U * I * D
    from chaorec_amd import _lib as _l
    print("lib", os.path.basename(_l.current_lib_path()))
    print(f"times cold {out[0]:.3f} ms ({fl / out[0] / 1e9 / 2500:.3f})  hinted {out[1]:.3f} ms ({fl / out[1] / 1e9 / 2500:.3f})  pack+sweep {out[2]:.3f} ms ({fl / out[2] / 1e9 / 2500:.3f})")
