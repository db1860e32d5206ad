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
for rep in range(int(os.environ.get("REPS", "2"))):
    ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=False, stats=st)
    ops.score_topk(ue, ie, (rowptr, col), 1e-6, 50, id_offset=U, hint=hint, hint_valid=True)
torch.cuda.synchronize()
print(st)
