"""Experiment: does the ranking call get faster when the users are split into chunks whose sweep (MFMA-bound) and
selection (VALU / L2-gather-bound) overlap on two HIP streams?  Steady-state call of tools/score_profile.py
(EPOCH_APART: thresholds one epoch old), the whole user range against P chunks alternating between two streams."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import _lib, dataload, ops, ranking  # noqa: E402

_lib.ensure_built()          # (an experiment build -- CHAOREC_EXTRA_HIPCC_FLAGS -- compiles its own library on first use)
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402

dev = torch.device("cuda:0")
d = dataload.packed_interactions("sports")
U, I, edges = d["num_user"], d["num_item"], d["train"]
torch.manual_seed(42)
m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
step = FusedLightGCNStep(m, opt, batch_size=1024, edges=edges_dev, seed=42, step_dev=cnt, steps_per_replay=5)
step.run(3000)
res = m.result.detach().clone()
rank = ranking.hint_rank_for(50)
old = torch.empty(U, dtype=torch.float32, device=dev)
ops.score_topk(res[:U], res[U:U + I], m.hist, 1e-6, 50, id_offset=U, hint=old, hint_valid=False, hint_rank=rank)
step.run(155)
res = m.result.detach().clone()
ue, ie = res[:U], res[U:U + I]
rowptr, col = m.hist
hint = torch.empty(U, dtype=torch.float32, device=dev)
ref_idx = None


def wall(fn, n=12):
    ts = []
    for _ in range(n):
        hint.copy_(old)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts[2:]))


def whole():
    global ref_idx
    ref_idx, _ = ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=rank, light=True)


print(f"whole range, one stream: {wall(whole):7.1f} us")
side = torch.cuda.Stream()
for P in (2, 3, 4, 6):
    cuts = [U * k // P // 64 * 64 for k in range(P)] + [U]
    outs = [None] * P

    def chunks(two_streams):
        main = torch.cuda.current_stream()
        if two_streams:
            side.wait_stream(main)
        for k in range(P):
            a, b = cuts[k], cuts[k + 1]
            s = side if (two_streams and k % 2) else main
            with torch.cuda.stream(s):
                outs[k] = ops.score_topk(ue[a:b], ie, (rowptr[a:b + 1], col), 1e-6, 50, id_offset=U, hint=hint[a:b],
                                         hint_valid=True, hint_rank=rank, light=True)[0]
        if two_streams:
            main.wait_stream(side)

    t1, t2 = wall(lambda: chunks(False)), wall(lambda: chunks(True))
    same = torch.equal(torch.cat(outs, 0), ref_idx)
    print(f"{P} chunks: one stream {t1:7.1f} us   two streams {t2:7.1f} us   same ranking: {same}")
