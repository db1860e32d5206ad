"""gene_ranklist on a trained LightGCN/sports model, three ways: cold (sampled thresholds), carried thresholds on the SAME
tables, carried thresholds from one epoch (155 steps) earlier.  Prints event timings; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload, ops  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


def main():
    dev = torch.device("cuda:0")
    train_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    d = dataload.packed_interactions("sports")
    U, I, edges = d["num_user"], d["num_item"], d["train"]
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    step = FusedLightGCNStep(m, opt, batch_size=1024, edges=edges_dev, seed=42, step_dev=cnt, steps_per_replay=5)
    step.run(train_steps)
    res = m.result.detach().clone()
    ue, ie = res[:U], res[U:U + I]
    hint = torch.empty(U, dtype=torch.float32, device=dev)
    st = {}
    if os.environ.get("ONLY_CARRIED"):
        ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False)
        for _ in range(8):
            ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, light=True)
        torch.cuda.synchronize()
        return
    if os.environ.get("EPOCH_APART"):
        # bench.py's steady state: thresholds of rank 100 from the tables one epoch (155 steps) earlier, light mode
        old = torch.empty(U, dtype=torch.float32, device=dev)
        ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=old, hint_valid=False, hint_rank=int(os.environ.get("HINT_RANK0", "100")))
        step.run(155)
        res = m.result.detach().clone()
        ue, ie = res[:U], res[U:U + I]
        counters = torch.zeros(4, dtype=torch.int32, device=dev)

        def steady():
            hint.copy_(old)
            ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True,
                           hint_rank=int(os.environ.get("TIMED_HINT_RANK", "100")), light=True, counters=counters)
        t = timed(steady, 8)
        ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=100, light=True, stats=st)
        print(f"epoch apart, light        {t * 1e3:7.1f} us  cand/user {st['candidates'] / U:6.1f}  queues {counters.tolist()}")
        return
    cold = timed(lambda: ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U))
    ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, stats=st)
    print(f"cold                      {cold * 1e3:7.1f} us  cand/user {st['candidates'] / U:6.1f}  exact-route users {st['fallback_users']}")
    ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False)
    same = timed(lambda: ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True))
    ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, stats=st)
    print(f"carried, same tables      {same * 1e3:7.1f} us  cand/user {st['candidates'] / U:6.1f}  exact-route users {st['fallback_users']}")
    # one epoch apart: hints from the tables before 155 more steps
    for rank in [int(x) for x in os.environ.get("HINT_RANKS", "80").split(",")]:
        ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False, hint_rank=rank)
        counters = torch.zeros(4, dtype=torch.int32, device=dev)
        for ep in range(int(os.environ.get('EPOCHS', '4'))):
            step.run(155)
            res = m.result.detach()
            ue, ie = res[:U], res[U:U + I]
            light = ep > 0 and int(counters[0]) <= 256
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=rank, light=light,
                           counters=counters)
            e.record()
            torch.cuda.synchronize()
            st2 = {}
            keep = hint.clone()
            ops.score_topk(ue, ie, m.hist, 1e-6, 50, id_offset=U, hint=keep, hint_valid=True, hint_rank=rank, light=light, stats=st2)
            print(f"carried (rank {rank}), one epoch apart, light={int(light)}  {s.elapsed_time(e) * 1e3:7.1f} us  "
                  f"queues [retry, exact, wide, retry->exact] {counters.tolist()}  (same tables again: cand/user {st2['candidates'] / U:6.1f})")


if __name__ == "__main__":
    main()
