"""Could a per-call threshold from a DENSE exact pass over a few hundred "hot" items replace the carried / sampled
thresholds of the prefilter?  For user u the K-th best non-history score among any item subset is a valid lower bound
of the true K-th best.  This measures how tight it is on trained LightGCN/sports tables: hot = the H items of largest
embedding norm (or of largest degree); T_u = K-th best over the hot items not in u's history; candidates = items whose
bf16 upper bound (score + 1.05 * 2^-8 * |u||i|) exceeds T_u.

    python3 tools/hot_threshold.py [train_steps ...]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import load_graph
    from chaorec_amd import _lib
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    _lib.ensure_built()
    dev = torch.device("cuda:0")
    edges, U, I, _ = load_graph("sports")
    K = 50
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    st = FusedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-3), batch_size=1024,
                           edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42,
                           step_dev=torch.zeros(1, dtype=torch.int64, device=dev), steps_per_replay=10)
    hr, hc = m.hist
    hist_mask = torch.zeros((U, I), dtype=torch.bool, device=dev)
    rows = torch.repeat_interleave(torch.arange(U, device=dev), hr[1:] - hr[:-1])
    hist_mask[rows, hc.long()] = True
    ideg = torch.bincount(torch.from_numpy(edges[:, 1].astype(np.int64) - U).to(dev), minlength=I)
    done, out = 0, {}
    for target in [int(a) for a in sys.argv[1:]] or [10, 300, 1500, 5000, 20000]:
        st.run(target - done)
        done = target
        res = m.result.detach()
        ue, ie = res[:U], res[U:]
        s = ue @ ie.t()
        ub = s + (1.05 / 256.0) * ue.norm(dim=1, keepdim=True) * ie.norm(dim=1)[None, :]
        s_m = s.masked_fill(hist_mask, -1e30)
        true_k = torch.topk(s_m, K, dim=1).values[:, -1]
        rec = {}
        for kind, order in (("norm", torch.argsort(ie.norm(dim=1), descending=True)), ("degree", torch.argsort(ideg, descending=True))):
            for H in (128, 256, 512, 1024):
                hot = order[:H]
                T = torch.topk(s_m[:, hot], K, dim=1).values[:, -1]
                cand = (ub > T[:, None]).sum(1).float()
                exact_above = (s_m > T[:, None]).sum(1).float()
                rec[f"{kind}_H{H}"] = dict(cand_mean=float(cand.mean()), cand_median=float(cand.median()), cand_p99=float(cand.quantile(0.99)),
                                           over512=float((cand > 512).float().mean()), exact_above_mean=float(exact_above.mean()),
                                           gap_rel=float(((true_k - T) / true_k.abs().clamp_min(1e-12)).mean()))
        # for comparison: threshold = exact score of rank 110 (what a carried threshold aims at)
        t110 = torch.topk(s_m, 110, dim=1).values[:, -1]
        rec["rank110"] = dict(cand_mean=float((ub > t110[:, None]).sum(1).float().mean()))
        out[f"steps_{target}"] = rec
        print(target, json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "hot_threshold_sports.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
