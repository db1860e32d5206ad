"""Could the bf16 sweep SKIP item tiles?  An item j can only be a candidate of user u (upper bound of its score above the
threshold T_u) if  ||u|| ||i_j|| (1 + c) > T_u  (Cauchy-Schwarz), i.e. if ||i_j|| > r_u = T_u / ||u||.  With the items
walked in order of descending norm, a user block can stop at the first tile whose largest norm is below the block's
smallest r_u.  This measures, on trained LightGCN/sports tables, which share of the items survives that cut per user and
per group of 32 / 96 / 384 users (sweep block / wave / workgroup), for T_u = the exact score of rank 110.

    python3 tools/norm_cutoff.py [train_steps ...]
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
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    st = FusedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-3), batch_size=1024,
                           edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42,
                           step_dev=torch.zeros(1, dtype=torch.int64, device=dev), steps_per_replay=10)
    done, out = 0, {}
    for target in [int(a) for a in sys.argv[1:]] or [300, 1500, 5000, 20000]:
        st.run(target - done)
        done = target
        res = m.result.detach()
        ue, ie = res[:U], res[U:]
        s = ue @ ie.t()
        T = torch.topk(s, 110, dim=1).values[:, -1]
        un, inorm = ue.norm(dim=1), ie.norm(dim=1)
        r = torch.where(T > 0, T / un, torch.zeros_like(T))          # T <= 0: no item can be excluded by its norm
        srt = torch.sort(inorm).values
        rec = {"positive_threshold_share": float((T > 0).float().mean()),
               "item_norm_max_over_median": float(inorm.max() / inorm.median())}
        for grp in (1, 32, 96, 384):
            n = U // grp * grp
            rg = r[:n].view(-1, grp).min(1).values                   # the group's weakest user decides
            survive = (I - torch.searchsorted(srt, rg / 1.01)).float() / I
            rec[f"group{grp}"] = dict(mean=float(survive.mean()), median=float(survive.median()), p90=float(survive.quantile(0.9)))
        out[f"steps_{target}"] = rec
        print(target, json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "norm_cutoff_sports.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
