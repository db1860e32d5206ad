"""How large is the UNION of the candidate sets of a block of 32 consecutive users?  (Sizing of the block-joint exact
re-score of the ranking: one f32-MFMA pass over the union instead of one gathered chain per (user, candidate).)

    python3 tools/union_size.py [dataset=sports]

Trains LightGCN with the fused step, and at a few training states takes every user's top-R items (R = the carried
threshold rank 110, 150) from a dense torch matmul and counts distinct items per block of 32 / 64 users."""
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
    dataset = sys.argv[1] if len(sys.argv) > 1 else "sports"
    dev = torch.device("cuda:0")
    edges, U, I, _ = load_graph(dataset)
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    st = FusedLightGCNStep(m, opt, batch_size=1024, edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42,
                           step_dev=torch.zeros(1, dtype=torch.int64, device=dev), steps_per_replay=10)
    done = 0
    out = {}
    for target in (10, 300, 1500, 5000):
        st.run(target - done)
        done = target
        res = m.result.detach()
        s = res[:U] @ res[U:].t()
        rec = {}
        for R in (110, 150):
            top = torch.topk(s, R, dim=1).indices                     # [U, R]
            for blk in (32, 64):
                nb = U // blk
                t = top[:nb * blk].reshape(nb, blk * R).sort(1).values
                distinct = 1 + (t[:, 1:] != t[:, :-1]).sum(1)
                d = distinct.float()
                rec[f"R{R}_block{blk}"] = dict(mean=float(d.mean()), median=float(d.median()), p99=float(d.quantile(0.99)),
                                                max=int(d.max()), tiles32_mean=float(torch.ceil(d / 32).mean()))
        out[f"steps_{target}"] = rec
        print(target, json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"union_size_{dataset}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
