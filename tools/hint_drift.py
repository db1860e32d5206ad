"""How good is "last evaluation's K'-th best score" as this evaluation's candidate threshold?  Trains LightGCN on the
real sports graph with the fused step, one evaluation per epoch (155 steps), and reports per epoch, for several K':
mean / p99 number of items above the carried threshold and the share of users with fewer than 50 above it (a failed
hint: those users fall back to the sampled threshold).  Exact fp32 scores via torch (chunks of users)."""
import sys
import os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload  # noqa: E402
from chaorec_amd.Model import LightGCN  # noqa: E402
from chaorec_amd.optim import FusedAdam, FusedLightGCNStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    d = dataload.packed_interactions("sports")
    U, I, edges = d["num_user"], d["num_item"], d["train"]
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    step = FusedLightGCNStep(m, opt, batch_size=1024, edges=edges_dev, seed=42, step_dev=cnt, steps_per_replay=5)
    rp, col = m.hist
    rows = torch.repeat_interleave(torch.arange(U, device=dev), rp[1:] - rp[:-1])
    KS = [50, 56, 64, 80, 100, 128]
    prev = None
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    for ep in range(epochs):
        step.run(155)
        res = m.result.detach()
        sc = res[:U] @ res[U:].t()
        sc[rows, col.long()] = 1e-6
        top = torch.topk(sc, 256, dim=1).values                  # [U, 256] descending
        line = [f"ep {ep + 1:3d} |s_50| med {float(top[:, 49].abs().median()):.3e}"]
        if prev is not None:
            for k in KS:
                thr = prev[:, k - 1:k]
                n = (sc > thr).sum(1).float()
                fail = float((n < 50).float().mean())
                line.append(f"K'={k}: n {float(n.mean()):6.1f} p99 {float(n.quantile(0.99)):6.0f} fail {fail * 100:5.2f}%")
            # and a relative-margin variant: thr = s_50 - 0.15 (s_1 - s_50)
            for a in (0.1, 0.2, 0.3):
                thr = (prev[:, 49] - a * (prev[:, 0] - prev[:, 49])).unsqueeze(1)
                n = (sc > thr).sum(1).float()
                line.append(f"a={a}: n {float(n.mean()):6.1f} fail {float((n < 50).float().mean()) * 100:5.2f}%")
        prev = top
        if ep < 6 or ep % 4 == 3:
            print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
