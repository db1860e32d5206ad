"""The cpu_baseline leg: the oracle timed on the host cores (the ONLY bench code that touches oracle/)."""
import json
import os
import sys
import time

import numpy as np
import torch


def cpu_baseline(edges, U, I, D, L, B, reg, budget_s):
    """The reference CPU path restated in plain torch (oracle/torch_ref.py), timed on this box's host cores on a
    bounded number of steps.  Best of a small sweep over torch's intra-op thread count, capped at 64 (the box has far
    more cores than a 0.3 M-edge scatter can use: all of them is slower than a few); `cores` = the thread count of the
    best run, the one `value` is quoted from."""
    from oracle.torch_ref import TorchRefLightGCN
    from chaorec_amd.graph import user_item_dict_from_edges
    ncpu = os.cpu_count() or 8
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})   # (all 256 threads: 20 s per step, never the best)
    uid = user_item_dict_from_edges(edges)
    rng = np.random.default_rng(0)
    E = len(edges)
    old_threads = torch.get_num_threads()

    def run(threads, budget):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        m = TorchRefLightGCN(U, I, edges, uid, D, reg, L)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            b = rng.integers(0, E, B)
            u, p = torch.from_numpy(edges[b, 0].astype(np.int64)), torch.from_numpy(edges[b, 1].astype(np.int64))
            n = torch.from_numpy(rng.integers(U, U + I, B))
            opt.zero_grad()
            loss = m.loss(u, p, n)
            loss.backward()
            opt.step()

        step()  # warm-up
        t0 = time.perf_counter()
        n_steps = 0
        while n_steps < 3 or (time.perf_counter() - t0 < budget and n_steps < 200):
            step()
            n_steps += 1
        return (time.perf_counter() - t0) / n_steps, n_steps, m

    share = budget_s * 0.6 / len(cands)
    tried = {}
    best = None
    for t in cands:
        dt, n_steps, m = run(t, share)
        tried[t] = dt * 1e3
        if best is None or dt < best[0]:
            best = (dt, n_steps, t, m)
    dt, n_steps, threads, m = best
    torch.set_num_threads(threads)
    t1 = time.perf_counter()
    with torch.no_grad():
        m.gene_ranklist()
    t_rank = time.perf_counter() - t1
    torch.set_num_threads(old_threads)
    e_dir = 2 * E
    return {
        "value": 2 * L * e_dir / dt, "unit": "directed-edge messages/s", "cores": threads, "kind": "port",
        "sample": f"{n_steps} train steps of the same workload ({dt * 1e3:.1f} ms/step) + 1 gene_ranklist "
                  f"({t_rank:.2f} s) with oracle/torch_ref.py (reference op sequence in plain torch, CPU); best of "
                  f"torch threads {cands} on {ncpu} host cores",
        "ms_per_step": dt * 1e3, "users_scored_per_s": U / t_rank,
        "ms_per_step_by_threads": {str(k): round(v, 1) for k, v in tried.items()},
    }
