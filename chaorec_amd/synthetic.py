"""Synthetic interaction graphs with the shape of the reference's datasets (bench / full-size tests).

The GPU box has no copy of the reference's Data/ directory, so bench.py and the full-size
property tests run on seeded synthetic graphs of the same U, I, E and the same degree shape
(user degree min 3 / median 3 with a heavy tail, item degree power-law, no duplicate
edges, file sorted by user, item ids global = item + U).
"""
import numpy as np

DATASET_SHAPES = {  # dataload.py:36-56 + measured E of train.npy
    "baby": (12351, 4794, 64330),
    "sports": (28940, 15207, 158554),
    "clothing": (18072, 11384, 76054),
    "microlens": (46420, 14079, 210567),
    # one GPU's share of BASELINE configs[4] (10 M users x 2 M items, ~200 M edges over 8 GPUs): the HBM-bound regime
    "config5_shard": (1_250_000, 2_000_000, 25_000_000),
    # BASELINE configs[4] whole, on ONE GPU (the N = 1 anchor of its scaling curve): ~55 GB of tables, Adam state and CSR
    "config5": (10_000_000, 2_000_000, 200_000_000),
}
DEVICE_BUILT = {"config5"}     # generated and laid out with torch ops on the GPU (synthetic_interactions_torch)


def synthetic_interactions(num_user, num_item, num_edges, seed=42, min_deg=3):
    """-> int32 [E, 2] of (user, global item id), sorted by user, unique edges, every user >= min_deg."""
    rng = np.random.default_rng(seed)
    extra = num_edges - min_deg * num_user
    if extra < 0:
        raise ValueError("num_edges < min_deg * num_user")
    w = rng.pareto(2.2, num_user) + 0.05   # calibrated on sports: p90 ~10, p99 ~25, max ~200
    deg = min_deg + rng.multinomial(extra, w / w.sum())
    cap = max(min_deg, min(num_item // 4, 256))
    over = int((deg - np.minimum(deg, cap)).sum())
    deg = np.minimum(deg, cap)
    while over > 0:  # hand the clipped mass to random users
        take = rng.integers(0, num_user, over)
        np.add.at(deg, take, 1)
        over = int((deg - np.minimum(deg, cap)).sum())
        deg = np.minimum(deg, cap)
    pop = (np.arange(1, num_item + 1, dtype=np.float64) + 12.0) ** -0.75   # item p50 ~6, p99 ~75, max ~600
    pop = pop[rng.permutation(num_item)]
    pop /= pop.sum()
    cdf = np.cumsum(pop)
    users = np.repeat(np.arange(num_user, dtype=np.int64), deg)
    items = np.searchsorted(cdf, rng.random(users.size)).clip(0, num_item - 1).astype(np.int64)
    key = np.unique(users * num_item + items)
    for _ in range(64):  # top up what deduplication removed
        have = np.bincount(key // num_item, minlength=num_user)
        miss = deg - have
        if not miss.any():
            break
        mu = np.repeat(np.arange(num_user, dtype=np.int64), np.maximum(miss, 0))
        # later rounds draw uniformly so heavy users cannot stall on the popular head
        mi = rng.integers(0, num_item, mu.size)
        key = np.unique(np.concatenate([key, mu * num_item + mi]))
        # drop overshoot (a user can only overshoot if a top-up duplicated nothing): trim per user
        have = np.bincount(key // num_item, minlength=num_user)
        if (have > deg).any():
            start = np.zeros(num_user + 1, np.int64)
            np.cumsum(have, out=start[1:])
            keep = np.ones(key.size, bool)
            for u in np.nonzero(have > deg)[0]:
                keep[start[u] + deg[u]:start[u + 1]] = False
            key = key[keep]
    u = key // num_item
    i = key % num_item
    # shuffle item order inside each user like the real files (not sorted by item)
    perm = np.lexsort((rng.random(key.size), u))
    return np.stack([u[perm], i[perm] + num_user], 1).astype(np.int32)


def synthetic_eval_lists(num_user, num_item, train_edges, per_user=1, seed=7):
    """val/test style lists [user, pos...] that never overlap train (SURVEY 8 data notes)."""
    rng = np.random.default_rng(seed)
    seen = set((int(u) << 32) | int(i) for u, i in np.asarray(train_edges, dtype=np.int64))
    out = []
    for u in range(num_user):
        row = [u]
        while len(row) < 1 + per_user:
            c = int(rng.integers(num_user, num_user + num_item))
            if ((u << 32) | c) not in seen and c not in row[1:]:
                row.append(c)
        out.append(row)
    return out


def synthetic_interactions_torch(num_user, num_item, num_edges, seed=42, min_deg=3, device="cpu", chunk_users=1_000_000):
    """The same graph family as synthetic_interactions() for sizes where its one-shot numpy passes (a 2e8-key np.unique,
    a 2e8-row lexsort) take minutes and tens of GB: generated in USER CHUNKS with torch ops on `device` (the GPU for
    BASELINE configs[4]).  Deduplication is per user, so chunking by users is exact; every chunk tops its users up to their
    drawn degree.  -> int32 [E', 2] tensor on `device`, (user, GLOBAL item id), sorted by user, unique edges, every user >=
    min_deg; E' = num_edges up to the Poisson noise of the degree draw (~1e-4 relative at 2e8).
    Degrees: min_deg + Poisson(extra * w / sum w), w ~ Pareto(2.2) + 0.05, capped at min(num_item / 4, 256) -- the same
    law as the numpy generator's multinomial, without torch.multinomial's 2^24-category limit.  Item popularity:
    (rank + 12)^-0.75 over a seeded permutation.  A different device gives a different (equally distributed) graph."""
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    extra = num_edges - min_deg * num_user
    if extra < 0:
        raise ValueError("num_edges < min_deg * num_user")
    u01 = torch.rand(num_user, generator=g, device=dev, dtype=torch.float64)
    w = (1.0 - u01).clamp_min(1e-12).pow(-1.0 / 2.2) - 1.0 + 0.05
    lam = (w * (extra / float(w.sum()))).to(torch.float32)
    cap = max(min_deg, min(num_item // 4, 256))
    deg = (min_deg + torch.poisson(lam, generator=g).to(torch.int64)).clamp_max(cap)
    for _ in range(4):               # hand the mass the cap clipped to the users below it (the numpy generator does the same)
        short = num_edges - int(deg.sum())
        if short <= 0.0002 * num_edges:
            break
        wf = torch.where(deg < cap, w, torch.zeros_like(w))
        deg = (deg + torch.poisson((wf * (short / float(wf.sum()))).to(torch.float32), generator=g).to(torch.int64)).clamp_max(cap)
    del u01, w, lam
    pop = (torch.arange(1, num_item + 1, device=dev, dtype=torch.float64) + 12.0).pow(-0.75)
    pop = pop[torch.randperm(num_item, generator=g, device=dev)]
    cdf = torch.cumsum(pop / pop.sum(), 0)
    del pop
    parts = []
    for u0 in range(0, num_user, chunk_users):
        u1 = min(num_user, u0 + chunk_users)
        d = deg[u0:u1]
        users = torch.repeat_interleave(torch.arange(u0, u1, device=dev, dtype=torch.int64), d)
        items = torch.searchsorted(cdf, torch.rand(users.numel(), generator=g, device=dev, dtype=torch.float64)).clamp_(0, num_item - 1)
        key = torch.unique(users * num_item + items)
        del users, items
        for _ in range(64):          # top up what deduplication removed, uniformly (heavy users cannot stall on the head)
            have = torch.bincount(torch.div(key, num_item, rounding_mode="floor") - u0, minlength=u1 - u0)
            miss = (d - have).clamp_min(0)
            if int(miss.sum()) == 0:
                break
            mu = torch.repeat_interleave(torch.arange(u0, u1, device=dev, dtype=torch.int64), miss)
            mi = torch.randint(0, num_item, (mu.numel(),), generator=g, device=dev, dtype=torch.int64)
            key = torch.unique(torch.cat([key, mu * num_item + mi]))
            have = torch.bincount(torch.div(key, num_item, rounding_mode="floor") - u0, minlength=u1 - u0)
            if bool((have > d).any()):          # a top-up can only overshoot by landing on fresh items twice: trim per user
                start = torch.cumsum(have, 0) - have
                pos = torch.arange(key.numel(), device=dev) - torch.repeat_interleave(start, have)
                key = key[pos < torch.repeat_interleave(d, have)]
        uu = torch.div(key, num_item, rounding_mode="floor")
        ii = key - uu * num_item
        # item order inside a user like the real files (not sorted by item): sort by (user, random)
        r = torch.rand(key.numel(), generator=g, device=dev, dtype=torch.float64)
        order = torch.argsort(uu.to(torch.float64) + r * 0.999, stable=True)
        parts.append(torch.stack([uu[order], ii[order] + num_user], 1).to(torch.int32))
        del key, uu, ii, r, order
    return torch.cat(parts, 0)
