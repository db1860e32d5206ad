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
}


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
