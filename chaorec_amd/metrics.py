"""Per-user ranking metrics with the reference's names and semantics (metrics.py:13-57).

Kept as plain functions for API parity; the evaluation loop uses utils.gene_metrics (vectorised, host) or
utils.gene_metrics_device (one launch on the GPU), which compute the same numbers for all users at once.
All five are derived from one helper: the positions of ranked[:k] that hit the test list."""
import numpy as np


def _hit_positions(ranked_list, test_list, k):
    """0-based ranks p < k with ranked_list[p] in test_list (membership per position, duplicates included)."""
    wanted = set(test_list)
    return [p for p, item in enumerate(ranked_list[:k]) if item in wanted]


def _distinct_hits(ranked_list, test_list, k):
    """|set(ranked[:k]) & set(test)|: a repeated id in the ranking counts once."""
    return len({ranked_list[p] for p in _hit_positions(ranked_list, test_list, k)})


def precision_at_k(ranked_list, test_list, k):
    return _distinct_hits(ranked_list, test_list, k) / k


def recall_at_k(ranked_list, test_list, k):
    n = len(test_list)
    return _distinct_hits(ranked_list, test_list, k) / n if n else 0


def hit_rate_at_k(ranked_list, test_list, k):
    return 1 if _hit_positions(ranked_list, test_list, k) else 0


def ndcg_at_k(ranked_list, test_list, k):
    n = len(test_list)
    if n == 0:
        return 0
    gain = 1.0 / np.log(np.arange(k) + 2.0)                      # natural log, as the reference; the base cancels
    ideal = sum(float(gain[i]) for i in range(min(n, k)))        # left-to-right, like the reference's sum()
    got = sum(float(gain[p]) for p in _hit_positions(ranked_list, test_list, k))
    return got / ideal


def map_at_k(ranked_list, test_list, k):
    n = len(test_list)
    if n == 0:
        return 0
    total = 0
    for seen, p in enumerate(_hit_positions(ranked_list, test_list, k), start=1):
        total += seen / (p + 1)
    return total / n
