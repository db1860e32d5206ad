"""Per-user ranking metrics with the reference's names and semantics (metrics.py:13-57).

Kept as plain functions for API parity; the evaluation loop uses the vectorised
utils.gene_metrics, which computes the same numbers for all users at once."""
import numpy as np


def precision_at_k(ranked_list, test_list, k):
    return len(set(ranked_list[:k]) & set(test_list)) / k


def recall_at_k(ranked_list, test_list, k):
    if len(test_list) == 0:
        return 0
    return len(set(ranked_list[:k]) & set(test_list)) / len(test_list)


def ndcg_at_k(ranked_list, test_list, k):
    if not test_list:
        return 0
    test = set(test_list)
    idcg = sum(1.0 / np.log(i + 2) for i in range(min(len(test_list), k)))
    dcg = sum(1.0 / np.log(i + 2) for i, item in enumerate(ranked_list[:k]) if item in test)
    return dcg / idcg


def hit_rate_at_k(ranked_list, test_list, k):
    return int(bool(set(ranked_list[:k]) & set(test_list)))


def map_at_k(ranked_list, test_list, k):
    if not test_list:
        return 0
    test = set(test_list)
    scores, num_hits = 0, 0
    for i, item in enumerate(ranked_list[:k]):
        if item in test:
            num_hits += 1
            scores += num_hits / (i + 1)
    return scores / len(test_list)
