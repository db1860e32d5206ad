"""Seed / device / early stopping / metric aggregation for the hot path (reference utils.py).

gene_metrics is the vectorised form of utils.py:112-139 + metrics.py: one hit matrix
[users, max_k] built with a sorted membership test, then every metric at every k from prefix sums.
Numbers equal the reference's per-user python loops (checked against its goldens)."""
import datetime
import random

import numpy as np
import torch


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def gpu():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def get_local_time():
    return datetime.datetime.now().strftime('%b-%d-%Y-%H-%M-%S')


class EarlyStopping:
    """utils.py:57-79: stop after `patience` evaluations without reaching the best score again."""

    def __init__(self, patience=50, verbose=True):
        self.patience = patience
        self.verbose = verbose
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.best_metrics = None

    def __call__(self, score, metrics):
        if self.best_score is None:
            self.best_score = score
            self.best_metrics = metrics
        elif score < self.best_score:
            self.counter += 1
            if self.verbose:
                print(f'EarlyStopping counter: {self.counter} out of {self.patience}')
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_score = score
            self.best_metrics = metrics
            self.counter = 0


def gene_metrics(val_data, rank_list, k_list):
    """val_data: sequence of [user, pos...]; rank_list: [U, >=max(k)] global item ids (tensor or array).
    -> {k: {'precision','recall','ndcg','hit_rate','map'}} averaged over len(val_data)."""
    k_list = [int(k) for k in k_list]
    rank = rank_list.cpu().numpy() if isinstance(rank_list, torch.Tensor) else np.asarray(rank_list)
    kmax = min(max(k_list), rank.shape[1])          # (ranked_items[:k] of a shorter list is the whole list: utils.py:124)
    n = len(val_data)
    users = np.fromiter((int(d[0]) for d in val_data), dtype=np.int64, count=n)
    lens = np.fromiter((len(d) - 1 for d in val_data), dtype=np.int64, count=n)
    flat_items = np.fromiter((int(x) for d in val_data for x in d[1:]), dtype=np.int64, count=int(lens.sum()))
    flat_rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    stride = int(max(rank.max(initial=0), flat_items.max(initial=0))) + 1
    pos_keys = np.unique(flat_rows * stride + flat_items)             # set(test_list) per row
    top = rank[users][:, :kmax].astype(np.int64)                        # ranked_items[:k]
    keys = np.arange(n, dtype=np.int64)[:, None] * stride + top
    hit = np.isin(keys, pos_keys)                                       # [n, kmax]
    # duplicates inside ranked[:k] count once in the set intersection; keep the first occurrence only
    first = np.ones_like(hit)
    srt = np.argsort(top, axis=1, kind="stable")
    ts = np.take_along_axis(top, srt, 1)
    dup_sorted = np.zeros_like(hit)
    dup_sorted[:, 1:] = ts[:, 1:] == ts[:, :-1]
    np.put_along_axis(first, srt, ~dup_sorted, 1)
    hit_set = hit & first
    disc = 1.0 / np.log(np.arange(max(k_list)) + 2.0)
    names = ("precision", "recall", "ndcg", "hit_rate", "map")
    out = {k: dict.fromkeys(names, 0.0) for k in k_list}
    has = lens > 0
    safe_len = np.maximum(lens, 1).astype(np.float64)
    idcg_prefix = np.concatenate([[0.0], np.cumsum(disc)])
    for k_asked in k_list:
        k = min(k_asked, kmax)
        inter = hit_set[:, :k].sum(1).astype(np.float64)
        out[k_asked]["precision"] = float((inter / k_asked).sum() / n)
        out[k_asked]["recall"] = float(np.where(has, inter / safe_len, 0.0).sum() / n)
        dcg = (hit[:, :k] * disc[:k]).sum(1)                            # `item in test_list` per position
        idcg = idcg_prefix[np.minimum(lens, k_asked)]                   # (metrics.py:34: min(len(test), k) with the k asked for)
        out[k_asked]["ndcg"] = float(np.where(has, dcg / np.where(has, idcg, 1.0), 0.0).sum() / n)
        out[k_asked]["hit_rate"] = float((inter > 0).sum() / n)
        cum = np.cumsum(hit[:, :k], 1)
        ap = (hit[:, :k] * cum / np.arange(1, k + 1)).sum(1)
        out[k_asked]["map"] = float(np.where(has, ap / safe_len, 0.0).sum() / n)
    return out


class EvalLists:
    """val.npy / test.npy rows ([user, pos...]) as a device CSR, built once per split (the lists never change)."""

    def __init__(self, data, device):
        n = len(data)
        users = np.fromiter((int(d[0]) for d in data), dtype=np.int64, count=n)
        lens = np.fromiter((len(d) - 1 for d in data), dtype=np.int64, count=n)
        items = np.fromiter((int(x) for d in data for x in d[1:]), dtype=np.int64, count=int(lens.sum()))
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.n = n
        self.row_user = torch.from_numpy(users).to(device)
        self.rowptr = torch.from_numpy(rowptr).to(device)
        self.items = torch.from_numpy(items if len(items) else np.zeros(1, np.int64)).to(device)


_NAMES = ("precision", "recall", "ndcg", "hit_rate", "map")


def gene_metrics_device(eval_lists, rank_idx, k_list):
    """utils.gene_metrics (utils.py:112-139) on the GPU: rank_idx [U, >=max(k)] int64 in HBM (gene_ranklist(to_cpu=False)),
    eval_lists an EvalLists.  One launch + a 3x5 fp64 read-back instead of the [U, 50] transfer and the host loop."""
    from . import ops
    k_list = [int(k) for k in k_list]
    out = ops.rank_metrics(rank_idx, eval_lists.row_user, eval_lists.rowptr, eval_lists.items, k_list).cpu().numpy()
    return {k: {name: float(out[i, j]) for j, name in enumerate(_NAMES)} for i, k in enumerate(k_list)}
