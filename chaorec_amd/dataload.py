"""Data loading + negative sampling for the hot path (reference dataload.py:21-106).

`data_load` reads the reference's file formats (train.npy int32 [E,2] with GLOBAL item ids, val/test.npy
object arrays of [user, pos...], user_item_dict.npy) and fills the documented gaps: a missing
user_item_dict.npy is rebuilt from train.npy, missing v_feat/t_feat blobs are replaced by
seeded synthetic features, and nothing is moved to the GPU unconditionally.

Two samplers produce the reference's batch format:
  * TrainingDataset  -- the DataLoader-compatible host sampler (same __getitem__ contract);
  * DeviceBatchSampler -- the MI355X path: edge permutation + rejection sampling in one HIP kernel, batches
    never leave HBM.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from . import graph, ops
from .synthetic import DATASET_SHAPES, synthetic_eval_lists, synthetic_interactions

DATASET_SIZES = {  # dataload.py:36-56
    'netfilx': (14971, 7444), 'clothing': (18072, 11384), 'baby': (12351, 4794), 'sports': (28940, 15207),
    'beauty': (15482, 8643), 'electronics': (150179, 51901), 'microlens': (46420, 14079),
}
SYNTHETIC_FEATURE_DIMS = {"default": (4096, 384), "microlens": (128, 768)}  # SURVEY 8(d) assumptions


def synthetic_features(num_item, dataset, seed=0):
    dv, dt = SYNTHETIC_FEATURE_DIMS.get(dataset, SYNTHETIC_FEATURE_DIMS["default"])
    g = torch.Generator().manual_seed(seed)
    return torch.randn(num_item, dv, generator=g), torch.randn(num_item, dt, generator=g)


FIXTURE_DIR = os.environ.get("CHAOREC_DATA_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "tests", "golden"))


def packed_interactions(dataset, fixture_dir=None):
    """The reference's Data/<dataset>/{train,val,test}.npy as shipped with this repository: one
    <dataset>_interactions.npz (train int32 [E,2] with GLOBAL item ids; val/test rows [user, pos...] flattened with
    offsets), written by tests/golden/gen_golden.py / gen_fullsize.py from the reference's files.  None if absent."""
    path = os.path.join(fixture_dir or FIXTURE_DIR, f"{dataset}_interactions.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path, allow_pickle=False)
    vf, vo, tf, to = g["val_flat"], g["val_off"], g["test_flat"], g["test_off"]
    return dict(num_user=int(g["U"]), num_item=int(g["I"]), train=g["train"],
                val=[vf[vo[i]:vo[i + 1]].tolist() for i in range(len(vo) - 1)],
                test=[tf[to[i]:to[i + 1]].tolist() for i in range(len(to) - 1)])


def data_load(dataset, has_v=True, has_t=True, data_root='./Data', synthetic=False):
    """-> train_data, val_data, test_data, user_item_dict, num_user, num_item, v_feat, t_feat (dataload.py:21-58).
    Interactions come from <data_root>/<dataset>/*.npy (the reference's layout) or, failing that, from the packed copy
    of the same files shipped under tests/golden/ (packed_interactions)."""
    num_user, num_item = DATASET_SIZES[dataset]
    dir_str = os.path.join(data_root, dataset)
    packed = None
    if not synthetic and not os.path.exists(os.path.join(dir_str, 'train.npy')):
        packed = packed_interactions(dataset)
    if packed is not None:
        train_data, val_data, test_data = packed["train"], packed["val"], packed["test"]
        user_item_dict = graph.user_item_dict_from_edges(train_data)
    elif synthetic or not os.path.exists(os.path.join(dir_str, 'train.npy')):
        if not synthetic:
            raise FileNotFoundError(f"{dir_str}/train.npy not found and no packed copy under {FIXTURE_DIR} "
                                    f"(pass --synthetic for a generated graph)")
        _, _, E = DATASET_SHAPES[dataset]
        train_data = synthetic_interactions(num_user, num_item, E, seed=42)
        val_data = synthetic_eval_lists(num_user, num_item, train_data, 1, seed=7)
        test_data = synthetic_eval_lists(num_user, num_item, train_data, 1, seed=8)
        user_item_dict = graph.user_item_dict_from_edges(train_data)
    else:
        train_data = np.load(os.path.join(dir_str, 'train.npy'), allow_pickle=True)
        val_data = np.load(os.path.join(dir_str, 'val.npy'), allow_pickle=True)
        test_data = np.load(os.path.join(dir_str, 'test.npy'), allow_pickle=True)
        uid_path = os.path.join(dir_str, 'user_item_dict.npy')
        if os.path.exists(uid_path):
            user_item_dict = np.load(uid_path, allow_pickle=True).item()
        else:
            user_item_dict = graph.user_item_dict_from_edges(train_data)
    v_feat = t_feat = None
    if has_v or has_t:
        vp, tp = os.path.join(dir_str, 'v_feat.npy'), os.path.join(dir_str, 't_feat.npy')
        sv, st = synthetic_features(num_item, dataset)
        if has_v:
            v_feat = torch.tensor(np.load(vp, allow_pickle=True), dtype=torch.float) if os.path.exists(vp) else sv
        if has_t:
            t_feat = torch.tensor(np.load(tp, allow_pickle=True), dtype=torch.float) if os.path.exists(tp) else st
    return train_data, val_data, test_data, user_item_dict, num_user, num_item, v_feat, t_feat


def history_sequences(hist, users, src_len, generator=None):
    """LightGT's per-sample history sequence (dataload.py:89-101,129-141) for a batch of users, on the device: the user's
    history when it has at most src_len items, else a uniformly random subset of src_len of them (the reference shuffles the
    list and cuts it), zero-padded.  -> (user_item int64 [B, src_len + 1]: LOCAL item ids behind a -1 in the user token's
    place, mask bool [B, src_len + 1]: True = padding).  hist: (rowptr, col) with local ids; users: int64 [B]."""
    rowptr, col = hist
    dev = users.device
    B = users.numel()
    start = rowptr[users]
    deg = rowptr[users + 1] - start
    total = int(deg.sum())
    user_item = torch.zeros((B, src_len + 1), dtype=torch.int64, device=dev)
    user_item[:, 0] = -1
    if total:
        seg = torch.repeat_interleave(torch.arange(B, device=dev), deg)
        first = torch.cumsum(deg, 0) - deg                                   # position of every row's first entry
        entry = col[start[seg] + (torch.arange(total, device=dev) - first[seg])].to(torch.int64)
        key = torch.rand(total, device=dev, generator=generator, dtype=torch.float64)
        order = torch.argsort(seg.to(torch.float64) + key)                   # by row, random inside a row
        rank = torch.arange(total, device=dev) - first[seg[order]]
        keep = rank < src_len
        user_item[seg[order][keep], 1 + rank[keep]] = entry[order][keep]
    mask = torch.arange(src_len + 1, device=dev)[None, :] > torch.clamp(deg, max=src_len)[:, None]
    return user_item, mask


def device_eval_batches(hist, num_user, src_len, chunk, device, generator=None):
    """DataLoader(EvalDataset(...), 2000, shuffle=False) (main.py:198-199) on the device: (users, user_item, mask) per chunk."""
    for s in range(0, num_user, chunk):
        users = torch.arange(s, min(s + chunk, num_user), device=device)
        user_item, mask = history_sequences(hist, users, src_len, generator)
        yield users, user_item, mask


class EvalDataset(Dataset):
    """dataload.py:109-143 (LightGT's evaluation batches, host side): per user the shuffled history cut / padded to src_len = 20
    behind a -1, and its padding mask."""

    def __init__(self, num_user, num_item, user_item_dict, src_len=20):
        self.num_user, self.num_item, self.user_item_dict, self.src_len = num_user, num_item, user_item_dict, src_len

    def __len__(self):
        return self.num_user

    def __getitem__(self, index):
        user_item, mask = _host_sequence(self.user_item_dict[index], self.num_user, self.src_len)
        return torch.LongTensor([index]), user_item, mask


def _host_sequence(items, num_user, src_len):
    temp = list(items)
    random.shuffle(temp)
    if len(temp) > src_len:
        mask = torch.ones(src_len + 1) == 0
        temp = temp[:src_len]
    else:
        mask = torch.cat((torch.ones(len(temp) + 1), torch.zeros(src_len - len(temp)))) == 0
        temp.extend([num_user for _ in range(src_len - len(temp))])
    return torch.cat((torch.tensor([-1]), torch.tensor(temp, dtype=torch.int64) - num_user)), mask


class TrainingDataset(Dataset):
    """dataload.py:61-106: one uniform negative per positive, rejected while it is in the user's history.
    Same return contract ([user, pos, neg] ints, or (LongTensor[u,u], LongTensor[pos,neg]) for MMGCN);
    the draw itself is O(1) (randrange + set lookup) instead of the reference's O(I) random.sample(set)."""

    def __init__(self, num_user, num_item, user_item_dict, edge_index, model_name="LightGCN"):
        self.edge_index = edge_index
        self.num_user = num_user
        self.num_item = num_item
        self.user_item_dict = user_item_dict
        self._sets = {}
        self.model_name = model_name
        self.src_len = 50

    def __len__(self):
        return len(self.edge_index)

    def _seen(self, user):
        s = self._sets.get(user)
        if s is None:
            s = self._sets[user] = set(int(i) for i in self.user_item_dict[user])
        return s

    def __getitem__(self, index):
        user, pos_item = self.edge_index[index]
        user, pos_item = int(user), int(pos_item)
        seen = self._seen(user)
        while True:
            neg_item = random.randrange(self.num_user, self.num_user + self.num_item)
            if neg_item not in seen:
                break
        if self.model_name in ["MMGCN", "GRCN"]:
            return torch.LongTensor([user, user]), torch.LongTensor([pos_item, neg_item])
        if self.model_name == "LightGT":            # dataload.py:89-101: + the sample's history sequence and its padding mask
            user_item, mask = _host_sequence(self.user_item_dict[user], self.num_user, self.src_len)
            return [torch.LongTensor([user, user]), torch.LongTensor([pos_item, neg_item]), mask, user_item]
        if self.model_name == "MCLN":               # dataload.py:81-84: the second rejection loop's item
            while True:
                int_item = random.randrange(self.num_user, self.num_user + self.num_item)
                if int_item not in seen:
                    break
            return [user, pos_item, neg_item, int_item]
        return [user, pos_item, neg_item]


class DeviceBatchSampler:
    """Epoch iterator over (users, pos, neg) batches of GLOBAL ids that live in HBM.

    Stands in for DataLoader(TrainingDataset, batch_size, shuffle=True): a device-side permutation of the
    edge list per epoch and chaorec_sample_negatives for the rejection draw (counter-based RNG keyed by
    (seed, global step, position), so a run is reproducible regardless of launch geometry)."""

    def __init__(self, num_user, num_item, user_item_dict, edge_index, batch_size, device, model_name="LightGCN",
                 seed=42):
        self.num_user, self.num_item, self.batch_size = num_user, num_item, batch_size
        self.device, self.model_name, self.seed = device, model_name, seed
        self.edges = torch.as_tensor(np.asarray(edge_index), dtype=torch.int64).to(device)
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.global_step = 0
        # in-launch mode (models with loss_drawn): the epoch permutation, its read position and the global batch
        # counter live in HBM at fixed addresses; the fused BPR forward draws from them and moves them on
        self.perm = self.perm_pos = self.step_dev = None

    def begin_epoch(self):
        """In-launch mode: a fresh permutation of the edge list (the same one __iter__ would draw), read position 0.
        The batches the fused forward then draws are exactly those __iter__ yields for this epoch."""
        E = self.edges.shape[0]
        if self.perm is None:
            self.perm = torch.empty(E, dtype=torch.int64, device=self.device)
            self.perm_pos = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.step_dev = torch.full((1,), self.global_step, dtype=torch.int64, device=self.device)
        self.perm.copy_(torch.randperm(E, device=self.device, generator=self.gen))
        self.perm_pos.zero_()

    def drawn_loss_fn(self, model):
        """loss of the next full batch, drawn inside the fused BPR forward (capturable, no arguments)."""
        return lambda: model.loss_drawn(self.edges, self.batch_size, self.seed, 0, step_dev=self.step_dev,
                                        advance=True, perm=self.perm, perm_pos=self.perm_pos)

    def __len__(self):
        return (self.edges.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        perm = torch.randperm(self.edges.shape[0], device=self.device, generator=self.gen)
        for s in range(0, perm.numel(), self.batch_size):
            e = self.edges[perm[s:s + self.batch_size]]
            users, pos = e[:, 0].contiguous(), e[:, 1].contiguous()
            neg = ops.sample_negatives(self.hist, users, self.num_item, self.seed, self.global_step, self.num_user)
            second = None
            if self.model_name == "MCLN":            # dataload.py:81-84,103-104: a second negative per sample
                second = ops.sample_negatives(self.hist, users, self.num_item, self.seed, self.global_step, self.num_user,
                                              second=True)
            self.global_step += 1
            if self.model_name == "LightGT":          # dataload.py:89-101: history sequences (src_len 50) drawn on the device
                user_item, mask = history_sequences(self.hist, users, 50, self.gen)
                yield torch.stack((users, users), 1), torch.stack((pos, neg), 1), mask, user_item
            elif self.model_name in ["MMGCN", "GRCN"]:
                yield torch.stack((users, users), 1), torch.stack((pos, neg), 1)
            elif second is not None:
                yield users, pos, neg, second
            else:
                yield users, pos, neg
