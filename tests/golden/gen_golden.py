#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own classes.

Runs only in the build container (needs /root/reference; the GPU box has no reference).
Nothing from the reference is copied: this script imports its modules, feeds them seeded
inputs and stores inputs + outputs as .npz data.

    python tests/golden/gen_golden.py

LightGCN / MMGCN import torch_geometric, which is not installed: oracle/pyg_standin.py provides
the restated propagate (see its docstring), so those goldens pin "reference model code + restated
third-party propagate".  FREEDOM, metrics.py, utils.gene_metrics and dataload.TrainingDataset are
pure reference.
"""
import os
import random
import sys
import warnings

import numpy as np
import torch

REF = os.environ.get("CHAOREC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyg_standin  # noqa: E402

pyg_standin.install()
sys.path.insert(0, REF)
_argv = sys.argv
sys.argv = ["main.py", "--Model", "LightGCN", "--data_path", "baby"]  # parse_args() runs at import
warnings.filterwarnings("ignore")
from Model.LightGCN import LightGCN  # noqa: E402
from Model.FREEDOM import FREEDOM  # noqa: E402
from Model.MMGCN import MMGCN  # noqa: E402
from Model.NGCF import NGCF  # noqa: E402
from Model.MGCN import MGCN  # noqa: E402
from Model.LayerGCN import LayerGCN  # noqa: E402
from Model.BPR import BPRMF  # noqa: E402
from Model.VBPR import VBPR  # noqa: E402
import metrics as ref_metrics  # noqa: E402,F401
import utils as ref_utils  # noqa: E402
import dataload as ref_dataload  # noqa: E402

sys.argv = _argv
torch.set_num_threads(8)
DEV = torch.device("cpu")


def tiny_graph():
    """6 users x 5 items, sorted by user, global item ids = item + 6; item 4 is isolated."""
    U, I = 6, 5
    pairs = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 3), (1, 0), (2, 2), (2, 3), (2, 0),
             (3, 0), (3, 1), (3, 3), (4, 2), (4, 1), (4, 3), (5, 3), (5, 0), (5, 2)]
    e = np.array([(u, i + U) for u, i in pairs], dtype=np.int32)
    return U, I, e


def uid(edges):
    d = {}
    for u, i in edges.tolist():
        d.setdefault(u, []).append(i)
    return d


def seeded(shape, seed, scale=0.1):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def lightgcn_case(U, I, edges, D, L, reg, batch, x0):
    m = LightGCN(U, I, edges, uid(edges), D, reg, L, "add", DEV)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(x0[:U]))
        m.item_embedding.weight.copy_(torch.from_numpy(x0[U:]))
    users, pos, neg = (torch.from_numpy(b) for b in batch)
    # per-layer outputs
    with torch.no_grad():
        x = torch.cat((m.user_embedding.weight, m.item_embedding.weight), 0)
        layers = [x.numpy().copy()]
        for conv in m.conv_layers:
            x = conv(x, m.edge_index)
            layers.append(x.numpy().copy())
    loss = m.loss(users, pos, neg)
    loss.backward()
    with torch.no_grad():
        emb = m.result
        bpr = m.bpr_loss(users, pos - U, neg - U, emb)
        regl = m.regularization_loss(users, pos - U, neg - U, emb)
    rank = m.gene_ranklist()
    with torch.no_grad():
        sc = m.result[:U] @ m.result[U:].t()
        for r, c in m.user_item_dict.items():
            sc[r][torch.LongTensor(list(c)) - U] = 1e-6
        rank_val = torch.gather(sc, 1, rank - U)
    return dict(layers=np.stack(layers), result=m.result.detach().numpy(), loss=np.float64(loss.item()),
                bpr=np.float64(bpr.item()), reg_loss=np.float64(regl.item()),
                g_user=m.user_embedding.weight.grad.numpy(), g_item=m.item_embedding.weight.grad.numpy(),
                rank=rank.numpy(), rank_val=rank_val.numpy()), m


def gen_lightgcn_tiny():
    U, I, e = tiny_graph()
    D, L, reg = 8, 2, 1e-3
    x0 = seeded((U + I, D), 1, 0.5)
    batch = (np.array([0, 1, 2, 3, 4, 5, 0, 3], np.int64),
             np.array([6, 9, 8, 7, 9, 6, 8, 9], np.int64),      # global ids
             np.array([9, 8, 7, 8, 6, 7, 10, 10], np.int64))
    # topk=50 > 5 items: the reference's gene_ranklist would raise, so rank with a patched default
    LightGCN.gene_ranklist.__defaults__ = (3,)
    out, _ = lightgcn_case(U, I, e, D, L, reg, batch, x0)
    LightGCN.gene_ranklist.__defaults__ = (50,)
    np.savez_compressed(os.path.join(HERE, "lightgcn_tiny.npz"), U=U, I=I, edges=e, D=D, L=L, reg=reg,
                        x0=x0, users=batch[0], pos=batch[1], neg=batch[2], topk=3, **out)


def load_baby():
    d = os.path.join(REF, "Data", "baby")
    train = np.load(os.path.join(d, "train.npy"), allow_pickle=True)
    val = np.load(os.path.join(d, "val.npy"), allow_pickle=True)
    test = np.load(os.path.join(d, "test.npy"), allow_pickle=True)
    uidict = np.load(os.path.join(d, "user_item_dict.npy"), allow_pickle=True).item()
    return 12351, 4794, train, val, test, uidict


def ragged(obj_arr):
    """object array of [user, pos...] lists -> (flat int32, offsets int64)."""
    lens = np.array([len(x) for x in obj_arr], dtype=np.int64)
    off = np.zeros(len(obj_arr) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.concatenate([np.asarray(x, dtype=np.int32) for x in obj_arr])
    return flat, off


def gen_baby():
    U, I, train, val, test, uidict = load_baby()
    # the interactions themselves (data shipped by the reference; dataload.py:26-30)
    vf, vo = ragged(val)
    tf, to = ragged(test)
    np.savez_compressed(os.path.join(HERE, "baby_interactions.npz"), U=U, I=I, train=train.astype(np.int32),
                        val_flat=vf, val_off=vo, test_flat=tf, test_off=to)
    # reconstruction rule for user_item_dict verified against the shipped dict
    rebuilt = uid(train)
    assert list(rebuilt.keys()) == [int(k) for k in uidict.keys()]
    assert all([int(v) for v in uidict[k]] == rebuilt[int(k)] for k in uidict)

    D, L, reg = 64, 2, 1e-3
    x0 = seeded((U + I, D), 42, 0.1)
    rng = np.random.default_rng(7)
    bidx = rng.choice(len(train), 1024, replace=False)
    users = train[bidx, 0].astype(np.int64)
    pos = train[bidx, 1].astype(np.int64)
    neg = np.empty(1024, np.int64)
    for k, u in enumerate(users):
        while True:
            c = int(rng.integers(U, U + I))
            if c not in rebuilt[int(u)]:
                neg[k] = c
                break
    out, m = lightgcn_case(U, I, train, D, L, reg, (users, pos, neg), x0)
    k_list = [5, 10, 20]
    val_m = ref_utils.gene_metrics(val, torch.from_numpy(out["rank"]), k_list)
    test_m = ref_utils.gene_metrics(test, torch.from_numpy(out["rank"]), k_list)
    names = ["precision", "recall", "ndcg", "hit_rate", "map"]
    rows = np.sort(rng.choice(U + I, 512, replace=False))
    urows = np.sort(rng.choice(U, 1024, replace=False))
    np.savez_compressed(
        os.path.join(HERE, "lightgcn_baby.npz"), D=D, L=L, reg=reg, x0_seed=42, x0_scale=0.1,
        users=users, pos=pos, neg=neg, loss=out["loss"], bpr=out["bpr"], reg_loss=out["reg_loss"],
        rows=rows, result_rows=out["result"][rows], layer_rows=out["layers"][:, rows],
        g_rows=np.concatenate([out["g_user"], out["g_item"]], 0)[rows],
        g_abs_sum=np.float64(np.abs(out["g_user"]).sum() + np.abs(out["g_item"]).sum()),
        result_sum=np.float64(out["result"].astype(np.float64).sum()),
        urows=urows, rank_rows=out["rank"][urows].astype(np.int32), rank_val_rows=out["rank_val"][urows],
        k_list=np.array(k_list), metric_names=np.array(names),
        val_metrics=np.array([[val_m[k][n] for n in names] for k in k_list]),
        test_metrics=np.array([[test_m[k][n] for n in names] for k in k_list]))
    # metrics fixture on a FIXED (seeded, not model-produced) rank list: pins metrics.py alone
    fixed_rank = np.stack([np.random.default_rng(1000 + u).permutation(I)[:50] + U for u in range(U)])
    fm = ref_utils.gene_metrics(val, torch.from_numpy(fixed_rank), k_list)
    np.savez_compressed(os.path.join(HERE, "metrics_baby_fixed_rank.npz"), k_list=np.array(k_list),
                        metric_names=np.array(names),
                        val_metrics=np.array([[fm[k][n] for n in names] for k in k_list]))


def gen_sampler():
    """dataload.py:61-106 on the tiny graph: draw every training edge many times."""
    U, I, e = tiny_graph()
    ds = ref_dataload.TrainingDataset(U, I, uid(e), e)
    random.seed(42)
    hist = np.zeros((U, I), np.int64)
    reps = 400
    for _ in range(reps):
        for idx in range(len(e)):
            u, p, n = ds[idx]
            assert p == e[idx, 1]
            hist[u, n - U] += 1
    np.savez_compressed(os.path.join(HERE, "sampler_tiny.npz"), U=U, I=I, edges=e, reps=reps, neg_hist=hist)


def freedom_case(U, I, edges, v_feat, t_feat, D, L, mm_layers, knn, w, dropout, reg, batch, x0, topk):
    torch.manual_seed(0)
    m = FREEDOM(U, I, edges, uid(edges), torch.from_numpy(v_feat), torch.from_numpy(t_feat), D, D, reg,
                dropout, L, mm_layers, knn, w, DEV)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(x0[:U]))
        m.item_embedding.weight.copy_(torch.from_numpy(x0[U:]))
    torch.manual_seed(123)
    m.pre_epoch_processing()
    masked = m.masked_adj.coalesce() if dropout > 0 else m.masked_adj.coalesce()
    masked_raw = m.masked_adj
    users, pos, neg = (torch.from_numpy(b) for b in batch)
    loss = m.loss(users, pos, neg)
    loss.backward()
    FREEDOM.gene_ranklist.__defaults__ = (topk,)
    rank = m.gene_ranklist()
    FREEDOM.gene_ranklist.__defaults__ = (50,)
    mm = m.mm_adj.coalesce()
    norm = m.norm_adj.coalesce()
    out = dict(
        edge_values=m.edge_values.numpy(), edge_indices=m.edge_indices.numpy(),
        mm_idx=mm.indices().numpy(), mm_val=mm.values().numpy(),
        norm_idx=norm.indices().numpy(), norm_val=norm.values().numpy(),
        masked_idx_raw=masked_raw._indices().numpy(), masked_val_raw=masked_raw._values().numpy(),
        masked_idx=masked.indices().numpy(), masked_val=masked.values().numpy(),
        result=m.result.detach().numpy(), loss=np.float64(loss.item()),
        g_user=m.user_embedding.weight.grad.numpy(), g_item=m.item_embedding.weight.grad.numpy(),
        g_image_trs_w=m.image_trs.weight.grad.numpy(), g_image_trs_b=m.image_trs.bias.grad.numpy(),
        g_text_trs_w=m.text_trs.weight.grad.numpy(), g_text_trs_b=m.text_trs.bias.grad.numpy(),
        g_image_emb=m.image_embedding.weight.grad.numpy(), g_text_emb=m.text_embedding.weight.grad.numpy(),
        image_trs_w=m.image_trs.weight.detach().numpy(), image_trs_b=m.image_trs.bias.detach().numpy(),
        text_trs_w=m.text_trs.weight.detach().numpy(), text_trs_b=m.text_trs.bias.detach().numpy(),
        rank=rank.numpy())
    return out


def small_graph(U, I, deg_lo, deg_hi, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for u in range(U):
        k = int(rng.integers(deg_lo, deg_hi + 1))
        for i in rng.choice(I, k, replace=False):
            rows.append((u, int(i) + U))
    return np.array(rows, dtype=np.int32)


def gen_freedom():
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 3)
    D = 16
    v_feat = seeded((I, 24), 11, 1.0)
    t_feat = seeded((I, 12), 12, 1.0)
    x0 = seeded((U + I, D), 13, 0.3)
    rng = np.random.default_rng(5)
    b = rng.choice(len(e), 32, replace=False)
    users = e[b, 0].astype(np.int64)
    pos = e[b, 1].astype(np.int64)
    neg = rng.integers(U, U + I, 32).astype(np.int64)
    for dropout, tag in ((0.1, "drop"), (0.0, "nodrop")):
        out = freedom_case(U, I, e, v_feat, t_feat, D, 2, 1, 5, 0.8, dropout, 1e-3, (users, pos, neg), x0, 10)
        np.savez_compressed(os.path.join(HERE, f"freedom_small_{tag}.npz"), U=U, I=I, edges=e, D=D, L=2,
                            mm_layers=1, knn=5, w=0.8, dropout=dropout, reg=1e-3, v_feat=v_feat, t_feat=t_feat,
                            x0=x0, users=users, pos=pos, neg=neg, topk=10, **out)


def gen_mmgcn():
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 4)
    dim_x = 64
    v_feat = seeded((I, 20), 21, 1.0)
    t_feat = seeded((I, 12), 22, 1.0)
    torch.manual_seed(0)
    m = MMGCN(U, I, e, uid(e), torch.from_numpy(v_feat), torch.from_numpy(t_feat), dim_x, 1e-4, "add", "False",
              True, DEV)
    state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    extra = dict(v_pref=m.v_gcn.preference.detach().numpy().copy(), t_pref=m.t_gcn.preference.detach().numpy().copy(),
                 id_embedding=m.id_embedding.detach().numpy().copy(), init_result=m.result.detach().numpy().copy())
    rng = np.random.default_rng(6)
    b = rng.choice(len(e), 16, replace=False)
    user_tensor = np.stack([e[b, 0], e[b, 0]], 1).astype(np.int64)
    neg = rng.integers(U, U + I, 16)
    item_tensor = np.stack([e[b, 1], neg], 1).astype(np.int64)
    loss = m.loss(torch.from_numpy(user_tensor), torch.from_numpy(item_tensor))
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    MMGCN.gene_ranklist.__defaults__ = (20, 10)
    rank = m.gene_ranklist()
    MMGCN.gene_ranklist.__defaults__ = (200, 50)
    np.savez_compressed(os.path.join(HERE, "mmgcn_small.npz"), U=U, I=I, edges=e, dim_x=dim_x, reg=1e-4,
                        v_feat=v_feat, t_feat=t_feat, user_tensor=user_tensor, item_tensor=item_tensor,
                        loss=np.float64(loss.item()), result=m.result.detach().numpy(), rank=rank.numpy(),
                        topk=10, step=20, param_names=np.array([k for k, _ in m.named_parameters()]),
                        **{"p_" + k: v for k, v in state.items()}, **extra, **grads)


def gen_ngcf():
    """Reference NGCF (model code + restated propagate / dropout_adj): one loss + backward + ranking, without and
    with edge dropout.  The keep masks dropout_adj drew are stored as inputs (bidirectional edge-list order)."""
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 8)
    D, L = 16, 2
    rng = np.random.default_rng(9)
    b = rng.choice(len(e), 32, replace=False)
    users = e[b, 0].astype(np.int64)
    pos = e[b, 1].astype(np.int64)
    neg = rng.integers(U, U + I, 32).astype(np.int64)
    for dropout, tag in ((0.0, "nodrop"), (0.3, "drop")):
        torch.manual_seed(0)
        m = NGCF(U, I, e, uid(e), D, 1e-3, dropout, L, "add", DEV)
        state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        del pyg_standin.DROPOUT_LOG[:]
        torch.manual_seed(77)
        loss = m.loss(torch.from_numpy(users), torch.from_numpy(pos), torch.from_numpy(neg))
        loss.backward()
        masks = np.stack([k.numpy() for k in pyg_standin.DROPOUT_LOG]) if dropout > 0 else np.zeros((0, 2 * len(e)), bool)
        assert masks.shape[0] == (L if dropout > 0 else 0)
        grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
        NGCF.gene_ranklist.__defaults__ = (10,)
        rank = m.gene_ranklist()
        NGCF.gene_ranklist.__defaults__ = (50,)
        np.savez_compressed(os.path.join(HERE, f"ngcf_small_{tag}.npz"), U=U, I=I, edges=e, D=D, L=L, reg=1e-3,
                            dropout=dropout, users=users, pos=pos, neg=neg, keep_masks=masks,
                            loss=np.float64(loss.item()), result=m.result.detach().numpy(), rank=rank.numpy(), topk=10,
                            param_names=np.array([k for k, _ in m.named_parameters()]),
                            **{"p_" + k: v for k, v in state.items()}, **grads)


def gen_mgcn():
    """Reference MGCN (a member of the torch.sparse.mm family; deterministic forward): construction,
    one loss + backward, ranking.  torch_scatter.scatter_add is the restated one of oracle/pyg_standin.py."""
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 14)
    D = 16
    v_feat = seeded((I, 24), 31, 1.0)
    t_feat = seeded((I, 12), 32, 1.0)
    rng = np.random.default_rng(15)
    b = rng.choice(len(e), 32, replace=False)
    users = e[b, 0].astype(np.int64)
    pos = e[b, 1].astype(np.int64)
    neg = rng.integers(U, U + I, 32).astype(np.int64)
    torch.manual_seed(0)
    m = MGCN(U, I, e, uid(e), torch.from_numpy(v_feat), torch.from_numpy(t_feat), D, 1e-4, 2, "add", 0.2, 0.01, DEV)
    state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    loss = m.loss(torch.from_numpy(users), torch.from_numpy(pos), torch.from_numpy(neg))
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    MGCN.gene_ranklist.__defaults__ = (10,)
    rank = m.gene_ranklist()
    MGCN.gene_ranklist.__defaults__ = (50,)
    coo = lambda t: (t.coalesce().indices().numpy(), t.coalesce().values().numpy())
    (na_i, na_v), (r_i, r_v) = coo(m.norm_adj), coo(m.R)
    (im_i, im_v), (tx_i, tx_v) = coo(m.image_original_adj), coo(m.text_original_adj)
    np.savez_compressed(os.path.join(HERE, "mgcn_small.npz"), U=U, I=I, edges=e, D=D, reg=1e-4, ssl_temp=0.2,
                        ssl_alpha=0.01, v_feat=v_feat, t_feat=t_feat, users=users, pos=pos, neg=neg,
                        loss=np.float64(loss.item()), result=m.result.detach().numpy(), rank=rank.numpy(), topk=10,
                        norm_adj_idx=na_i, norm_adj_val=na_v, R_idx=r_i, R_val=r_v, image_adj_idx=im_i,
                        image_adj_val=im_v, text_adj_idx=tx_i, text_adj_val=tx_v,
                        param_names=np.array([k for k, _ in m.named_parameters()]),
                        **{"p_" + k: v for k, v in state.items()}, **grads)


def gen_layergcn():
    """Reference LayerGCN (torch.sparse.mm family; LightGCN propagate + cosine layer weights + FREEDOM-style pruning).
    Its get_norm_adj_mat calls scipy's private dok_matrix._update (Model/LayerGCN.py:65), gone from the scipy
    installed here: the generator restores it as "assign every (row, col) -> value of the dict" -- what the old
    method did -- so this golden pins "reference model code + that one restated scipy method"."""
    import scipy.sparse as sp
    if not hasattr(sp.dok_matrix, "_update"):
        def _update(self, data):
            for (r, c), v in data.items():
                self[r, c] = v
        sp.dok_matrix._update = _update
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 21)
    D, L = 16, 3
    rng = np.random.default_rng(22)
    b = rng.choice(len(e), 32, replace=False)
    users = e[b, 0].astype(np.int64)
    pos = e[b, 1].astype(np.int64)
    neg = rng.integers(U, U + I, 32).astype(np.int64)
    torch.manual_seed(0)
    m = LayerGCN(U, I, e, uid(e), D, 1e-3, L, 0.2, DEV)
    state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    norm = m.norm_adj_matrix.coalesce()
    out = dict(norm_idx=norm.indices().numpy(), norm_val=norm.values().numpy(),
               edge_indices=m.edge_indices.numpy(), edge_values=m.edge_values.numpy())
    import random as pyrandom
    torch.manual_seed(5)
    pyrandom.seed(5)
    for ep in range(2):                      # epoch 0 prunes by multinomial, epoch 1 by random.sample (:101-105)
        m.pre_epoch_processing()
        raw = m.masked_adj
        co = raw.coalesce()
        out[f"masked_idx_raw{ep}"] = raw._indices().numpy().copy()
        out[f"masked_idx{ep}"], out[f"masked_val{ep}"] = co.indices().numpy(), co.values().numpy()
    loss = m.loss(torch.from_numpy(users), torch.from_numpy(pos), torch.from_numpy(neg))
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    LayerGCN.gene_ranklist.__defaults__ = (10,)
    rank = m.gene_ranklist()
    LayerGCN.gene_ranklist.__defaults__ = (50,)
    m.forward_adj = m.norm_adj_matrix
    ue, ie = m.forward()
    np.savez_compressed(os.path.join(HERE, "layergcn_small.npz"), U=U, I=I, edges=e, D=D, L=L, reg=1e-3, dropout=0.2,
                        users=users, pos=pos, neg=neg, loss=np.float64(loss.item()), rank=rank.numpy(), topk=10,
                        eval_result=torch.cat([ue, ie]).detach().numpy(),
                        param_names=np.array([k for k, _ in m.named_parameters()]),
                        **{"p_" + k: v for k, v in state.items()}, **grads, **out)


def gen_bpr_family():
    """Reference BPRMF and VBPR (models that are the BPR kernel + the shared ranking): initial state, one
    loss + backward, ranking; for BPRMF the item bias is given non-zero values first (it is initialised to zero, which
    would hide it from the loss)."""
    U, I = 48, 40
    e = small_graph(U, I, 3, 7, 21)
    rng = np.random.default_rng(22)
    b = rng.choice(len(e), 32, replace=False)
    users, pos = e[b, 0].astype(np.int64), e[b, 1].astype(np.int64)
    neg = rng.integers(U, U + I, 32).astype(np.int64)
    batch = tuple(torch.from_numpy(x) for x in (users, pos, neg))
    # BPRMF
    torch.manual_seed(0)
    m = BPRMF(U, I, uid(e), 16, 1e-3, DEV)
    with torch.no_grad():
        m.item_bias.weight.copy_(torch.from_numpy(seeded((I, 1), 23, 0.3)))
    state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    loss = m.loss(*batch)
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    rank = m.gene_ranklist(topk=10)
    np.savez_compressed(os.path.join(HERE, "bprmf_small.npz"), U=U, I=I, edges=e, D=16, reg=1e-3, users=users, pos=pos,
                        neg=neg, loss=np.float64(loss.item()), rank=rank.numpy(), topk=10,
                        param_names=np.array([k for k, _ in m.named_parameters()]),
                        **{"p_" + k: v for k, v in state.items()}, **grads)
    # VBPR
    v_feat = seeded((I, 24), 24, 1.0)
    torch.manual_seed(0)
    m = VBPR(U, I, uid(e), torch.from_numpy(v_feat), 16, 64, 1e-3, DEV)
    state = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    loss = m.loss(*batch)
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().copy() for k, p in m.named_parameters()}
    rank = m.gene_ranklist(topk=10)
    np.savez_compressed(os.path.join(HERE, "vbpr_small.npz"), U=U, I=I, edges=e, D=16, reg=1e-3, users=users, pos=pos,
                        neg=neg, v_feat=v_feat, loss=np.float64(loss.item()), result=m.result.detach().numpy(),
                        rank=rank.numpy(), topk=10, param_names=np.array([k for k, _ in m.named_parameters()]),
                        **{"p_" + k: v for k, v in state.items()}, **grads)


if __name__ == "__main__":
    which = sys.argv[1:] or ["lightgcn_tiny", "baby", "sampler", "freedom", "mmgcn", "ngcf", "mgcn", "layergcn", "bpr_family"]
    for w in which:
        print("generating", w, flush=True)
        globals()["gen_" + w]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f:40s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB")
