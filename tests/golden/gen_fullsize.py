#!/usr/bin/env python3
"""Full-size golden fixtures: the reference's own classes on the REAL interaction files of BASELINE.json's configs
(Data/{sports,clothing,microlens}, plus baby for the trajectory), in the build container.

    python tests/golden/gen_fullsize.py [interactions lightgcn_sports trajectory_baby trajectory_sports
                                         freedom_clothing mmgcn_microlens]

Stored: the interaction files themselves as .npz data (the GPU box has no /root/reference), and per config the
inputs that are not reproducible from a seed (batches, the kept-edge mask the reference drew) together with the
reference's outputs on row subsets + checksums.  Everything else (initial weights, synthetic features) is a pure
function of a torch CPU seed and the constructor's call order, which the product classes keep -- the tests check
that on stored sample rows before they compare anything else.  Nothing of the reference's source is stored.
Same caveat as gen_golden.py: LightGCN / MMGCN use the restated torch_geometric propagate of oracle/pyg_standin.py;
FREEDOM, gene_metrics and the interaction files are pure reference.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the stand-in, imports the reference's classes)

REF = G.REF
SIZES = {"baby": (12351, 4794), "sports": (28940, 15207), "clothing": (18072, 11384), "microlens": (46420, 14079)}
NAMES = ["precision", "recall", "ndcg", "hit_rate", "map"]
K_LIST = [5, 10, 20]


def load_real(name):
    d = os.path.join(REF, "Data", name)
    train = np.load(os.path.join(d, "train.npy"), allow_pickle=True).astype(np.int32)
    val = np.load(os.path.join(d, "val.npy"), allow_pickle=True)
    test = np.load(os.path.join(d, "test.npy"), allow_pickle=True)
    U, I = SIZES[name]
    return U, I, train, val, test


def gen_interactions():
    for name in ("sports", "clothing", "microlens"):
        U, I, train, val, test = load_real(name)
        vf, vo = G.ragged(val)
        tf, to = G.ragged(test)
        np.savez_compressed(os.path.join(HERE, f"{name}_interactions.npz"), U=U, I=I, train=train, val_flat=vf,
                            val_off=vo, test_flat=tf, test_off=to)
        # user_item_dict.npy is a missing blob for sports / microlens: the rule "train edges grouped by user in file
        # order" is verified where the blob exists (clothing here, baby in gen_golden.py)
        p = os.path.join(REF, "Data", name, "user_item_dict.npy")
        if os.path.exists(p):
            shipped = np.load(p, allow_pickle=True).item()
            rebuilt = G.uid(train)
            assert list(rebuilt.keys()) == [int(k) for k in shipped.keys()]
            assert all([int(v) for v in shipped[k]] == rebuilt[int(k)] for k in shipped)


def fixed_batches(train, U, I, hist, T, B, seed):
    """T batches of B distinct training edges + one rejected-uniform negative each (global ids)."""
    rng = np.random.default_rng(seed)
    out = np.empty((T, 3, B), np.int64)
    for t in range(T):
        b = rng.choice(len(train), B, replace=False)
        out[t, 0] = train[b, 0]
        out[t, 1] = train[b, 1]
        for k, u in enumerate(out[t, 0]):
            while True:
                c = int(rng.integers(U, U + I))
                if c not in hist[int(u)]:
                    out[t, 2, k] = c
                    break
    return out


def masked_scores(result, U, uidict, rank, mask):
    sc = result[:U] @ result[U:].t()
    for r, c in uidict.items():
        sc[r][torch.LongTensor(list(c)) - U] = mask
    return torch.gather(sc, 1, rank - U)


def metrics_table(data, rank):
    m = G.ref_utils.gene_metrics(data, rank, K_LIST)
    return np.array([[m[k][n] for n in NAMES] for k in K_LIST])


def sample_rows(t, rows):
    return t.detach().numpy()[rows].copy()


def gen_lightgcn_sports():
    """configs[1]: reference LightGCN on the real sports graph, D=64, L=3, B=1024: one loss + backward + ranking +
    metrics from the seed-42 initialisation."""
    U, I, train, val, test = load_real("sports")
    uidict = G.uid(train)
    hist = {u: set(v) for u, v in uidict.items()}
    D, L, reg = 64, 3, 1e-3
    torch.manual_seed(42)
    m = G.LightGCN(U, I, train, uidict, D, reg, L, "add", G.DEV)
    x0 = torch.cat((m.user_embedding.weight, m.item_embedding.weight), 0).detach().clone()
    batch = fixed_batches(train, U, I, hist, 1, 1024, 7)[0]
    users, pos, neg = (torch.from_numpy(b) for b in batch)
    with torch.no_grad():
        x = x0
        layers = [x.numpy().copy()]
        for conv in m.conv_layers:
            x = conv(x, m.edge_index)
            layers.append(x.numpy().copy())
    loss = m.loss(users, pos, neg)
    loss.backward()
    with torch.no_grad():
        bpr = m.bpr_loss(users, pos - U, neg - U, m.result)
        regl = m.regularization_loss(users, pos - U, neg - U, m.result)
    rank = m.gene_ranklist()
    with torch.no_grad():
        rank_val = masked_scores(m.result, U, uidict, rank, 1e-6)
    rng = np.random.default_rng(11)
    rows = np.sort(rng.choice(U + I, 512, replace=False))
    urows = np.sort(rng.choice(U, 1024, replace=False))
    g_all = torch.cat((m.user_embedding.weight.grad, m.item_embedding.weight.grad), 0)
    np.savez_compressed(
        os.path.join(HERE, "lightgcn_sports.npz"), D=D, L=L, reg=reg, init_seed=42, users=batch[0], pos=batch[1],
        neg=batch[2], rows=rows, x0_rows=sample_rows(x0, rows), x0_sum=np.float64(x0.double().sum().item()),
        layer_rows=np.stack(layers)[:, rows], result_rows=sample_rows(m.result, rows),
        result_sum=np.float64(m.result.double().sum().item()), loss=np.float64(loss.item()),
        bpr=np.float64(bpr.item()), reg_loss=np.float64(regl.item()), g_rows=sample_rows(g_all, rows),
        g_abs_sum=np.float64(g_all.double().abs().sum().item()), urows=urows,
        rank_rows=rank.numpy()[urows].astype(np.int32), rank_val_rows=rank_val.numpy()[urows],
        k_list=np.array(K_LIST), metric_names=np.array(NAMES), val_metrics=metrics_table(val, rank),
        test_metrics=metrics_table(test, rank))


def trajectory(name, L, T):
    """Row L of SURVEY 8(a): the reference's generic train branch (train_and_evaluate.py:43-48: zero_grad, loss,
    backward, Adam step, main.py:397 lr 1e-3) over T fixed batches, then the evaluation of :655-659 on the STALE
    self.result (quirk Q4: the forward of step T, i.e. the weights after T-1 updates)."""
    U, I, train, val, test = load_real(name)
    uidict = G.uid(train)
    hist = {u: set(v) for u, v in uidict.items()}
    D, reg = 64, 1e-3
    torch.manual_seed(42)
    m = G.LightGCN(U, I, train, uidict, D, reg, L, "add", G.DEV)
    opt = torch.optim.Adam([{"params": m.parameters(), "lr": 1e-3}])
    batches = fixed_batches(train, U, I, hist, T, 1024, 21)
    losses = []
    for t in range(T):
        users, pos, neg = (torch.from_numpy(b) for b in batches[t])
        opt.zero_grad()
        loss = m.loss(users, pos, neg)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    rank = m.gene_ranklist()
    with torch.no_grad():
        rank_val = masked_scores(m.result, U, uidict, rank, 1e-6)
    w = torch.cat((m.user_embedding.weight, m.item_embedding.weight), 0).detach()
    rng = np.random.default_rng(12)
    rows = np.sort(rng.choice(U + I, 512, replace=False))
    urows = np.sort(rng.choice(U, 1024, replace=False))
    np.savez_compressed(
        os.path.join(HERE, f"lightgcn_trajectory_{name}.npz"), D=D, L=L, reg=reg, lr=1e-3, init_seed=42, T=T,
        batches=batches.astype(np.int32), losses=np.array(losses, np.float64), rows=rows,
        weight_rows=sample_rows(w, rows), weight_sum=np.float64(w.double().sum().item()),
        result_rows=sample_rows(m.result, rows), result_sum=np.float64(m.result.double().sum().item()),
        urows=urows, rank_rows=rank.numpy()[urows].astype(np.int32), rank_val_rows=rank_val.numpy()[urows],
        k_list=np.array(K_LIST), metric_names=np.array(NAMES), val_metrics=metrics_table(val, rank),
        test_metrics=metrics_table(test, rank))


def gen_trajectory_baby():
    trajectory("baby", 2, 10)


def gen_trajectory_sports():
    trajectory("sports", 3, 10)


def synthetic_features(num_item, dv, dt, seed=0):
    """the feature blobs are missing; N(0,1) from one torch CPU generator, visual first."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(num_item, dv, generator=g), torch.randn(num_item, dt, generator=g)


def grad_digest(named, big=70000, rows=48):
    """Every gradient in full when small, else its first `rows` rows, plus |.|-sums of all of them."""
    out = {}
    for k, gten in named:
        a = gten.detach().numpy()
        out["gsum_" + k] = np.float64(np.abs(a.astype(np.float64)).sum())
        out["g_" + k] = a.copy() if a.size <= big else a[:rows].copy()
    return out


def gen_freedom_clothing():
    """configs[2]: reference FREEDOM on the real clothing graph with the SURVEY 8(d) feature widths (4096 / 384),
    YAML hyper-parameters (L=2, mm_layers=1, ii_topk=10, dropout 0.1, w=0.8 in the lambda_coeff slot, Q5)."""
    U, I, train, val, test = load_real("clothing")
    uidict = G.uid(train)
    hist = {u: set(v) for u, v in uidict.items()}
    v_feat, t_feat = synthetic_features(I, 4096, 384)
    D, reg, dropout = 64, 1e-3, 0.1
    t0 = time.time()
    torch.manual_seed(0)
    m = G.FREEDOM(U, I, train, uidict, v_feat, t_feat, D, D, reg, dropout, 2, 1, 10, 0.8, G.DEV)
    print(f"  FREEDOM constructed in {time.time() - t0:.1f} s", flush=True)
    drawn = {}
    real_multinomial = torch.multinomial

    def recording(*a, **k):
        drawn["idx"] = real_multinomial(*a, **k)
        return drawn["idx"]

    torch.manual_seed(123)
    torch.multinomial = recording
    try:
        m.pre_epoch_processing()
    finally:
        torch.multinomial = real_multinomial
    keep = np.zeros(len(train), bool)
    keep[drawn["idx"].numpy()] = True
    assert int(keep.sum()) == int(len(train) * (1 - dropout))
    batch = fixed_batches(train, U, I, hist, 1, 1024, 8)[0]
    users, pos, neg = (torch.from_numpy(b) for b in batch)
    loss = m.loss(users, pos, neg)
    loss.backward()
    rank = m.gene_ranklist()
    with torch.no_grad():
        rank_val = masked_scores(m.result, U, uidict, rank, 1e-6)
    mm = m.mm_adj.coalesce()
    mm_rows = np.sort(np.random.default_rng(13).choice(I, 256, replace=False))
    sel = np.isin(mm.indices()[0].numpy(), mm_rows)
    # the whole item-item graph: every row has ii_topk neighbours per modality (self included) and every degree is
    # ii_topk, so an entry takes one of three values (image only / text only / both): columns + a code per entry
    mm_r, mm_c, mm_v = mm.indices()[0].numpy(), mm.indices()[1].numpy(), mm.values().numpy()
    levels = np.unique(mm_v)
    assert len(levels) == 3 and I < 32768, levels
    mm_counts = np.bincount(mm_r, minlength=I)
    assert mm_counts.max() < 256 and np.all(np.diff(mm_r) >= 0)
    masked = m.masked_adj.coalesce()
    rng = np.random.default_rng(14)
    rows = np.sort(rng.choice(U + I, 512, replace=False))
    urows = np.sort(rng.choice(U, 1024, replace=False))
    x0 = torch.cat((m.user_embedding.weight, m.item_embedding.weight), 0).detach()
    g_all = torch.cat((m.user_embedding.weight.grad, m.item_embedding.weight.grad), 0)
    cols_v = np.arange(0, 4096, 16)
    np.savez_compressed(
        os.path.join(HERE, "freedom_clothing.npz"), D=D, reg=reg, dropout=dropout, L=2, mm_layers=1, knn=10, w=0.8,
        init_seed=0, feat_seed=0, dv=4096, dt=384, keep_bits=np.packbits(keep), users=batch[0], pos=batch[1],
        neg=batch[2], rows=rows, x0_rows=sample_rows(x0, rows), x0_sum=np.float64(x0.double().sum().item()),
        image_trs_w_rows=m.image_trs.weight.detach().numpy()[:4, :64].copy(),
        image_trs_w_sum=np.float64(m.image_trs.weight.double().sum().item()),
        text_trs_w_sum=np.float64(m.text_trs.weight.double().sum().item()),
        v_feat_sum=np.float64(v_feat.double().sum().item()), t_feat_sum=np.float64(t_feat.double().sum().item()),
        mm_rows=mm_rows, mm_idx=mm.indices().numpy()[:, sel].astype(np.int32), mm_val=mm.values().numpy()[sel],
        mm_nnz=np.int64(mm.values().numel()), mm_val_sum=np.float64(mm.values().double().sum().item()),
        mm_counts=mm_counts.astype(np.uint8), mm_cols=mm_c.astype(np.int16), mm_levels=levels,
        mm_code=np.searchsorted(levels, mm_v).astype(np.uint8),
        masked_nnz=np.int64(masked.values().numel()), masked_val_sum=np.float64(masked.values().double().sum().item()),
        edge_values_sum=np.float64(m.edge_values.double().sum().item()),
        result_rows=sample_rows(m.result, rows), result_sum=np.float64(m.result.double().sum().item()),
        loss=np.float64(loss.item()), g_rows=sample_rows(g_all, rows),
        g_abs_sum=np.float64(g_all.double().abs().sum().item()),
        g_image_trs_w_cols=m.image_trs.weight.grad.numpy()[:, cols_v].copy(),
        g_image_trs_w_abs_sum=np.float64(m.image_trs.weight.grad.double().abs().sum().item()),
        g_image_trs_b=m.image_trs.bias.grad.numpy().copy(), g_text_trs_w=m.text_trs.weight.grad.numpy().copy(),
        g_text_trs_b=m.text_trs.bias.grad.numpy().copy(),
        g_image_emb_abs_sum=np.float64(m.image_embedding.weight.grad.double().abs().sum().item()),
        g_text_emb_abs_sum=np.float64(m.text_embedding.weight.grad.double().abs().sum().item()),
        g_image_emb_rows=m.image_embedding.weight.grad.numpy()[(pos - U).numpy()[:16]][:, cols_v].copy(),
        g_text_emb_rows=m.text_embedding.weight.grad.numpy()[(pos - U).numpy()[:16]].copy(),
        urows=urows, rank_rows=rank.numpy()[urows].astype(np.int32), rank_val_rows=rank_val.numpy()[urows],
        k_list=np.array(K_LIST), metric_names=np.array(NAMES), val_metrics=metrics_table(val, rank),
        test_metrics=metrics_table(test, rank))


def gen_mmgcn_microlens():
    """configs[3], single-process half: reference MMGCN on the real microlens graph, 128-d visual / 768-d textual
    synthetic features, reg 1e-4, the 'False'-string concat quirk (Q1).  preference / id_embedding / result are random
    NON-parameters created by the constructor (Q2): reproducible from the seed, checked on stored sample rows."""
    U, I, train, val, test = load_real("microlens")
    uidict = G.uid(train)
    hist = {u: set(v) for u, v in uidict.items()}
    v_feat, t_feat = synthetic_features(I, 128, 768)
    torch.manual_seed(0)
    m = G.MMGCN(U, I, train, uidict, v_feat, t_feat, 64, 1e-4, "add", "False", True, G.DEV)
    rng = np.random.default_rng(15)
    rows = np.sort(rng.choice(U + I, 512, replace=False))
    urows = np.sort(rng.choice(U, 1024, replace=False))
    prows = np.sort(rng.choice(U, 64, replace=False))
    init = dict(v_pref_rows=sample_rows(m.v_gcn.preference, prows), t_pref_rows=sample_rows(m.t_gcn.preference, prows),
                id_rows=sample_rows(m.id_embedding, rows),
                v_pref_sum=np.float64(m.v_gcn.preference.double().sum().item()),
                t_pref_sum=np.float64(m.t_gcn.preference.double().sum().item()),
                id_sum=np.float64(m.id_embedding.double().sum().item()))
    psum = {"psum_" + k: np.float64(p.double().sum().item()) for k, p in m.named_parameters()}
    batch = fixed_batches(train, U, I, hist, 1, 1024, 9)[0]
    user_tensor = np.stack([batch[0], batch[0]], 1)
    item_tensor = np.stack([batch[1], batch[2]], 1)
    t0 = time.time()
    loss = m.loss(torch.from_numpy(user_tensor), torch.from_numpy(item_tensor))
    loss.backward()
    print(f"  MMGCN loss+backward in {time.time() - t0:.1f} s", flush=True)
    grads = grad_digest([(k, p.grad) for k, p in m.named_parameters()])
    t0 = time.time()
    rank = m.gene_ranklist()
    print(f"  MMGCN gene_ranklist in {time.time() - t0:.1f} s", flush=True)
    with torch.no_grad():
        rank_val = masked_scores(m.result.detach(), U, uidict, rank, 1e-5)
    np.savez_compressed(
        os.path.join(HERE, "mmgcn_microlens.npz"), dim_x=64, reg=1e-4, init_seed=0, feat_seed=0, dv=128, dt=768,
        user_tensor=user_tensor, item_tensor=item_tensor, rows=rows, prows=prows, loss=np.float64(loss.item()),
        result_rows=sample_rows(m.result, rows), result_sum=np.float64(m.result.double().sum().item()),
        param_names=np.array([k for k, _ in m.named_parameters()]), urows=urows,
        rank_rows=rank.numpy()[urows].astype(np.int32), rank_val_rows=rank_val.numpy()[urows],
        k_list=np.array(K_LIST), metric_names=np.array(NAMES), val_metrics=metrics_table(val, rank),
        test_metrics=metrics_table(test, rank), **init, **psum, **grads)


if __name__ == "__main__":
    which = sys.argv[1:] or ["interactions", "lightgcn_sports", "trajectory_baby", "trajectory_sports",
                             "freedom_clothing", "mmgcn_microlens"]
    for w in which:
        t0 = time.time()
        print("generating", w, flush=True)
        globals()["gen_" + w]()
        print(f"  done in {time.time() - t0:.1f} s", flush=True)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f:40s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB")
