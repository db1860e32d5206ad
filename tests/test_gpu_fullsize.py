"""Full-size checks on the GPU box for the multimodal configs of BASELINE.json (configs[2], configs[3]) and the
entry points.  Graphs are the reference's real interaction files (tests/golden/<ds>_interactions.npz); the feature
blobs do not travel, so features are seeded at the configured widths.  The fused HIP path is compared with a plain-torch restatement of the
reference op sequence (torch.sparse.mm / F.linear on the same parameters) -- fp32, tolerance stated per check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _batch(edges, U, I, B, seed, dev):
    rng = np.random.default_rng(seed)
    b = rng.choice(len(edges), B, replace=False)
    return (torch.from_numpy(edges[b, 0].astype(np.int64)).to(dev), torch.from_numpy(edges[b, 1].astype(np.int64)).to(dev),
            torch.from_numpy(rng.integers(U, U + I, B)).to(dev))


def _real(name):
    from conftest import load_interactions
    d = load_interactions(name)
    return d["U"], d["I"], np.asarray(d["train"])


def test_freedom_clothing_size_vs_torch(dev):
    """configs[2]: FREEDOM on the REAL clothing graph (U=18072, I=11384, 76054 interactions) at the feature widths
    SURVEY 8(d) names (4096 visual / 384 textual, seeded: the feature blobs do not travel), dim 64, L=2, mm_layers=1,
    kNN 10, dropout 0.1, w=0.8, batch 1024 -- against the plain-torch restatement on the same parameters.  (The
    reference-class golden for this configuration is tests/test_gpu_real_data.py::test_freedom_clothing_*.)"""
    from chaorec_amd import graph
    from chaorec_amd.Model import FREEDOM
    from oracle.torch_ref import freedom_reference_loss
    U, I, edges = _real("clothing")
    E = len(edges)
    g = torch.Generator().manual_seed(0)
    v_feat, t_feat = torch.randn(I, 4096, generator=g), torch.randn(I, 384, generator=g)
    torch.manual_seed(1)
    m = FREEDOM(U, I, edges, graph.user_item_dict_from_edges(edges), v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8,
                dev).to(dev)
    # kNN graph: every item has exactly 10 neighbours incl. itself, normalised weights 1/10
    mm = m.mm_adj
    assert mm.n_rows == I and float(mm.val.sum()) > 0
    m.pre_epoch_processing()
    assert m.masked_adj.nnz == 2 * int(E * 0.9)
    users, pos, neg = _batch(edges, U, I, 1024, 0, dev)
    loss = m.loss(users, pos, neg)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    ref_loss, ref_result = freedom_reference_loss(m, users, pos, neg)
    ref_loss.backward()
    assert torch.allclose(m.result, ref_result, rtol=1e-4, atol=1e-6)
    assert float(loss.detach()) == pytest.approx(float(ref_loss.detach()), rel=1e-5)
    for n, p in m.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        # (the trs biases get +c*u for the positive and -c*u for the negative row: analytically zero, pure rounding)
        assert float((grads[n] - p.grad).abs().max()) <= 2e-4 * scale + 1e-10, n
    rank = m.gene_ranklist()
    assert rank.shape == (U, 50) and int(rank.min()) >= U and int(rank.max()) < U + I


@pytest.mark.parametrize("lazy", [False, True])
def test_freedom_clothing_size_training_without_the_dense_feature_gradient(dev, lazy):
    """configs[2] at full widths, TRAINING: ten steps of torch.optim.Adam on the plain-torch restatement of
    Model/FREEDOM.py:164-217 (dense [I, 4096] / [I, 384] feature gradients, exactly what the reference does) against ten
    FusedAdam steps of the product path -- batch rows projected only, feature tables updated by chaorec_adam_lowrank_f32
    (eager, and with lazily updated rows + flush), the captured step of train_and_evaluate included.  Every parameter,
    the 46.6 M-element image table among them, within 2e-6 (the table's entries move by up to 4e-4 in these ten steps)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import FREEDOM
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from oracle.torch_ref import freedom_reference_loss
    U, I, edges = _real("clothing")
    g = torch.Generator().manual_seed(0)
    v_feat, t_feat = torch.randn(I, 4096, generator=g), torch.randn(I, 384, generator=g)
    uid = graph.user_item_dict_from_edges(edges)
    batches = [_batch(edges, U, I, 1024, k, dev) for k in range(10)]
    torch.manual_seed(1)
    ref = FREEDOM(U, I, edges, uid, v_feat, t_feat, 64, 64, 1e-3, 0.0, 2, 1, 10, 0.8, dev).to(dev)
    torch.manual_seed(1)
    m = FREEDOM(U, I, edges, uid, v_feat, t_feat, 64, 64, 1e-3, 0.0, 2, 1, 10, 0.8, dev).to(dev)
    ref.pre_epoch_processing()
    m.pre_epoch_processing()
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for b in batches:
        topt.zero_grad()
        freedom_reference_loss(ref, *b)[0].backward()
        topt.step()
    opt = FusedAdam(m.parameters(), lr=1e-3, lazy_rows=lazy)
    assert len(opt._claimed) == 2
    step = GraphedTrainStep(m, opt, example_batch=batches[0])
    for b in batches:
        step(*b)
    opt.flush()
    torch.cuda.synchronize()
    assert m.image_embedding.weight.grad is None
    want = dict(ref.named_parameters())
    for n, p in m.named_parameters():
        assert float((p - want[n]).abs().max()) <= 2e-6, n
    assert float((m.image_embedding.weight - v_feat.to(dev)).abs().max()) > 1e-4       # the table did train


def test_mmgcn_microlens_size_vs_torch(dev):
    """configs[3] (single-GPU part): MMGCN on the REAL microlens graph (U=46420, I=14079, 210567 interactions), dim 64,
    visual 128-d / textual 768-d seeded features, 2 branches x 4 layers: representation, loss and EVERY gradient against
    the plain-torch restatement (torch.sparse.mm / F.linear / autograd) on the same parameters.  Gradient tolerance
    2e-3 of the tensor's largest entry: 8 leaky-relu layers over 60 k rows flip a few branch decisions at rounding
    between any two fp32 associations (the forward of the wide layers runs as a split-bf16 product here)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MMGCN
    from oracle.torch_ref import mmgcn_reference_forward
    U, I, edges = _real("microlens")
    g = torch.Generator().manual_seed(0)
    v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 768, generator=g)
    torch.manual_seed(2)
    m = MMGCN(U, I, edges, graph.user_item_dict_from_edges(edges), v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
    users, pos, neg = _batch(edges, U, I, 1024, 1, dev)
    loss = m.loss(torch.stack((users, users), 1), torch.stack((pos, neg), 1))
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert len(grads) == 50                      # only the Linear layers train
    m.zero_grad()
    ref = mmgcn_reference_forward(m)
    assert torch.allclose(m.result, ref.detach(), rtol=2e-3, atol=2e-5)
    # Model/MMGCN.py:188-202 in plain torch (the regulariser is over non-parameters: a constant)
    ref_bpr = -torch.log(torch.sigmoid((ref[users] * ref[pos]).sum(1) - (ref[users] * ref[neg]).sum(1))).mean()
    ref_bpr.backward()
    with torch.no_grad():
        ut, it = torch.cat((users, users)), torch.cat((pos, neg))
        reg = (m.id_embedding[ut] ** 2 + m.id_embedding[it] ** 2).mean() + (m.v_gcn.preference ** 2).mean()
    assert float(loss.detach()) == pytest.approx(float(ref_bpr.detach() + m.reg_weight * reg), rel=1e-5)
    for n, p in m.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((grads[n] - p.grad).abs().max()) <= 2e-3 * scale + 1e-10, n
    rank = m.gene_ranklist()
    assert rank.shape == (U, 50)


@pytest.mark.parametrize("model", ["LightGCN", "FREEDOM", "MMGCN", "NGCF", "MGCN", "LayerGCN", "BPR", "VBPR", "FREEDOM-lazy"])
def test_main_entry_point_two_epochs(dev, model, tmp_path, monkeypatch):
    """python -m chaorec_amd.main --Model X --data_path baby --synthetic: grid search, train, evaluate.
    (FREEDOM-lazy: the same with CHAOREC_LAZY_ADAM=1 -- lazily updated feature rows through the per-epoch re-pruning, the
    captured step and the flush at the end of training.)"""
    import logging
    from chaorec_amd import main as cmain, dataload
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(dataload.SYNTHETIC_FEATURE_DIMS, "default", (96, 64))
    if model.endswith("-lazy"):
        model = model[:-5]
        monkeypatch.setenv("CHAOREC_LAZY_ADAM", "1")
    logging.getLogger().handlers.clear()
    best = cmain.main(["--Model", model, "--data_path", "baby", "--synthetic", "--num_epoch", "2"])
    assert set(best.keys()) == {5, 10, 20}
    for k in best:
        assert set(best[k]) == {"precision", "recall", "ndcg", "hit_rate", "map"}
        assert 0.0 <= best[k]["recall"] <= 1.0
    assert (tmp_path / "log" / f"{model}_baby.log").exists()


def test_weighted_sample_beyond_the_multinomial_limit(dev):
    """torch.multinomial stops at 2^24 categories (the reason FREEDOM's pruning cannot grow with the graph,
    SURVEY 8(f).4); the device sampler keeps exactly k of 2^25 + 5 weighted edges, reproducibly, and favours heavy
    edges as the sequential law does."""
    from chaorec_amd import ops
    n = (1 << 25) + 5
    k = int(n * 0.8)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    w = torch.rand(n, device=dev, generator=g) + 0.05
    with pytest.raises(RuntimeError):
        torch.multinomial(w, k)
    keep = ops.weighted_sample_keep(w, k, seed=3, step=1)
    assert int(keep.sum()) == k
    assert torch.equal(keep, ops.weighted_sample_keep(w, k, seed=3, step=1))
    assert not torch.equal(keep, ops.weighted_sample_keep(w, k, seed=3, step=2))
    kept, dropped = w[keep.bool()].mean().item(), w[~keep.bool()].mean().item()
    assert kept > dropped + 0.1          # light edges are the ones that get pruned


def test_ngcf_sports_size_step(dev):
    """NGCF at the sports shape: the dropped-and-renormalised values stay a symmetric normalisation of the kept
    graph (row sums of D^-1/2 (keep*A + I) D^-1/2 against its own degrees), a training step runs, ranking is valid."""
    from chaorec_amd import graph, ops
    from chaorec_amd.Model import NGCF
    from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
    U, I, E = DATASET_SHAPES["sports"]
    edges = synthetic_interactions(U, I, E, seed=42)
    torch.manual_seed(0)
    m = NGCF(U, I, edges, graph.user_item_dict_from_edges(edges), 64, 1e-3, 0.2, 3, "add", dev).to(dev)
    s = m.graph
    val, val_t = ops.edge_dropout_norm(s, 0.2, seed=5, step=9)
    er, col = s.entry_row.long(), s.col.long()
    kept = val > 0
    assert bool(kept[er == col].all())                                   # self loops always survive
    frac = float(kept[er != col].float().mean())
    assert abs(frac - 0.8) < 0.005
    deg = torch.zeros(U + I, device=dev).index_add_(0, col[kept], torch.ones(int(kept.sum()), device=dev))
    want = (deg[col] ** -0.5) * (deg[er] ** -0.5)
    assert torch.allclose(val[kept], want[kept], rtol=1e-6)
    # val_t is val read through the reversed-edge map
    assert torch.equal(val_t, val[s.transpose_entry.long()])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    b = torch.from_numpy(edges[:1024].astype(np.int64))
    neg = torch.randint(U, U + I, (1024,))
    loss = m.loss(b[:, 0], b[:, 1], neg)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters())
    rank = m.gene_ranklist(topk=20)
    assert rank.shape == (U, 20) and int(rank.min()) >= U and int(rank.max()) < U + I


def test_ngcf_captured_step_at_sports_size_equals_eager(dev):
    """Regression: the degree workspace of chaorec_edge_dropout_norm used to be cleared with hipMemsetAsync; inside a
    captured step that memset node did not take effect at this size, the degrees accumulated from replay to replay
    and the captured model trained on an ever fainter graph (the 88-node golden graph did not show it).  Captured and
    eager training on the same mask / batch streams must stay together, and the degrees must stay degrees."""
    from chaorec_amd import graph, dataload
    from chaorec_amd.Model import NGCF
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
    U, I, E = DATASET_SHAPES["sports"]
    edges = synthetic_interactions(U, I, E, seed=42)
    uid = graph.user_item_dict_from_edges(edges)

    def make():
        torch.manual_seed(0)
        mm = NGCF(U, I, edges, uid, 64, 1e-3, 0.2, 3, "add", dev).to(dev)
        return mm, FusedAdam(mm.parameters(), lr=1e-3)

    sampler = dataload.DeviceBatchSampler(U, I, uid, edges, 1024, dev, "NGCF")
    batches = [tuple(t.clone() for t in b) for _, b in zip(range(40), sampler)]
    eager, oe = make()
    cap, oc = make()
    step = GraphedTrainStep(cap, oc, example_batch=batches[0])
    eager._drop_calls.copy_(cap._drop_calls)
    true_max = int(np.bincount(np.concatenate([edges[:, 0], edges[:, 1]])).max()) + 1
    for it, b in enumerate(batches):
        oe.zero_grad()
        le = eager.loss(*b)
        le.backward()
        oe.step()
        lc = step(*b)
        assert float(lc.detach()) == pytest.approx(float(le.detach()), rel=2e-4), it
        assert 1 <= int(cap.graph.deg_ws.min()) and int(cap.graph.deg_ws.max()) <= true_max, it


@pytest.mark.parametrize("name,ds", [("LightGCN", "sports"), ("LayerGCN", "sports"), ("FREEDOM", "clothing"),
                                     ("MGCN", "baby"), ("MMGCN", "baby"), ("BPR", "sports"), ("VBPR", "baby")])
def test_captured_training_equals_eager_at_dataset_size(dev, name, ds, monkeypatch):
    """The step the benchmark times is the CAPTURED one: at dataset size, over 40 different batches, it must produce
    the losses of the eager step (same kernels, same batches).  (A captured-only failure at this size is exactly how
    the memset-node hazard of DESIGN 3.5 showed up.)"""
    from chaorec_amd import graph, dataload
    from chaorec_amd import Model
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
    monkeypatch.setitem(dataload.SYNTHETIC_FEATURE_DIMS, "default", (96, 64))
    monkeypatch.setitem(dataload.SYNTHETIC_FEATURE_DIMS, ds, (96, 64))
    U, I, E = DATASET_SHAPES[ds]
    edges = synthetic_interactions(U, I, E, seed=42)
    uid = graph.user_item_dict_from_edges(edges)
    v_feat, t_feat = dataload.synthetic_features(I, ds)

    def make():
        torch.manual_seed(0)
        if name == "LightGCN":
            m = Model.LightGCN(U, I, edges, uid, 64, 1e-3, 3, "add", dev)
        elif name == "LayerGCN":
            m = Model.LayerGCN(U, I, edges, uid, 64, 1e-3, 3, 0.1, dev)
        elif name == "FREEDOM":
            m = Model.FREEDOM(U, I, edges, uid, v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev)
        elif name == "MGCN":
            m = Model.MGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, 2, "add", 0.2, 0.01, dev)
        elif name == "BPR":
            m = Model.BPRMF(U, I, uid, 64, 1e-3, dev)
        elif name == "VBPR":
            m = Model.VBPR(U, I, uid, v_feat, 64, 64, 1e-3, dev)
        else:
            m = Model.MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev)
        m = m.to(dev)
        if hasattr(m, "pre_epoch_processing"):
            m.pre_epoch_processing()
        return m, FusedAdam(m.parameters(), lr=1e-3)

    sampler = dataload.DeviceBatchSampler(U, I, uid, edges, 1024, dev, name)
    batches = [tuple(t.clone() for t in b) for _, b in zip(range(40), sampler)]
    eager, oe = make()
    cap, oc = make()
    if name == "MMGCN":          # preference / id_embedding are random non-parameters (Q2): share them
        cap.v_gcn.preference, cap.t_gcn.preference = eager.v_gcn.preference.clone(), eager.t_gcn.preference.clone()
        cap.id_embedding = eager.id_embedding.clone()
    step = GraphedTrainStep(cap, oc, example_batch=batches[0])
    for it, b in enumerate(batches):
        oe.zero_grad()
        le = eager.loss(*b)
        le.backward()
        oe.step()
        lc = step(*b)
        assert float(lc.detach()) == pytest.approx(float(le.detach()), rel=5e-4), (name, it)
