"""GPU parity of the FREEDOM and MMGCN model surfaces against goldens produced by the reference's
own classes (tests/golden/gen_golden.py).  FREEDOM goldens are pure reference (torch.sparse.mm, whose
CPU accumulation uses FMA/axpy in an unspecified order -> tolerance 1e-5, not bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, tie_aware_rank_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _csr_dense(csr):
    rp, col, val = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), csr.val.cpu().numpy()
    out = np.zeros((csr.n_rows, csr.n_cols), np.float64)
    for r in range(csr.n_rows):
        for e in range(rp[r], rp[r + 1]):
            out[r, col[e]] += val[e]
    return out


def _coo_dense(idx, val, shape):
    out = np.zeros(shape, np.float64)
    np.add.at(out, (idx[0], idx[1]), val)
    return out


def _make_freedom(g, dev):
    from chaorec_amd.Model import FREEDOM
    from chaorec_amd import graph
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = FREEDOM(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["D"]), float(g["reg"]), float(g["dropout"]),
                int(g["L"]), int(g["mm_layers"]), int(g["knn"]), float(g["w"]), dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(g["x0"][:U]))
        m.item_embedding.weight.copy_(torch.from_numpy(g["x0"][U:]))
        m.image_trs.weight.copy_(torch.from_numpy(g["image_trs_w"]))
        m.image_trs.bias.copy_(torch.from_numpy(g["image_trs_b"]))
        m.text_trs.weight.copy_(torch.from_numpy(g["text_trs_w"]))
        m.text_trs.bias.copy_(torch.from_numpy(g["text_trs_b"]))
    return m.to(dev), U, I


@pytest.mark.parametrize("tag", ["drop", "nodrop"])
def test_freedom_golden(dev, tag):
    g = load_golden(f"freedom_small_{tag}.npz")
    m, U, I = _make_freedom(g, dev)
    N = U + I
    # graph construction (P10)
    assert np.array_equal(m.edge_indices.cpu().numpy(), g["edge_indices"])
    assert np.array_equal(m.edge_values.cpu().numpy(), g["edge_values"])
    assert np.allclose(_csr_dense(m.norm_adj), _coo_dense(g["norm_idx"], g["norm_val"], (N, N)), rtol=0, atol=1e-9)
    assert np.allclose(_csr_dense(m.mm_adj), _coo_dense(g["mm_idx"], g["mm_val"], (I, I)), rtol=1e-6, atol=1e-9)
    # pruning (P11): same kept edges as the reference's multinomial draw -> same masked graph
    if float(g["dropout"]) > 0:
        k = int(g["masked_idx_raw"].shape[1] // 2)
        keep = torch.from_numpy(g["masked_idx_raw"][:, :k].copy())
        keep[1] -= U
        m._set_masked_adj(keep.to(dev))
        m.pre_epoch_processing.__func__  # the public entry point stays argument-free
    else:
        m.pre_epoch_processing()
    assert np.allclose(_csr_dense(m.masked_adj), _coo_dense(g["masked_idx"], g["masked_val"], (N, N)), rtol=1e-6, atol=1e-9)
    # loss + grads (P12, P13)
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert np.allclose(m.result.detach().cpu().numpy(), g["result"], rtol=1e-5, atol=1e-6)
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    for name, p in (("g_user", m.user_embedding.weight), ("g_item", m.item_embedding.weight),
                    ("g_image_trs_w", m.image_trs.weight), ("g_image_trs_b", m.image_trs.bias),
                    ("g_text_trs_w", m.text_trs.weight), ("g_text_trs_b", m.text_trs.bias),
                    ("g_image_emb", m.image_embedding.weight), ("g_text_emb", m.text_embedding.weight)):
        assert np.allclose(p.grad.cpu().numpy(), g[name], rtol=2e-4, atol=1e-8), name
    # ranking (R)
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    res = g["result"]
    sc = res[:U] @ res[U:].T
    from chaorec_amd import graph
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_freedom_pre_epoch_statistics(dev):
    """The public pre_epoch_processing(): keeps int(E*(1-dropout)) distinct edges, symmetric, re-normalised."""
    g = load_golden("freedom_small_drop.npz")
    m, U, I = _make_freedom(g, dev)
    m.pre_epoch_processing()
    A = _csr_dense(m.masked_adj)
    E = g["edges"].shape[0]
    keep = int(E * (1 - float(g["dropout"])))
    assert (A != 0).sum() == 2 * keep
    assert np.allclose(A, A.T)
    ui = A[:U, U:]
    du, di = (ui != 0).sum(1), (ui != 0).sum(0)
    r, c = np.nonzero(ui)
    want = (1e-7 + du[r]).astype(np.float32) ** -0.5 * (1e-7 + di[c]).astype(np.float32) ** -0.5
    assert np.allclose(ui[r, c], want, rtol=1e-5)
    m.pre_epoch_processing()                 # next epoch: another draw of the same size
    B = _csr_dense(m.masked_adj)
    assert (B != 0).sum() == 2 * keep and ((A != 0) != (B != 0)).any()


def test_mmgcn_fused_layer_is_the_composition_bit_for_bit(dev, monkeypatch):
    """ops.mmgcn_layer (one autograd node per layer, CHAOREC_MMGCN_LAYER=fused) against the composition of ops.linear /
    ops.spmm / F.leaky_relu / + / torch.cat it replaces: same GEMM and SpMM kernels, same elementwise arithmetic -> loss,
    representation and every gradient (parameters and id_embedding) bit-identical.  With ops.normalize_rows in place of
    cat + F.normalize as well (a row's squares summed in another order) everything agrees to 2e-5 relative."""
    from chaorec_amd.Model import MMGCN
    from chaorec_amd import graph, ops
    g = load_golden("mmgcn_small.npz")
    U, I = int(g["U"]), int(g["I"])
    names = [str(n) for n in g["param_names"]]

    def run(mode, torch_normalize=False):
        monkeypatch.setattr(ops, "MMGCN_LAYER", mode)
        if torch_normalize:
            monkeypatch.setattr(ops, "normalize_rows", lambda a, b: torch.nn.functional.normalize(torch.cat((a, b), dim=0)))
        torch.manual_seed(0)
        m = MMGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                  torch.from_numpy(g["t_feat"]), int(g["dim_x"]), float(g["reg"]), "add", "False", True, dev)
        m.load_state_dict({n: torch.from_numpy(g["p_" + n]) for n in names})
        m = m.to(dev)
        m.v_gcn.preference = torch.from_numpy(g["v_pref"]).to(dev)
        m.t_gcn.preference = torch.from_numpy(g["t_pref"]).to(dev)
        m.id_embedding = torch.from_numpy(g["id_embedding"]).to(dev).requires_grad_(True)   # (exercises grad_id too)
        loss = m.loss(torch.from_numpy(g["user_tensor"]), torch.from_numpy(g["item_tensor"]))
        loss.backward()
        monkeypatch.undo()
        return m.result.detach(), loss.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}, m.id_embedding.grad.clone()

    ru, lu, gu, idu = run("unfused")
    rf, lf, gf, idf = run("fused", torch_normalize=True)
    assert torch.equal(rf, ru) and torch.equal(lf, lu) and torch.equal(idf, idu)
    for n in names:
        assert torch.equal(gf[n], gu[n]), n
    rn, ln, gn, idn = run("fused")
    for n in names:
        scale = float(gu[n].abs().max()) + 1e-30
        assert float((gn[n] - gu[n]).abs().max()) <= 2e-5 * scale, n
    assert float((ln - lu).abs()) <= 2e-6 * float(lu.abs())
    assert float((idn - idu).abs().max()) <= 2e-5 * float(idu.abs().max())
    assert torch.allclose(rn, ru, rtol=2e-5, atol=1e-7)


def test_normalize_rows_is_f_normalize(dev):
    from chaorec_amd import ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    a = torch.randn(1000, 256, device=dev, generator=gen)
    b = (torch.randn(777, 256, device=dev, generator=gen) * 3).requires_grad_(True)
    a[3] = 0                                     # |x| < eps: y = x / eps = 0, gradient gy / eps
    a[4] = 1e-14
    a = a.requires_grad_(True)
    w = torch.randn(1777, 256, device=dev, generator=gen)
    y = ops.normalize_rows(a, b)
    ref = torch.nn.functional.normalize(torch.cat((a.detach(), b.detach())).double())
    assert torch.allclose(y.double(), ref, rtol=0, atol=2e-7)
    (y * w).sum().backward()
    a64, b64 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    (torch.nn.functional.normalize(torch.cat((a64, b64))) * w.double()).sum().backward()
    for got, want in ((a.grad, a64.grad), (b.grad, b64.grad)):
        err, scale = (got.double() - want).abs().amax(1), want.abs().amax(1)      # per row: the |x| < eps rows are 1e12 x
        assert bool((err <= 1e-5 * scale + 1e-30).all()), float((err / (scale + 1e-30)).max())
    # the gradient of `a` is skipped when nobody asks for it
    a2, b2 = a.detach(), b.detach().clone().requires_grad_(True)
    (ops.normalize_rows(a2, b2) * w).sum().backward()
    assert torch.equal(b2.grad, b.grad)
    # single-source form
    assert torch.equal(ops.normalize_rows(a.detach()), y[:1000].detach())


def test_mmgcn_golden(dev):
    from chaorec_amd.Model import MMGCN
    from chaorec_amd import graph
    g = load_golden("mmgcn_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MMGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
              torch.from_numpy(g["t_feat"]), int(g["dim_x"]), float(g["reg"]), "add", "False", True, dev)
    names = [str(n) for n in g["param_names"]]
    # Q2: only the Linear layers are parameters, in the reference's registration order
    assert [n for n, _ in m.named_parameters()] == names
    sd = {n: torch.from_numpy(g["p_" + n]) for n in names}
    m.load_state_dict(sd)
    m = m.to(dev)
    m.v_gcn.preference = torch.from_numpy(g["v_pref"]).to(dev)
    m.t_gcn.preference = torch.from_numpy(g["t_pref"]).to(dev)
    m.id_embedding = torch.from_numpy(g["id_embedding"]).to(dev)
    assert m.v_gcn.g_layer1.weight.shape == (64, 256 + 64)      # Q1: concat branch
    assert not hasattr(m.t_gcn, "MLP")                           # Q3
    loss = m.loss(torch.from_numpy(g["user_tensor"]), torch.from_numpy(g["item_tensor"]))
    loss.backward()
    assert np.allclose(m.result.detach().cpu().numpy(), g["result"], rtol=2e-4, atol=2e-6)
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-5)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-4 * scale, n
    rank = m.gene_ranklist(step=int(g["step"]), topk=int(g["topk"])).numpy()
    res = g["result"]
    sc = res[:U] @ res[U:].T
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-5
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=2e-4, atol=1e-7)
    assert ok, why


def test_sharded_mmgcn_one_rank_equals_mmgcn_on_the_kernels(dev):
    """dist.ShardedMMGCN (BASELINE configs[3]) with a single shard covering every user, on the HIP kernels, against
    the reference golden of the unsharded model: the sharded propagate (two block SpMMs + diagonal self-loop terms)
    and its partial-gradient backward give the same representation, loss and Linear gradients."""
    from chaorec_amd.Model import MMGCN
    from chaorec_amd import graph, dist as cdist
    g = load_golden("mmgcn_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MMGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
              torch.from_numpy(g["t_feat"]), int(g["dim_x"]), float(g["reg"]), "add", "False", True, dev)
    names = [str(n) for n in g["param_names"]]
    m.load_state_dict({n: torch.from_numpy(g["p_" + n]) for n in names})
    m = m.to(dev)
    m.v_gcn.preference = torch.from_numpy(g["v_pref"]).to(dev)
    m.t_gcn.preference = torch.from_numpy(g["t_pref"]).to(dev)
    m.id_embedding = torch.from_numpy(g["id_embedding"]).to(dev)
    shard = cdist.UserShard(g["edges"], U, I, 1, 0, dev, self_loops=True)
    sm = cdist.ShardedMMGCN(m, shard, dev)
    loss = sm.loss(torch.from_numpy(g["user_tensor"]), torch.from_numpy(g["item_tensor"]))
    loss.backward()
    sm.sync_grads()
    assert np.allclose(sm.result.detach().cpu().numpy(), g["result"], rtol=2e-4, atol=2e-6)
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-5)
    for n, p in sm.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 3e-4 * (np.abs(ref).max() + 1e-12), n
    rank = sm.gene_ranklist(topk=int(g["topk"])).numpy()
    assert rank.shape == (U, int(g["topk"])) and rank.min() >= U


def test_knn_streamed_kdim_bit_exact(dev, oracle):
    """The kNN build over wide modality features (K-dim 384 here, streamed path) vs the oracle."""
    from chaorec_amd import ops
    rng = np.random.default_rng(0)
    n, d, k = 300, 384, 10
    f = rng.standard_normal((n, d)).astype(np.float32)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    want_i, want_v = oracle.score_topk(f, f, None, 0.0, k, 0)
    t = torch.from_numpy(f).to(dev)
    got_i, got_v = ops.score_topk(t, t, None, 0.0, k)
    assert np.array_equal(got_v.cpu().numpy(), want_v)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    assert np.array_equal(got_i[:, 0].cpu().numpy(), np.arange(n))   # self is the nearest neighbour (kept, FREEDOM.py:116)


def test_sparse_mm_adapter_matches_torch_sparse_mm(dev):
    """the `torch.sparse.mm(norm_adj, x)` family.  Adjacency built the way Model/SimGCL.py:64-104
    does (scipy D^-1/2 A D^-1/2 -> torch COO), multiplied by our adapter and by torch, forward and backward."""
    import scipy.sparse as sp
    from chaorec_amd import sparse
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D = 3000, 1700, 20000, 64
    e = synthetic_interactions(U, I, E, seed=9)
    inter = sp.coo_matrix((np.ones(E, np.float32), (e[:, 0], e[:, 1] - U)), shape=(U, I))
    A = sp.bmat([[None, inter], [inter.T, None]], format="csr", dtype=np.float32)
    diag = np.power(np.asarray((A > 0).sum(axis=1)).flatten() + 1e-7, -0.5)
    L = sp.coo_matrix(sp.diags(diag) @ A @ sp.diags(diag))
    adj = torch.sparse_coo_tensor(torch.tensor(np.array([L.row, L.col]), dtype=torch.long),
                                  torch.FloatTensor(L.data), torch.Size(L.shape), dtype=torch.float32).to(dev)
    x = torch.randn(U + I, D, device=dev, requires_grad=True)
    w = torch.randn(U + I, D, device=dev)
    y = sparse.mm(adj, x)
    (y * w).sum().backward()
    gx = x.grad.clone()
    x.grad = None
    yr = torch.sparse.mm(adj, x)
    (yr * w).sum().backward()
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gx, x.grad, rtol=1e-5, atol=1e-6)
    assert getattr(adj, "_chaorec_csr").symmetric           # converted once, cached, recognised as symmetric
    # a non-symmetric matrix goes through the transposed CSR in the backward
    idx = torch.stack([torch.randint(0, 500, (4000,)), torch.randint(0, 300, (4000,))]).to(dev)
    m = torch.sparse_coo_tensor(idx, torch.randn(4000, device=dev), (500, 300))
    x2 = torch.randn(300, 32, device=dev, requires_grad=True)
    y2 = sparse.mm(m, x2)
    y2.square().sum().backward()
    g2 = x2.grad.clone()
    x2.grad = None
    torch.sparse.mm(m, x2).square().sum().backward()
    assert torch.allclose(g2, x2.grad, rtol=1e-4, atol=1e-5)


def test_shared_gene_ranklist_helper(dev, oracle):
    """the shared ranking helper, both mask values."""
    from chaorec_amd import ranking, graph
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D = 500, 3000, 4000, 64
    e = synthetic_interactions(U, I, E, seed=10)
    uid = graph.user_item_dict_from_edges(e)
    hist = ranking.history_csr(uid, U, dev)
    res = (np.random.default_rng(0).standard_normal((U + I, D)) * 0.1).astype(np.float32)
    for mask in (1e-6, 1e-5):
        got = ranking.gene_ranklist(torch.from_numpy(res).to(dev), U, I, hist, mask, 50)
        want, _ = oracle.gene_ranklist(res, U, I, oracle.user_hist_csr(e, U), mask, 50)
        assert got.dtype == torch.int64 and got.device.type == "cpu" and np.array_equal(got.numpy(), want)


def _ngcf_from_golden(g, dev, dropout):
    from chaorec_amd.Model import NGCF
    from chaorec_amd import graph
    U, I = int(g["U"]), int(g["I"])
    m = NGCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), dropout,
             int(g["L"]), "add", dev)
    m.load_state_dict({str(n): torch.from_numpy(g["p_" + str(n)]) for n in g["param_names"]})
    return m.to(dev)


def _edge_masks_to_entries(edges, n_nodes, masks):
    """keep masks over the 2E bidirectional edges (the reference's dropout_adj order) -> uint8 masks over the entries
    of the destination-major CSR with self loops (stable sort by destination, loops appended last and always kept)."""
    from chaorec_amd import graph
    ei = graph.bidirectional_edge_index(edges)
    dst = torch.cat([ei[1], torch.arange(n_nodes)])
    order = torch.argsort(dst, stable=True)
    out = []
    for k in masks:
        looped = torch.cat([torch.from_numpy(k.astype(np.uint8)), torch.ones(n_nodes, dtype=torch.uint8)])
        out.append(looped[order].contiguous())
    return out


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_ngcf_golden(dev, tag):
    """NGCF on the kernels (SpMM + two GEMMs per layer) against the reference model evaluated edge-wise: same
    representation, loss, gradients of every parameter and ranking, to fp32 re-association tolerance.  With dropout the
    masks the reference drew are fed to chaorec_edge_dropout_norm (keep_in)."""
    g = load_golden(f"ngcf_small_{tag}.npz")
    U, I = int(g["U"]), int(g["I"])
    m = _ngcf_from_golden(g, dev, float(g["dropout"]))
    if tag == "drop":
        m.forced_keep = [k.to(dev) for k in _edge_masks_to_entries(g["edges"], U + I, g["keep_masks"])]
    loss = m.loss(torch.from_numpy(g["users"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"]))
    loss.backward()
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 2e-6 * np.abs(g["result"]).max()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-5 * (np.abs(ref).max() + 1e-12), n
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    sc = g["result"][:U] @ g["result"][U:].T
    from chaorec_amd import graph
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_ngcf_device_dropout_trains_under_graph_capture(dev):
    """The default configuration (dropout 0.2, masks from the device generator): a fresh mask per forward, the
    captured step equals the eager step on the same mask stream, and the loss goes down."""
    from chaorec_amd.Model import NGCF
    from chaorec_amd import graph
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = load_golden("ngcf_small_drop.npz")
    U, I = int(g["U"]), int(g["I"])
    batch = tuple(torch.from_numpy(g[k]).to(dev) for k in ("users", "pos", "neg"))

    def make():
        torch.manual_seed(3)
        return NGCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), 16, 1e-3, 0.2, 2, "add", dev).to(dev)

    a = make()
    r1 = a.forward().detach().clone()
    r2 = a.forward().detach().clone()
    assert not torch.equal(r1, r2)                       # a new mask every call
    b = make()
    assert torch.equal(b.forward().detach(), r1)         # same seed, same call index -> same mask

    eager, cap = make(), make()
    oe, oc = FusedAdam(eager.parameters(), lr=1e-2), FusedAdam(cap.parameters(), lr=1e-2)
    step = GraphedTrainStep(cap, oc, example_batch=batch)
    eager._drop_calls.copy_(cap._drop_calls)             # the warm-up forwards advanced the captured model's counter
    first = last = None
    for it in range(30):
        oe.zero_grad()
        le = eager.loss(*batch)
        le.backward()
        oe.step()
        lc = step(*batch)
        assert float(lc.detach()) == pytest.approx(float(le.detach()), rel=1e-5), it
        first = float(le.detach()) if first is None else first
        last = float(le.detach())
    assert last < first


def test_mgcn_golden(dev):
    """MGCN (a member of the torch.sparse.mm family) against the reference model's own output: the
    four graphs it builds, the representation, loss, the gradient of every parameter and the ranking."""
    from chaorec_amd.Model import MGCN
    from chaorec_amd import graph
    g = load_golden("mgcn_small.npz")
    U, I, N = int(g["U"]), int(g["I"]), int(g["U"]) + int(g["I"])
    torch.manual_seed(0)
    m = MGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
             torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), 2, "add", float(g["ssl_temp"]),
             float(g["ssl_alpha"]), dev)
    names = [str(n) for n in g["param_names"]]
    assert [n for n, _ in m.named_parameters()] == names
    for n, p in m.named_parameters():          # same seed, same creation order -> the reference's initial weights
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    m = m.to(dev)
    assert np.array_equal(_csr_dense(m.norm_adj), _coo_dense(g["norm_adj_idx"], g["norm_adj_val"], (N, N)))
    assert np.array_equal(_csr_dense(m.R), _coo_dense(g["R_idx"], g["R_val"], (U, I)))
    for mine, tag in ((m.image_original_adj, "image"), (m.text_original_adj, "text")):
        ref = _coo_dense(g[tag + "_adj_idx"], g[tag + "_adj_val"], (I, I))
        got = _csr_dense(mine)
        assert np.array_equal(got != 0, ref != 0) and np.allclose(got, ref, rtol=1e-5, atol=1e-7), tag
    loss = m.loss(torch.from_numpy(g["users"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"]))
    loss.backward()
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=5e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-12), n
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    sc = g["result"][:U] @ g["result"][U:].T
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_freedom_captured_step_follows_the_per_epoch_pruning(dev):
    """pre_epoch_processing() rewrites the pruned graph IN PLACE (same addresses, refreshed SpMM schedule), so ONE
    captured training step keeps training on the graph of the current epoch: per-step losses and final parameters
    equal those of eager training on the same pruning stream."""
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = load_golden("freedom_small_drop.npz")
    batch = tuple(torch.from_numpy(g[k]).to(dev) for k in ("users", "pos", "neg"))
    eager, _, _ = _make_freedom(g, dev)
    cap, _, _ = _make_freedom(g, dev)
    oe, oc = FusedAdam(eager.parameters(), lr=1e-2), FusedAdam(cap.parameters(), lr=1e-2)
    step, ptrs, graphs = None, None, []
    for epoch in range(4):
        eager.pre_epoch_processing()
        cap.pre_epoch_processing()
        assert torch.equal(eager.masked_adj.col, cap.masked_adj.col)
        graphs.append(cap.masked_adj.col.clone())
        if step is None:
            step = GraphedTrainStep(cap, oc, example_batch=batch)
            ptrs = (cap.masked_adj.rowptr.data_ptr(), cap.masked_adj.col.data_ptr(), cap.masked_adj.val.data_ptr())
        assert ptrs == (cap.masked_adj.rowptr.data_ptr(), cap.masked_adj.col.data_ptr(), cap.masked_adj.val.data_ptr())
        for it in range(3):
            oe.zero_grad()
            le = eager.loss(*batch)
            le.backward()
            oe.step()
            lc = step(*batch)
            assert float(lc.detach()) == pytest.approx(float(le.detach()), rel=1e-5), (epoch, it)
    assert not torch.equal(graphs[0], graphs[1])          # the epochs really trained on different graphs
    # (Adam turns rounding-level gradient differences of near-zero components into +-lr steps: loose bound)
    for (n, a), (_, b) in zip(eager.named_parameters(), cap.named_parameters()):
        assert float((a - b).abs().max()) < 0.03 * float(a.abs().max()) + 1e-3, n


def test_layergcn_golden(dev):
    """LayerGCN (torch.sparse.mm family) against the reference model's own output: same initial weights from the same
    seed, the normalised graph bit for bit, both prunings (fed with the edges the reference drew), loss, gradients,
    the evaluation forward on the unpruned graph and the ranking."""
    from chaorec_amd.Model import LayerGCN
    from chaorec_amd import graph
    g = load_golden("layergcn_small.npz")
    U, I, N = int(g["U"]), int(g["I"]), int(g["U"]) + int(g["I"])
    torch.manual_seed(0)
    m = LayerGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]),
                 int(g["L"]), float(g["dropout"]), dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    m = m.to(dev)
    assert np.array_equal(_csr_dense(m.norm_adj_matrix), _coo_dense(g["norm_idx"], g["norm_val"], (N, N)))
    assert np.array_equal(m.edge_indices.cpu().numpy(), g["edge_indices"])
    assert np.array_equal(m.edge_values.cpu().numpy(), g["edge_values"])
    for ep in range(2):
        raw = g[f"masked_idx_raw{ep}"]
        keep = torch.from_numpy(raw[:, :raw.shape[1] // 2].copy())
        keep[1] -= U
        m._set_masked_adj(keep.to(dev))
        assert np.allclose(_csr_dense(m.masked_adj), _coo_dense(g[f"masked_idx{ep}"], g[f"masked_val{ep}"], (N, N)),
                           rtol=1e-6, atol=1e-9), ep
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-5 * (np.abs(ref).max() + 1e-12), n
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    res = m.result.cpu().numpy()
    assert np.abs(res - g["eval_result"]).max() <= 2e-6 * np.abs(g["eval_result"]).max()
    sc = g["eval_result"][:U] @ g["eval_result"][U:].T
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why
    # the public pruning: alternates degree-sensitive / uniform draws of the same size, in place
    m.pre_epoch_processing()
    a, ptr = m.masked_adj.col.clone(), m.masked_adj.col.data_ptr()
    m.pre_epoch_processing()
    assert m.masked_adj.col.data_ptr() == ptr and not torch.equal(a, m.masked_adj.col)
    assert m.masked_adj.nnz == 2 * int(len(g["edges"]) * (1 - float(g["dropout"])))


def test_in_launch_batches_equal_the_samplers(dev):
    """The training loop's in-launch mode (LightGCN: the fused BPR forward draws from the epoch permutation) sees
    exactly the batches DeviceBatchSampler.__iter__ yields for the same seed: same users, positives, negatives, in the
    same order, including the short last batch; and one epoch of it equals one epoch over the sampler's tensors."""
    from chaorec_amd import graph, dataload
    from chaorec_amd import train_and_evaluate as tae
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 1500, 900, 10300, 1024            # 10 full batches + a tail of 60
    edges = synthetic_interactions(U, I, E, seed=5)
    uid = graph.user_item_dict_from_edges(edges)

    def make():
        torch.manual_seed(1)
        m = LightGCN(U, I, edges, uid, 64, 1e-3, 2, "add", dev).to(dev)
        return m, FusedAdam(m.parameters(), lr=1e-3), dataload.DeviceBatchSampler(U, I, uid, edges, B, dev, "LightGCN", 7)

    ma, oa, la = make()
    want = [tuple(t.clone() for t in b) for b in la]                      # epoch 0 as tensors
    mb, ob, lb = make()
    lb.begin_epoch()
    got = []
    for k in range(len(lb)):
        nb = min(B, E - k * B)
        mb.loss_drawn(lb.edges, nb, lb.seed, 0, step_dev=lb.step_dev, advance=True, perm=lb.perm, perm_pos=lb.perm_pos)
        got.append(tuple(t.clone() for t in mb.batch))
    for k, (w, g_) in enumerate(zip(want, got)):
        assert torch.equal(w[0], g_[0]) and torch.equal(w[1] - U, g_[1]) and torch.equal(w[2] - U, g_[2]), k
    # a whole epoch through train(): captured in-launch steps vs eager steps over the sampler's batches
    mc, oc, lc = make()
    md, od, ld = make()
    graphed = tae._capture_step(md, ld, od, "LightGCN")
    assert getattr(graphed, "draws_in_launch", False)
    ld.gen.manual_seed(7)                                                  # (the capture drew a permutation of its own)
    ld.global_step = 0
    ld.step_dev.zero_()
    loss_c = tae.train(mc, lc, oc, "LightGCN", None)
    loss_d = tae.train(md, ld, od, "LightGCN", graphed)
    assert loss_d == pytest.approx(loss_c, rel=1e-5)
    for (n, a), (_, b2) in zip(mc.named_parameters(), md.named_parameters()):
        assert torch.allclose(a, b2, rtol=1e-4, atol=1e-6), n


@pytest.mark.parametrize("tag", ["drop", "nodrop"])
def test_sharded_freedom_one_rank_equals_freedom(dev, tag):
    """dist.ShardedFREEDOM on ONE rank through the HIP kernels (sharded propagate blocks, keys-only sampler entry + the
    radix select over the histograms, item-item branch with the summed-gradient identity) against the single-process
    FREEDOM on the same weights: the same pruned edge set, representation, loss and gradients."""
    from chaorec_amd import dist as cdist, ops
    g = load_golden(f"freedom_small_{tag}.npz")
    m, U, I = _make_freedom(g, dev)
    sh = cdist.ShardedFREEDOM(m, [0, U], 1, 0, dev)
    m.pre_epoch_processing()
    sh.pre_epoch_processing()
    if m.dropout > 0:
        # the keys-only entry numbers entries by `ids`: a shuffled share of the list gets the whole list's keys
        w = m.edge_values
        _, keys = ops.weighted_sample_keep(w, 5, seed=3, step=2, return_keys=True)
        perm = torch.randperm(w.numel(), device=dev)
        assert torch.equal(ops.weighted_sample_keys(w[perm], perm, seed=3, step=2), keys[perm])
        assert torch.equal(ops.weighted_sample_keys(w, None, seed=3, step=2), keys)
        # k-th smallest by histograms == by sorting
        k = int(w.numel() * 0.7)
        assert cdist.global_kth_smallest(keys, k) == int(torch.sort(keys[keys >= 0]).values[k - 1])
        assert sh.shard.nnz * 2 == m.masked_adj.nnz
    batch = tuple(torch.from_numpy(g[k]).to(dev) for k in ("users", "pos", "neg"))
    loss = m.loss(*batch)
    loss.backward()
    sh.zero_grad()
    ls = sh.loss(batch[0], batch[1] - U, batch[2] - U)
    ls.backward()
    sh.sync_grads()
    assert float(ls.detach()) == pytest.approx(float(loss.detach()), rel=1e-5)
    assert torch.allclose(sh.result, m.result, rtol=1e-5, atol=1e-7)
    for n, p in m.named_parameters():
        q = dict(sh.named_parameters())[n]
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((q.grad - p.grad).abs().max()) <= 2e-4 * scale + 1e-9, n
    assert torch.equal(sh.gene_ranklist(topk=10), m.gene_ranklist(topk=10))


def _rank_check(m, g, U, result_u, result_i):
    from chaorec_amd import graph
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    sc = result_u @ result_i.T
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_bprmf_golden(dev):
    """BPRMF against the reference class: loss with the item bias and the reference's own regulariser
    (negative rows un-squared), every gradient -- the bias's through the padded table column --, and the ranking
    (which ignores the bias, as the reference's does)."""
    from chaorec_amd.Model import BPRMF
    from chaorec_amd import graph
    g = load_golden("bprmf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    m = BPRMF(U, I, graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(g["p_" + n]))
    m = m.to(dev)
    batch = tuple(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))
    loss = m.loss(*batch)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-5 * (np.abs(ref).max() + 1e-12), n
    pos_s, neg_s = m.forward(batch[0], batch[1] - U, batch[2] - U)
    assert pos_s.shape == neg_s.shape == (len(g["users"]),)
    _rank_check(m, g, U, g["p_user_embedding.weight"], g["p_item_embedding.weight"])


def test_vbpr_golden(dev):
    """VBPR against the reference class: the joined representation, loss, every gradient (the trainable
    visual feature table included: dense here, no optimizer has claimed it) and the ranking; then three FusedAdam steps
    with the feature table claimed against three with its dense gradient."""
    from chaorec_amd.Model import VBPR
    from chaorec_amd import graph
    from chaorec_amd.optim import FusedAdam
    g = load_golden("vbpr_small.npz")
    U, I = int(g["U"]), int(g["I"])

    def make():
        m = VBPR(U, I, graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), int(g["D"]), 64,
                 float(g["reg"]), dev)
        assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(torch.from_numpy(g["p_" + n]))
        return m.to(dev)

    m = make()
    batch = tuple(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))
    loss = m.loss(*batch)
    loss.backward()
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=5e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        # (+1e-8: the projection's bias gradient is a sum of +c u and -c u terms that cancel to ~1e-5 -- what is left
        # is the rounding of either implementation)
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-8, n
    _rank_check(m, g, U, g["result"][:U], g["result"][U:])
    out = {}
    for claim in (False, True):
        m = make()
        if not claim:
            del m.v_feat.weight._chaorec_projected_only
        opt = FusedAdam(m.parameters(), lr=1e-3)
        assert len(opt._claimed) == int(claim)
        for _ in range(3):
            opt.zero_grad()
            m.loss(*batch).backward()
            opt.step()
        out[claim] = {k: v.detach().clone() for k, v in m.named_parameters()}
    # (item_linear.bias: its gradient is analytically zero -- +c and -c per sample -- so what arrives is the rounding of
    #  the backward's float atomics, and Adam turns a gradient of pure noise into steps of +-lr: two RUNS of the same code
    #  differ there by up to 2 x 3 lr after three steps (each run walks its own way).  The bias shifts every item's visual part alike; it cancels in
    #  pos - neg and reaches the other parameters only through the 1e-3-weighted regulariser.)
    def close(a, b, atol):
        # two RUNS of the same code: the noise bias reaches the other parameters through the regulariser, and where an entry's
        # own gradient is ~0 Adam turns that too into a step of +-lr -- a handful of entries per run (one full-suite run in
        # eight had one beyond 2e-6): all but a handful (3, or 1e-3 of the entries) within atol, every entry within 2 x steps x lr
        d = (a - b).abs()
        return int((d > atol).sum()) <= max(3, int(1e-3 * d.numel())) and float(d.max()) <= 1e-2

    def same(a, b, k):
        return close(a[k], b[k], 7e-3 if k == "item_linear.bias" else 2e-6)   # (2 x 3 steps x lr + margin)
    for k in out[True]:
        assert same(out[True], out[False], k), k
    # lazy rows must not apply to a table whose forward reads EVERY row (ops.linear): rows outside the batch would be
    # projected and ranked stale.  FusedAdam(lazy_rows=True) has to train this model exactly like lazy_rows=False.
    lazy_out = {}
    for lazy in (False, True):
        m = make()
        opt = FusedAdam(m.parameters(), lr=1e-3, lazy_rows=lazy)
        for _ in range(4):
            opt.zero_grad()
            m.loss(*batch).backward()
            opt.step()
        lazy_out[lazy] = ({k: v.detach().clone() for k, v in m.named_parameters()}, m.result.detach().clone())
        assert "last" not in opt.state[m.v_feat.weight]
    for k in lazy_out[True][0]:       # (atomics order in the BPR backward: last-bit noise between any two runs)
        assert close(lazy_out[True][0][k], lazy_out[False][0][k], 9e-3 if k == "item_linear.bias" else 2e-6), k      # (2 x 4 steps x lr + margin)
    ra, rb = lazy_out[True][1], lazy_out[False][1]
    E = ra.shape[1] - 64                  # (the visual part of the item rows carries the bias: compare it up to that shift)
    assert close(ra[:, :E], rb[:, :E], 2e-6) and close(ra[:U], rb[:U], 2e-6)
    shift = lazy_out[True][0]["item_linear.bias"] - lazy_out[False][0]["item_linear.bias"]
    assert close(ra[U:, E:] - shift, rb[U:, E:], 4e-6)
