"""Five more members of the reference's `torch.sparse.mm` model family through the adapter ALONE (VERDICT
r3 #8): SimGCL, XSimGCL, NCL, SelfCF, SLMRec.  Their product classes (chaorec_amd/Model/{SimGCL,XSimGCL,NCL,SelfCF,SLMRec}.py) swap `torch.sparse.mm` for
`chaorec_amd.sparse.mm` and the per-model ranking loop for the shared `ranking.gene_ranklist` -- no kernel, no fusion was
written for them.  Goldens: the REFERENCE classes' own outputs (tests/golden/gen_sparse_family.py; what that generator had
to supply around them -- stored noise, seeded clusters, dropout switched off -- is listed in its docstring)."""
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
    out = np.zeros((csr.n_rows, csr.n_cols), dtype=np.float32)
    rp, col, val = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), csr.val.cpu().numpy()
    for r in range(csr.n_rows):
        out[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
    return out


def _coo_dense(idx, val, shape):
    out = np.zeros(shape, dtype=np.float32)
    out[idx[0], idx[1]] = val
    return out


def _check_common(m, g, dev, adj, rtol_grad):
    """Same seed -> the reference's initial weights; the adjacency bit for bit; then loss + every gradient."""
    U, I = int(g["U"]), int(g["I"])
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    assert np.array_equal(_csr_dense(adj), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=5e-6)
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= rtol_grad * (np.abs(ref).max() + 1e-12), n


def _check_rank(rank, g, scores, U):
    from chaorec_amd import graph
    sc = scores.copy()
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_simgcl_golden(dev):
    from chaorec_amd import graph
    from chaorec_amd.Model import SimGCL
    g = load_golden("simgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SimGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]),
               float(g["ssl_temp"]), float(g["ssl_reg"]), dev)
    noise = [torch.from_numpy(n).to(dev) for n in g["noise"]]          # what the reference's torch.rand_like drew, in order
    m.noise_fn = lambda x: noise.pop(0)
    m = m.to(dev)
    _check_common(m, g, dev, m.sparse_norm_adj, 2e-5)
    assert not noise
    assert np.abs(m.user_emb.detach().cpu().numpy() - g["user_emb"]).max() <= 2e-6 * np.abs(g["user_emb"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["user_emb"] @ g["item_emb"].T, U)
    # the default noise: uniform [0, 1) on the embeddings' device, a fresh draw per layer and view
    m.noise_fn = torch.rand_like
    a, _ = m.forward(perturbed=True)
    b, _ = m.forward(perturbed=True)
    assert not torch.equal(a, b) and float((a - m.forward()[0]).abs().max()) <= 2 * m.eps


def test_xsimgcl_golden(dev):
    """Model/XSimGCL.py: one perturbed forward (the second view = layer 1's output of the same pass), and a gene_ranklist that
    runs its own clean forward on the current weights."""
    from chaorec_amd import graph
    from chaorec_amd.Model import XSimGCL
    g = load_golden("xsimgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = XSimGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]),
                float(g["ssl_temp"]), float(g["ssl_reg"]), dev)
    noise = [torch.from_numpy(n).to(dev) for n in g["noise"]]          # what the reference's torch.rand_like drew, in order
    m.noise_fn = lambda x: noise.pop(0)
    m = m.to(dev)
    _check_common(m, g, dev, m.sparse_norm_adj, 2e-5)
    assert not noise
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()                # (no noise left: the ranking forward must not ask for any)
    assert np.abs(m.user_emb.cpu().numpy() - g["user_emb"]).max() <= 2e-6 * np.abs(g["user_emb"]).max()
    assert np.abs(m.item_emb.cpu().numpy() - g["item_emb"]).max() <= 2e-6 * np.abs(g["item_emb"]).max()
    _check_rank(rank, g, g["user_emb"] @ g["item_emb"].T, U)
    m.noise_fn = torch.rand_like
    out = m.forward(perturbed=True)
    assert len(out) == 4 and float((out[0] - m.forward()[0]).detach().abs().max()) <= 2 * m.eps


def test_slmrec_golden(dev):
    """Model/SLMRec.py: a multi-modal member (id / visual / textual item tables propagated over one graph, fused by Linears,
    InfoNCE losses) -- parameters in the reference's creation order, the adjacency, loss, every gradient, the ranked table."""
    from chaorec_amd import graph
    from chaorec_amd.Model import SLMRec
    g = load_golden("slmrec_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SLMRec(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
               torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_alpha"]), dev).to(dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    assert np.allclose(_csr_dense(m.norm_adj), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)), rtol=0, atol=0)
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=5e-6)
    unused = set(str(n) for n in g["no_grad"])
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None, n                     # (g_a_iva: created by the reference, never used)
            continue
        ref = g["g_" + n]
        # (+ 5e-8: the bias of one side of an InfoNCE has a gradient that is zero up to rounding -- 2e-9 in the reference's
        #  run, 1e-8 here -- while the weights' gradients are ~1e-2)
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 5e-5 * np.abs(ref).max() + 5e-8, n
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 2e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_ncl_golden(dev):
    from chaorec_amd import graph
    from chaorec_amd.Model import NCL
    g = load_golden("ncl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = NCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]), "add",
            float(g["ssl_temp"]), float(g["ssl_reg"]), dev)
    m.k = int(g["k"])
    m = m.to(dev)
    with pytest.raises(RuntimeError, match="e_step"):
        m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    for name in ("user_centroids", "user_2cluster", "item_centroids", "item_2cluster"):
        setattr(m, name, torch.from_numpy(g[name]).to(dev))
    _check_common(m, g, dev, m.norm_adj_mat, 2e-5)
    assert np.abs(m.restore_user_e.detach().cpu().numpy() - g["restore_user_e"]).max() <= 2e-6 * np.abs(g["restore_user_e"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["restore_user_e"] @ g["restore_item_e"].T, U)
    # e_step's contract (Model/NCL.py:67-95): L2-normalised centroids, every node assigned to its nearest one
    m.e_step()
    for cent, assign, emb in ((m.user_centroids, m.user_2cluster, m.user_embedding.weight),
                              (m.item_centroids, m.item_2cluster, m.item_embedding.weight)):
        assert cent.shape == (m.k, int(g["D"])) and assign.shape == (emb.shape[0],)
        assert torch.allclose(cent.norm(dim=1), torch.ones(m.k, device=dev), atol=1e-5)
        assert int(assign.min()) >= 0 and int(assign.max()) < m.k
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    assert torch.isfinite(loss)


def test_vgcl_golden(dev):
    """Model/VGCL.py: the variational encoder (mean over layers 1..L, log-std projection, the two recorded Gaussian draws), the
    node- and cluster-level contrasts (the fixture's [n, 1] cluster columns), the KL term: loss, every gradient, both noised
    views' source tensors, the ranking; then e_step's contract and the loop's forward -> e_step -> loss order."""
    from chaorec_amd import graph
    from chaorec_amd.Model import VGCL
    g = load_golden("vgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = VGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]),
             float(g["ssl_temp"]), float(g["ssl_alpha"]), dev).to(dev)
    it = iter([torch.from_numpy(n).to(dev) for n in g["noise"]])
    m.noise_fn = lambda x: next(it)
    m.forward()
    with pytest.raises(RuntimeError, match="e_step"):
        m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    m.user_2cluster, m.item_2cluster = torch.from_numpy(g["user_2cluster"]).to(dev), torch.from_numpy(g["item_2cluster"]).to(dev)
    _check_common(m, g, dev, m.adj_matrix, 5e-5)
    for name in ("user_emb", "item_emb", "mean", "std"):
        got = getattr(m, name).detach().cpu().numpy()
        assert np.abs(got - g[name]).max() <= 5e-6 * np.abs(g[name]).max(), name
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["user_emb"] @ g["item_emb"].T, U)
    m.noise_fn = None
    m.forward()
    m.e_step()
    for cent, assign, n, k in ((m.user_centroids, m.user_2cluster, U, min(50, U)), (m.item_centroids, m.item_2cluster, I, min(50, I))):
        assert cent.shape == (k, int(g["D"])) and assign.shape == (n, 1)
        assert torch.allclose(cent.norm(dim=1), torch.ones(k, device=dev), atol=1e-5)
        assert int(assign.min()) >= 0 and int(assign.max()) < k
    assert torch.isfinite(m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))))


def test_ddrec_golden(dev):
    """Model/DDRec.py: the per-layer re-filtered graphs as VALUE arrays over one CSR (the E scores as row dot products, the
    survivors' degrees, the dynamic-values SpMM forward and backward) against the reference's dense [U, I] scores + rebuilt edge
    lists + GCNConv; two steps -- ungated, then gated by the first step's item table --: both losses and tables, every gradient
    of the second, the ranking over the [N, 3 D] table."""
    from chaorec_amd import graph
    from chaorec_amd.Model import DDRec
    g = load_golden("ddrec_small.npz")
    U, I, D = int(g["U"]), int(g["I"]), int(g["D"])
    torch.manual_seed(0)
    m = DDRec(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
              torch.from_numpy(g["t_feat"]), D, D, float(g["reg"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_alpha"]),
              float(g["threshold"]), "add", dev).to(dev)
    for csr, tag in ((m.mm_adj, "mm"), (m.image_adj, "image"), (m.text_adj, "text")):
        want, got = _coo_dense(g[tag + "_idx"], g[tag + "_val"], (I, I)), _csr_dense(csr)
        assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 1e-7, tag
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    loss0 = m.loss(*args)
    assert float(loss0.detach()) == pytest.approx(float(g["loss0"]), rel=1e-5)
    assert np.abs(m.result.detach().cpu().numpy() - g["result0"]).max() <= 1e-5 * np.abs(g["result0"]).max()
    _golden_model_checks(m, g, dev, 1e-4, 1e-7)
    res = m.result.detach().cpu().numpy()
    assert res.shape == (U + I, 3 * D) and np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    # the filter is live: at threshold 0 some, not all, interactions of the gated table survive
    ego = torch.cat((m.user_embedding.weight, m.i_v_embeddings), 0).detach()
    kept = int((m.filter_edges(ego) > 0).sum())
    assert 0 < kept < m.n_edges


def test_dccf_golden(dev):
    """Model/DCCF.py: the two adaptively re-weighted propagates per layer as the dynamic-values SpMM with DIFFERENTIABLE values
    (d weight = <gy[head], x[tail]>, the gradient torch.sparse.mm gives its sparse operand) over one structure, a repeated
    interaction counted twice; intent read-outs on the GEMM: loss, every gradient, the layer-summed tables, the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import DCCF
    g = load_golden("dccf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = DCCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]),
             float(g["ssl_temp"]), float(g["ssl_alpha"]), int(g["K"]), float(g["cen_reg"]), dev).to(dev)
    assert m.n_edges == len(g["edges"]) - 1 and float(m._ew.max()) == 2.0
    _check_common(m, g, dev, m.norm_adj_mat, 1e-4)
    for got, name in ((m.ua_embedding, "ua"), (m.ia_embedding, "ia")):
        assert np.abs(got.detach().cpu().numpy() - g[name]).max() <= 5e-6 * np.abs(g[name]).max(), name
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["ua"] @ g["ia"].T, U)


def test_spmm_values_gradient_of_the_values(dev):
    """ops.spmm_values with values that require a gradient against torch.sparse.mm's own autograd (dense restatement): the
    product, the dense operand's gradient through the transposed values, and every stored value's gradient."""
    from chaorec_amd import graph, sparse
    gen = torch.Generator().manual_seed(5)
    n, nnz, D = 300, 2000, 32
    r, c = torch.randint(0, n, (nnz,), generator=gen), torch.randint(0, n, (nnz,), generator=gen)
    csr = graph.coo_to_csr_coalesced(torch.cat([r, c]), torch.cat([c, r]), torch.ones(2 * nnz), n, n, symmetric=True).to(dev)
    st = sparse._dropout_structure(csr)
    val = torch.rand(csr.nnz, generator=gen).to(dev).requires_grad_()
    x = torch.randn(n, D, generator=gen).to(dev).requires_grad_()
    w = torch.randn(n, D, generator=gen).to(dev)
    y = sparse.mm(sparse.DroppedAdj(st, val, val[st.transpose_entry.long()]), x)
    (y * w).sum().backward()
    rows = st.entry_row.long()
    vd, xd = val.detach().double().requires_grad_(), x.detach().double().requires_grad_()
    dense = torch.zeros(n, n, dtype=torch.float64, device=dev).index_put((rows, st.col.long()), vd)
    yd = dense @ xd
    (yd * w.double()).sum().backward()
    assert torch.allclose(y.detach().double(), yd.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(val.grad.double(), vd.grad, rtol=1e-5, atol=1e-5)


def test_micro_golden(dev):
    """Model/MICRO.py over two steps.  Step 1 builds the item graphs on the device (chunked cosine kNN of the projected features,
    mixed with the raw-feature graphs: entries and values against the reference's tensors) and multiplies with them as
    `sparse.LearnedAdj`: the loss and EVERY gradient, the projections' and feature tables' -- which only the sparse values
    reach -- included.  Step 2 multiplies with the detached graphs: loss, gradients (none for the projections), table, ranking."""
    from chaorec_amd import graph, sparse
    from chaorec_amd.Model import MICRO
    g = load_golden("micro_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MICRO(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
              int(g["D"]), int(g["L"]), float(g["reg"]), int(g["K"]), 1, float(g["ssl_temp"]), float(g["lambda_coeff"]),
              float(g["ssl_alpha"]), "add", dev).to(dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    with pytest.raises(AttributeError):
        m.loss(*args, False)
    loss1 = m.loss(*args, True)
    loss1.backward()
    assert isinstance(m.image_adj, sparse.LearnedAdj)
    for adj, tag in ((m.image_adj, "image"), (m.text_adj, "text")):
        want, got = _coo_dense(g[tag + "_idx"], g[tag + "_val"], (I, I)), _csr_dense(adj.detach())
        assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 2e-6, tag
    assert float(loss1.detach()) == pytest.approx(float(g["loss1"]), rel=1e-5)
    assert len(g["no_grad1"]) == 0
    for n, p in m.named_parameters():
        ref = g["g1_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-8, n
    assert np.abs(m.result.detach().cpu().numpy() - g["result1"]).max() <= 1e-5 * np.abs(g["result1"]).max()
    m.zero_grad(set_to_none=True)
    loss = m.loss(*args, False)
    loss.backward()
    assert isinstance(m.image_adj, graph.CSR)
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    unused = set(str(n) for n in g["no_grad"])
    assert unused == {"image_embedding.weight", "text_embedding.weight", "image_trs.weight", "image_trs.bias", "text_trs.weight", "text_trs.bias"}
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None, n
            continue
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-8, n
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_lattice_golden(dev):
    """Model/LATTICE.py over two steps: the item graph the reference keeps as a dense [I, I] matrix, here a sparse.LearnedAdj over
    the union of the four kNN patterns (entries and values against the dense matrix), its values carrying gradient into the
    projections, the feature tables and the two modality weights; then the detached step; ranking."""
    from chaorec_amd import graph, sparse
    from chaorec_amd.Model import LATTICE
    g = load_golden("lattice_small.npz")
    U, I, D = int(g["U"]), int(g["I"]), int(g["D"])
    torch.manual_seed(0)
    m = LATTICE(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
                D, D, float(g["reg"]), int(g["L"]), int(g["mm_layers"]), int(g["K"]), "add", float(g["lambda_coeff"]), dev).to(dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    with pytest.raises(AttributeError):
        m.loss(*args, False)
    loss1 = m.loss(*args, True)
    loss1.backward()
    assert isinstance(m.item_adj, sparse.LearnedAdj)
    got = _csr_dense(m.item_adj.detach())
    assert np.array_equal(got != 0, g["item_adj"] != 0) and np.abs(got - g["item_adj"]).max() <= 2e-6
    assert float(loss1.detach()) == pytest.approx(float(g["loss1"]), rel=1e-5)
    assert len(g["no_grad1"]) == 0
    for n, p in m.named_parameters():
        ref = g["g1_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-8, n
    assert np.abs(m.result.detach().cpu().numpy() - g["result1"]).max() <= 1e-5 * np.abs(g["result1"]).max()
    m.zero_grad(set_to_none=True)
    loss = m.loss(*args, False)
    loss.backward()
    assert isinstance(m.item_adj, graph.CSR)
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    unused = set(str(n) for n in g["no_grad"])
    assert "modal_weight" in unused and "image_trs.weight" in unused
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None, n
            continue
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-8, n
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_mmssl_golden(dev):
    """Model/MMSSL.py over three batches of the reference loop's order (loss_D, then loss(idx)): the row-normalised [U, I] / [I, U]
    operands, the discriminator loss with its double-backward gradient penalty (the reference run's host draws replayed) and its
    gradients, the generator loss and every gradient at batch 0, the modality graphs rewired on the device from batch 0's top-k
    (batch 2's losses see them), then rewired from the emptied lists -- all-zero operands -- under the ranking's forward."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MMSSL
    g = load_golden("mmssl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MMSSL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
              int(g["D"]), float(g["reg"]), float(g["ssl_alpha"]), float(g["ssl_temp"]), float(g["G_rate"]), int(g["mmlayer"]), dev).to(dev)
    m.dropout.p, m.D.net[3].p, m.D.net[7].p, m.m_topk_rate = 0.0, 0.0, 0.0, float(g["m_topk_rate"])
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    for csr, tag, shape in ((m.ui_graph, "ui", (U, I)), (m.iu_graph, "iu", (I, U))):
        want, got = _coo_dense(g[tag + "_idx"], g[tag + "_val"], shape), _csr_dense(csr)
        assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 6e-8, tag
    uniforms, alphas = iter(g["uniforms"]), iter(g["alphas"])
    m.uniform_fn = lambda shape: torch.from_numpy(next(uniforms))
    m.alpha_fn = lambda n: torch.from_numpy(next(alphas))
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    for idx in range(3):
        m.zero_grad(set_to_none=True)
        loss_D = m.loss_D(*args)
        assert float(loss_D.detach()) == pytest.approx(float(g[f"loss_D{idx}"]), rel=2e-5), idx
        if idx == 0:
            loss_D.backward()
            # (a bias in front of a BatchNorm has an exactly zero gradient: what both runs hold there is rounding of a loss of
            #  ~2 000 -- the tolerance is relative to the discriminator's largest gradient, not to each tensor's own)
            scale = max(np.abs(g["gD_" + n]).max() for n, _ in m.D.named_parameters())
            for n, p in m.D.named_parameters():
                assert np.abs(p.grad.cpu().numpy() - g["gD_" + n]).max() <= 2e-4 * scale, n
            m.zero_grad(set_to_none=True)
        loss = m.loss(*args, idx)
        assert float(loss.detach()) == pytest.approx(float(g[f"loss{idx}"]), rel=1e-5), idx
        if idx == 0:
            loss.backward()
            unused = set(str(n) for n in g["no_grad"])
            for n, p in m.named_parameters():
                if n in unused:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
                    continue
                ref = g["g_" + n]
                assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-8, n
        if idx == 1:
            assert m.image_ui_graph is not None and m.image_ui_graph is not m.ui_graph      # rewired from batch 0's top-k
    assert m.image_ui_graph is None and m.text_iu_graph is None                             # ... and from the emptied lists
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    _check_rank(rank, g, g["ua"] @ g["ia"].T, U)


def test_grcn_golden(dev):
    """Model/GRCN.py: PyG's attention layer (edge-wise softmax over a node's incoming edges), the confidence-weighted pruned edge
    weights and the weighted id propagation as segment softmaxes + value arrays over ONE CSR on the dynamic-values SpMM, the edge
    weights carrying gradient into the content GCNs; the reference run's dropout mask over the LISTED edges replayed (an
    interaction listed twice): loss, every gradient, the [N, dim_E + 2 dim_C] table, the ranking with the 1e-5 mask."""
    from chaorec_amd import graph
    from chaorec_amd.Model import GRCN
    g = load_golden("grcn_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = GRCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
             int(g["D"]), int(g["C"]), float(g["reg"]), float(g["dropout"]), int(g["num_routing"]), "add", dev).to(dev)
    assert m.n_listed == len(g["edges"]) == len(g["keep"]) and m.n_edges == m.n_listed - 1 and 0 < int((~g["keep"]).sum())
    m.edge_keep_fn = lambda n, p: torch.from_numpy(g["keep"])
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    user_tensor = torch.from_numpy(np.stack((g["users"], g["users"]), 1))
    item_tensor = torch.from_numpy(np.stack((g["pos"], g["neg"]), 1))
    loss = m.loss(user_tensor, item_tensor)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    assert len(g["no_grad"]) == 0
    for n, p in m.named_parameters():
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-8, n
    res = m.result.detach().cpu().numpy()
    assert res.shape == (U + I, int(g["D"]) + 2 * int(g["C"])) and np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    from chaorec_amd import graph as _g
    sc = (g["result"][:U] @ g["result"][U:].T).copy()
    for u, items in _g.user_item_dict_from_edges(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-5
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(sc, rank - U, 1), g["rank"], np.take_along_axis(sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why
    m.edge_keep_fn = None                                   # the default draws on the device: two losses differ
    assert m.loss(user_tensor, item_tensor).item() != m.loss(user_tensor, item_tensor).item()


def test_mgat_golden(dev):
    """Model/MGAT.py: three gated-attention layers per modality (asymmetric logits, source-degree gate, softmax per target) as
    segment softmaxes + differentiable value arrays over one CSR; an interaction listed twice: loss, every gradient, the
    [N, 3 dim_E] table, the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MGAT
    g = load_golden("mgat_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MGAT(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
             int(g["D"]), float(g["reg"]), dev).to(dev)
    assert m.n_edges == len(g["edges"]) - 1
    _golden_model_checks(m, g, dev, 2e-4, 1e-8)
    res = m.result.detach().cpu().numpy()
    assert res.shape == (U + I, 3 * int(g["D"])) and np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


@pytest.mark.parametrize("D", [4, 16, 64, 100, 256, 384])
def test_edge_dot_kernel_and_its_gradients(dev, D):
    """chaorec_edge_dot_f32 (edge scores over a CSR's entries: one pass instead of two [nnz, D] gathers + product + row sum)
    against the gathers in fp64, over all entries and over the first half of a bipartite structure (what the attention models
    use), with both tables' gradients (two dynamic-values SpMMs) against torch's own."""
    from chaorec_amd import graph, ops, sparse
    gen = torch.Generator().manual_seed(11 + D)
    U, I, nnz = 150, 90, 1500
    key = torch.unique(torch.randint(0, U * I, (nnz,), generator=gen))
    eu, ei = torch.div(key, I, rounding_mode="floor"), key % I
    n = key.numel()
    csr = graph.coo_to_csr_coalesced(torch.cat([eu, U + ei]), torch.cat([U + ei, eu]), torch.ones(2 * n), U + I, U + I, symmetric=True).to(dev)
    st = sparse._dropout_structure(csr)
    a = torch.randn(U + I, D, generator=gen).to(dev).requires_grad_()
    b = torch.randn(U + I, D, generator=gen).to(dev).requires_grad_()
    w = torch.randn(n, generator=gen).to(dev)
    rows, cols = st.entry_row.long(), st.col.long()
    full = ops.edge_dot_raw(st.entry_row, st.col, a.detach(), b.detach())
    want = (a.detach().double()[rows] * b.detach().double()[cols]).sum(1)
    assert torch.allclose(full.double(), want, rtol=1e-5, atol=1e-5)
    assert torch.equal(rows[:n].cpu(), eu) and torch.equal(cols[:n].cpu(), U + ei)         # the first half = the (user, item) pairs in key order
    out = ops.edge_dot(st, a, b, n)
    (out * w).sum().backward()
    ad, bd = a.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    ((ad[rows[:n]] * bd[cols[:n]]).sum(1) * w.double()).sum().backward()
    assert torch.allclose(out.detach().double(), want[:n], rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.grad.double(), ad.grad, rtol=1e-5, atol=1e-5) and torch.allclose(b.grad.double(), bd.grad, rtol=1e-5, atol=1e-5)
    x = torch.randn(U + I, D, generator=gen).to(dev).requires_grad_()                      # the same table on both sides: the gradients add
    ops.edge_dot(st, x, x, n).sum().backward()
    xd = x.detach().double().requires_grad_()
    (xd[rows[:n]] * xd[cols[:n]]).sum().backward()
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-5, atol=1e-5)


def test_learned_adj_gradients(dev):
    """sparse.LearnedAdj (a non-symmetric pattern, values with gradient) against a dense restatement in fp64: product, the
    dense operand's gradient through the transposed layout, every value's gradient; detach() is the same constant matrix."""
    from chaorec_amd import sparse
    gen = torch.Generator().manual_seed(6)
    n, m_, nnz, D = 200, 260, 3000, 32
    key = torch.unique(torch.randint(0, n * m_, (nnz,), generator=gen))
    rows, cols = torch.div(key, m_, rounding_mode="floor"), key % m_
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
    val = torch.rand(key.numel(), generator=gen).to(dev).requires_grad_()
    x = torch.randn(m_, D, generator=gen).to(dev).requires_grad_()
    w = torch.randn(n, D, generator=gen).to(dev)
    adj = sparse.LearnedAdj(rowptr.to(dev), cols.to(dev), val, n, m_)
    y = sparse.mm(adj, x)
    (y * w).sum().backward()
    vd, xd = val.detach().double().requires_grad_(), x.detach().double().requires_grad_()
    dense = torch.zeros(n, m_, dtype=torch.float64, device=dev).index_put((rows.to(dev), cols.to(dev)), vd)
    yd = dense @ xd
    (yd * w.double()).sum().backward()
    assert torch.allclose(y.detach().double(), yd.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(val.grad.double(), vd.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(sparse.mm(adj.detach(), x.detach()), y.detach(), rtol=1e-6, atol=1e-6)


def test_mentor_golden(dev):
    """Model/MENTOR.py: seven two-hop encoders (14 propagates through sparse.mm over ONE CSR = Base_gcn's remove-self-loops +
    degree normalisation + scatter-add), six item-graph products, the MLPs on the GEMM; the reference run's eight perturbation
    draws and two dropout masks replayed: loss, every gradient (41 parameters; none for the no_grad mask branch's Linear and
    the unused feature tables), the fused table, the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MENTOR
    g = load_golden("mentor_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    np.random.seed(0)
    m = MENTOR(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
               int(g["D"]), 1, float(g["reg"]), float(g["ssl_temp"]), float(g["dropout"]), float(g["align_weight"]),
               float(g["mask_weight_g"]), float(g["mask_weight_f"]), dev).to(dev)
    want, got = _coo_dense(g["mm_idx"], g["mm_val"], (I, I)), _csr_dense(m.mm_adj)
    assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 1e-7
    it = iter([torch.from_numpy(n).to(dev) for n in g["noise"]])
    m.noise_fn = lambda x: next(it)
    masks = iter([torch.from_numpy(g["mask_u"]).to(dev), torch.from_numpy(g["mask_i"]).to(dev)])
    m.dropout_fn = lambda x, p: x * next(masks) / (1 - p)
    assert set(str(n) for n in g["no_grad"]) == {"mlp.weight", "mlp.bias", "image_embedding.weight", "text_embedding.weight"}
    _golden_model_checks(m, g, dev, 2e-4, 1e-7)
    # (the reference's forward re-registers the three plain encoders' preferences on the model itself, :162-164: same here)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names_after"]]
    res = m.result_embed.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    m.noise_fn = m.dropout_fn = None                          # the default draws on the device: two losses differ
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    assert m.loss(*args).item() != m.loss(*args).item()


def test_hccf_golden(dev):
    """Model/HCCF.py at keepRate 1 (deterministic): loss, gradients, the layer-summed table, the ranking; then keepRate < 1: the
    propagate runs over sparse_dropout's value array (kept share ~ keepRate, kept values scaled by 1 / keepRate)."""
    from chaorec_amd import graph, sparse
    from chaorec_amd.Model import HCCF
    g = load_golden("hccf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = HCCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]), "add",
             float(g["ssl_alpha"]), float(g["ssl_temp"]), 1.0, 0.5, float(g["mult"]), dev).to(dev)
    _check_common(m, g, dev, m.adj, 5e-5)
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    m.keepRate = 0.5
    a = m.sp_adj_drop_edge()
    assert isinstance(a, sparse.DroppedAdj)
    kept = a.val != 0
    assert 0.35 < float(kept.float().mean()) < 0.65
    assert torch.allclose(a.val[kept], m.adj.val[kept] / 0.5)
    assert torch.isfinite(m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))))


def test_lightgcl_golden(dev):
    """Model/LightGCL.py: the [U, I] matrix (a repeated interaction counted twice) bit for bit, the step with the reference
    run's SVD factors -- loss, gradients, layer-summed tables, ranking --, and this construction's own randomised SVD through
    the HIP SpMM against the exact top-q triplets of the dense matrix."""
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGCL
    g = load_golden("lightgcl_small.npz")
    U, I, q = int(g["U"]), int(g["I"]), 5
    torch.manual_seed(0)
    m = LightGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]), "add",
                 float(g["ssl_alpha"]), float(g["ssl_temp"]), dev).to(dev)
    dense = _coo_dense(g["adj_idx"], g["adj_val"], (U, I))
    assert np.array_equal(_csr_dense(m.adj_norm) != 0, dense != 0) and np.abs(_csr_dense(m.adj_norm) - dense).max() <= 6e-8
    # own factors: the rank-q product is as close to A as the exact truncated SVD's (two power iterations: within a few %)
    s_exact = np.linalg.svd(dense.astype(np.float64), compute_uv=False)
    own = (m.u_mul_s @ m.vt).cpu().numpy().astype(np.float64)
    best = np.sqrt((s_exact[q:] ** 2).sum())
    assert np.linalg.norm(dense - own) <= 1.05 * best
    assert torch.allclose(m.ut @ m.ut.T, torch.eye(q, device=dev), atol=1e-4) and torch.allclose(m.vt @ m.vt.T, torch.eye(q, device=dev), atol=1e-4)
    ref = (g["u_mul_s"] @ g["vt"]).astype(np.float64)
    assert np.linalg.norm(dense - ref) <= 1.05 * best                      # (the reference run's factors: the same quality)
    for name in ("u_mul_s", "v_mul_s", "ut", "vt"):
        setattr(m, name, torch.from_numpy(g[name]).to(dev))
    _golden_model_checks(m, g, dev, 1e-4, 1e-8)
    for got, name in ((m.E_u, "E_u"), (m.E_i, "E_i")):
        assert np.abs(got.detach().cpu().numpy() - g[name]).max() <= 5e-6 * np.abs(g[name]).max(), name
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["E_u"] @ g["E_i"].T, U)


def test_sgl_golden(dev):
    """Model/SGL.py: the two edge-dropped views of a step as value arrays over one CSR (the reference run's recorded draws over
    the LISTED edges, a repeated interaction's copies adding up) against its rebuilt scipy Laplacians: loss, gradients, the
    main view's tables, the ranking; then the device draw keeps the right number of listed edges."""
    from chaorec_amd import graph
    from chaorec_amd.Model import SGL
    g = load_golden("sgl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SGL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]), "add",
            float(g["ssl_temp"]), float(g["ssl_reg"]), dev).to(dev)
    assert m.n_listed == len(g["edges"]) and m.n_edges == m.n_listed - 1
    keeps = iter([g["keep1"], g["keep2"]])

    def edge_keep(n, ratio):
        k = torch.zeros(n, dtype=torch.bool)
        k[torch.from_numpy(next(keeps))] = True
        return k

    m.edge_keep_fn = edge_keep
    want, got = _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)), _csr_dense(m.norm_adj)
    assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 2e-7
    _golden_model_checks(m, g, dev, 1e-4, 1e-8)
    for got, name in ((m.user_emb_final, "user_emb"), (m.item_emb_final, "item_emb")):
        assert np.abs(got.detach().cpu().numpy() - g[name]).max() <= 5e-6 * np.abs(g[name]).max(), name
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["user_emb"] @ g["item_emb"].T, U)
    m.edge_keep_fn = None
    a = m.create_adj_mat(is_subgraph=True, aug_type='ed')
    kept_pairs = int((a.val[:m.n_edges] != 0).sum())
    assert int(m.n_listed * 0.9) - 1 <= kept_pairs <= int(m.n_listed * 0.9)
    for aug in ("nd", "rw"):
        m.ssl_aug_type = aug
        assert torch.isfinite(m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))))


def test_bm3_golden(dev):
    """Model/BM3.py: LightGCN's propagate (GCNConv) as the hot path's layer-mean launch, the predictor and the [I, F] projections on
    the GEMM, the reference run's four dropout masks replayed: loss, every gradient, the table, the ranking over the PREDICTED
    tables."""
    from chaorec_amd import graph
    from chaorec_amd.Model import BM3
    g = load_golden("bm3_small.npz")
    U, I, D = int(g["U"]), int(g["I"]), int(g["D"])
    torch.manual_seed(0)
    m = BM3(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
            D, D, float(g["reg"]), float(g["dropout"]), int(g["L"]), float(g["cl_weight"]), "add", dev).to(dev)
    masks = iter([torch.from_numpy(g[k]).to(dev) for k in ("mask_u", "mask_i", "mask_t", "mask_v")])
    m.dropout_fn = lambda x, p: x * next(masks) / (1 - p)
    _golden_model_checks(m, g, dev, 1e-4, 1e-8)
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["scores"], U)


def test_mgcl_golden(dev):
    """Model/MGCL.py: three LightGCN encoders over one graph (layer-mean launches), three fused BPR terms, the cross-entropy
    contrast: loss, every gradient (none for the unused 0-dim lambda_m), the id table, the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MGCL
    g = load_golden("mgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]),
             int(g["D"]), float(g["reg"]), int(g["L"]), "add", float(g["ssl_temp"]), float(g["ssl_alpha"]), dev).to(dev)
    assert [str(n) for n in g["no_grad"]] == ["lambda_m"]
    _golden_model_checks(m, g, dev, 1e-4, 1e-8)
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_captured_step_draws_fresh_noise_per_replay(dev):
    """SimGCL's rand_like perturbation inside a CAPTURED step (train_and_evaluate captures SimGCL / XSimGCL / SLMRec): the graph
    advances the device generator per replay.  With a zero learning rate and the same batch every replay computes the same
    function of fresh draws: consecutive losses differ; with the perturbation radius at 0 they are identical."""
    from chaorec_amd import graph
    from chaorec_amd.Model import SimGCL
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = load_golden("simgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SimGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), 16, 1e-3, 2, 0.2, 0.1, dev).to(dev)
    batch = [torch.from_numpy(g[k]).to(dev) for k in ("users", "pos", "neg")]
    step = GraphedTrainStep(m, FusedAdam(m.parameters(), lr=0.0), example_batch=batch)
    losses = [float(step(*batch).item()) for _ in range(4)]
    assert len(set(losses)) == 4, losses
    m.eps = 0.0
    step = GraphedTrainStep(m, FusedAdam(m.parameters(), lr=0.0), example_batch=batch)
    losses = [float(step(*batch).item()) for _ in range(3)]
    assert len(set(losses)) == 1, losses


def test_selfcf_golden(dev):
    from chaorec_amd import graph
    from chaorec_amd.Model import SelfCF
    g = load_golden("selfcf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SelfCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]),
               float(g["dropout"]), dev)
    m.online_encoder.drop_flag = False               # (the golden's deterministic configuration)
    m = m.to(dev)
    _check_common(m, g, dev, m.online_encoder.sparse_norm_adj, 2e-5)
    emb = [t.cpu().numpy() for t in m.get_embedding()]
    for got, name in zip(emb, ("u_online", "u_target", "i_online", "i_target")):
        assert np.abs(got - g[name]).max() <= 5e-6 * np.abs(g[name]).max(), name
    # the ranking: ONE inner product over [u_online | u_target] . [i_target | i_online] = the reference's two-matrix sum
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["scores"], U)


def test_sparse_dropout_is_the_reference_helper(dev):
    """sparse.sparse_dropout (Model/SelfCF.py:101-112): with a given keep mask, mm() of the result equals torch.sparse.mm of
    the explicitly rebuilt COO tensor -- forward and the gradient (which needs the masked TRANSPOSE) --; with its own draw
    the kept share is 1 - rate and the kept values are scaled by 1 / (1 - rate)."""
    from chaorec_amd import graph, sparse
    g = load_golden("selfcf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    e = torch.from_numpy(g["edges"]).long()
    csr = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - U, U, I).to(dev)
    gen = torch.Generator().manual_seed(3)
    keep = torch.rand(csr.nnz, generator=gen) < 0.6
    rate = 0.4
    x = (torch.rand(U + I, 16, generator=gen) - 0.5).to(dev).requires_grad_(True)
    y = sparse.mm(sparse.sparse_dropout(csr, rate, keep=keep.to(dev)), x)
    w = (torch.rand(U + I, 16, generator=gen) - 0.5).to(dev)
    (y * w).sum().backward()
    rows = torch.repeat_interleave(torch.arange(U + I), (csr.rowptr[1:] - csr.rowptr[:-1]).cpu())
    coo = torch.sparse_coo_tensor(torch.stack([rows[keep], csr.col.cpu().long()[keep]]), csr.val.cpu()[keep], (U + I, U + I)) \
        * (1.0 / (1 - rate))
    xr = x.detach().cpu().clone().requires_grad_(True)
    yr = torch.sparse.mm(coo, xr)
    (yr * w.cpu()).sum().backward()
    assert torch.allclose(y.detach().cpu(), yr.detach(), rtol=0, atol=2e-6)
    assert torch.allclose(x.grad.cpu(), xr.grad, rtol=0, atol=2e-6)
    d = sparse.sparse_dropout(csr, 0.25)
    kept = d.val != 0
    assert abs(float(kept.float().mean()) - 0.75) < 0.08
    assert torch.allclose(d.val[kept], csr.val[kept] * (1.0 / 0.75))
    # one training step of SelfCF with its random dropouts on: finite loss, every parameter receives a gradient
    from chaorec_amd.Model import SelfCF
    torch.manual_seed(0)
    m = SelfCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), 16, 1e-3, 2, 0.5, dev).to(dev)
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_mcln_golden(dev):
    """Model/MCLN.py -- the reader of the sampler's second negative (dataload.py:81-84): parameters in the reference's
    creation order with its initial weights, the adjacency bit for bit, loss(users, pos, neg, int_items), every gradient, the
    [B, B] score matrix of forward(), and the ranking of user . item + user_v . visual + user_t . textual (one 3 D-wide dot
    product here) against the reference class's output."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MCLN
    g = load_golden("mcln_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MCLN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
             torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), int(g["L"]), int(g["n_mca"]), dev).to(dev)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    assert np.array_equal(_csr_dense(m.norm_adj_mat), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))
    batch = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg", "ints")]
    loss = m.loss(*batch)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    unused = set(str(n) for n in g["no_grad"])
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None, n
            continue
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-7, n
    assert np.abs(m.ua_embeddings.detach().cpu().numpy() - g["ua"]).max() <= 2e-6 * np.abs(g["ua"]).max()
    assert np.abs(m.ia_embeddings.detach().cpu().numpy() - g["ia"]).max() <= 2e-6 * np.abs(g["ia"]).max()
    with torch.no_grad():
        total = m.forward(batch[0].to(dev), *((b - U).to(dev) for b in batch[1:])).cpu().numpy()
    assert np.abs(total - g["total_scores"]).max() <= 2e-5 * np.abs(g["total_scores"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["scores"], U)


# ---- round 5: three more dependency-free members -----------------------------------------------------------
def _golden_model_checks(m, g, dev, rtol_grad, atol_grad=0.0, skip_params=()):
    """parameters (same seed -> the reference's weights, the reference's names), loss, every gradient"""
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    unused = set(str(n) for n in g["no_grad"])
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= rtol_grad * np.abs(ref).max() + atol_grad, n
    return loss


def test_powerec_golden(dev):
    """Model/POWERec.py: three prompt-tuned LayerGCN branches through sparse.mm + the fused cosine re-weighting, the branch
    Linears on the MFMA GEMM, the weak-modality negative; ranking over the concatenated tables of a fresh forward."""
    from chaorec_amd import graph
    from chaorec_amd.Model import POWERec
    g = load_golden("powerec_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = POWERec(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), 2, int(g["prompt_num"]), float(g["neg_weight"]), 0.0,
                dev).to(dev)
    m.pre_epoch_processing()
    assert np.array_equal(_csr_dense(m.norm_adj_matrix), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))
    _golden_model_checks(m, g, dev, 5e-5, 5e-8)
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    res = m.result.cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(rank, g, g["result"][:U] @ g["result"][U:].T, U)


def test_lgmrec_golden(dev):
    """Model/LGMRec.py: id propagate (fused layer mean), modality propagations and user <- item aggregations through sparse.mm,
    hypergraph memberships with the reference run's recorded Gumbel draws and dropout masks (gumbel_fn / drop_fn)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import LGMRec
    g = load_golden("lgmrec_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = LGMRec(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
               torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_alpha"]), dev).to(dev)
    assert np.array_equal(_csr_dense(m.norm_adj), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))

    def replay(prefix, n):
        it = iter([torch.from_numpy(g[f"{prefix}{k}"]).to(dev) for k in range(n)])
        return lambda x: next(it)

    m.train()
    m.gumbel_fn, m.drop_fn = replay("train_gumbel", 4), replay("train_drop", 4)
    _golden_model_checks(m, g, dev, 1e-4, 1e-7)
    m.eval()
    m.gumbel_fn, m.drop_fn = replay("eval_gumbel", 4), None
    rank = m.gene_ranklist(topk=int(g["topk"])).numpy()
    res = m.result.cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(rank, g, g["result"][:U] @ g["result"][U:].T, U)
    # and the hooks' defaults draw on the device: two forwards differ, memberships are rows of a simplex
    m.gumbel_fn = None
    a, b = m.forward()[0], m.forward()[0]
    assert float((a - b).detach().abs().max()) > 0


def test_dhcf_golden(dev):
    """Model/DHCF.py: the two-hop hypergraph operator as eight SpMMs per side and layer against the reference's MATERIALISED
    [H | H H^T H] chain (torch.linalg.multi_dot): the layers' weights + biases (a plain list there: not model parameters;
    the bias uninitialised memory) are copied in from the reference run."""
    from chaorec_amd import graph
    from chaorec_amd.Model import DHCF
    g = load_golden("dhcf_small.npz")
    U, I, L = int(g["U"]), int(g["I"]), int(g["L"])
    torch.manual_seed(0)
    m = DHCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), L, 0.0, dev).to(dev)
    assert len(list(m.parameters())) == 2                      # the list of layers is invisible to the optimizer, as there
    for k, layer in enumerate(m.layers):
        assert np.array_equal(layer.weight.detach().cpu().numpy(), g[f"layer{k}_weight"])     # (same seed, same order of draws)
        with torch.no_grad():
            layer.bias.copy_(torch.from_numpy(g[f"layer{k}_bias"]))
    _golden_model_checks(m, g, dev, 5e-5, 1e-8)
    for k, layer in enumerate(m.layers):
        for nme, p in (("weight", layer.weight), ("bias", layer.bias)):
            ref = g[f"g_layer{k}_{nme}"]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 5e-5 * np.abs(ref).max() + 1e-8, (k, nme)
    res = torch.cat([m.user_e, m.item_e], 0).detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_smore_golden(dev):
    """Model/SMORE.py: the weighted user-item graph and its R block, the two cosine-kNN item graphs and their max-pooled union
    (built on the device here), then loss, every gradient (the trainable [I, F] feature tables included), the tables of the
    last forward and the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import SMORE
    g = load_golden("smore_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = SMORE(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
              torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), int(g["L"]), int(g["K"]), 0.0, "none", dev).to(dev)
    for csr, tag, shape in ((m.norm_adj, "norm", (U + I, U + I)), (m.R, "R", (U, I)), (m.image_original_adj, "image", (I, I)),
                            (m.text_original_adj, "text", (I, I)), (m.fusion_adj, "fusion", (I, I))):
        want = _coo_dense(g[tag + "_idx"], g[tag + "_val"], shape)
        got = _csr_dense(csr)
        assert np.array_equal(got != 0, want != 0), tag               # the same entries ...
        assert np.abs(got - want).max() <= 2e-6, tag                  # ... (cosines of a device matmul: fp32 rounding)
    _golden_model_checks(m, g, dev, 1e-4, 1e-7)
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)


def test_gume_golden(dev):
    """Model/GUME.py: the two cosine-kNN item graphs, their modality intersection, the ENHANCED (not symmetric) user-item graph
    and its R block -- all built on the device here, against the reference's Python loops / scipy --, then loss (the four
    recorded perturbation draws replayed), every gradient, the table of the last forward and the ranking."""
    from chaorec_amd import graph
    from chaorec_amd.Model import GUME
    g = load_golden("gume_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = GUME(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
             torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["L"]), int(g["L_ui"]), float(g["um_loss"]), float(g["vt_loss"]),
             "none", dev).to(dev)
    assert np.array_equal(np.array(sorted(map(tuple, m.inter.t().tolist())), dtype=np.int64), g["inter"])
    for csr, tag, shape, tol in ((m.norm_adj, "norm", (U + I, U + I), 2e-7), (m.R, "R", (U, I), 2e-7),
                                 (m.image_original_adj, "image", (I, I), 2e-6), (m.text_original_adj, "text", (I, I), 2e-6)):
        want = _coo_dense(g[tag + "_idx"], g[tag + "_val"], shape)
        got = _csr_dense(csr)
        assert np.array_equal(got != 0, want != 0), tag
        assert np.abs(got - want).max() <= tol, tag
    assert not m.norm_adj.symmetric
    it = iter([torch.from_numpy(n).to(dev) for n in g["noise"]])
    m.noise_fn = lambda x: next(it)
    _golden_model_checks(m, g, dev, 1e-4, 1e-7)
    res = m.result.detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 1e-5 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    # the default draws on the device: two losses differ
    m.noise_fn = None
    args = [torch.from_numpy(g[k]) for k in ("users", "pos", "neg")]
    assert m.loss(*args).item() != m.loss(*args).item()


@pytest.mark.parametrize("model", ["DHCF", "LGMRec", "POWERec", "SMORE", "MMGCL", "FKAN_GCF", "LightGT", "GUME", "VGCL", "DDRec", "DCCF", "MICRO", "MENTOR", "HCCF", "LightGCL", "SGL", "BM3", "MGCL", "LATTICE",
                                   "SimGCL", "XSimGCL", "SLMRec", "NCL", "SelfCF", "MCLN", "MMSSL", "GRCN", "MGAT"])
def test_round5_members_train_through_the_main_entry(dev, model, tmp_path, monkeypatch):
    """python -m chaorec_amd.main --Model X --data_path baby --synthetic at the real baby size (the first point of the model's
    grid, two epochs): sampler, per-epoch hooks, training steps, device ranking + metrics, logging."""
    import logging
    from chaorec_amd import main as cmain, dataload
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(dataload.SYNTHETIC_FEATURE_DIMS, "default", (96, 64))
    real = cmain.load_yaml_config
    monkeypatch.setattr(cmain, "load_yaml_config",
                        lambda name: {k: (v if k == "hyper_parameters" else v[:1]) for k, v in real(name).items()})
    logging.getLogger().handlers.clear()
    best = cmain.main(["--Model", model, "--data_path", "baby", "--synthetic", "--num_epoch", "2"])
    assert set(best.keys()) == {5, 10, 20}
    for k in best:
        assert 0.0 <= best[k]["recall"] <= 1.0 and np.isfinite(best[k]["ndcg"])


def test_mmgcl_golden(dev):
    """Model/MMGCL.py: the per-step edge-dropped and node-dropped graphs as VALUE arrays over the one CSR (dynamic-values SpMM,
    forward and backward) against the reference's rebuilt scipy Laplacians, with the reference run's recorded draws: loss,
    every gradient, the tables, the ranking.  Then the device draws: the right number of edges / nodes kept."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MMGCL
    g = load_golden("mmgcl_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = MMGCL(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
              torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_alpha"]), float(g["ssl_temp"]),
              float(g["dropout"]), dev).to(dev)
    want = _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I))
    got = _csr_dense(m.norm_adj)
    assert np.array_equal(got != 0, want != 0) and np.abs(got - want).max() <= 2e-7

    def edge_keep(n, rate):
        k = torch.zeros(n, dtype=torch.bool)
        k[torch.from_numpy(g["edge_keep_idx"])] = True
        return k

    def node_keep(nu, ni, rate):
        ku, ki = torch.ones(nu, dtype=torch.bool), torch.ones(ni, dtype=torch.bool)
        ku[torch.from_numpy(g["drop_user_idx"])] = False
        ki[torch.from_numpy(g["drop_item_idx"])] = False
        return ku, ki

    m.edge_keep_fn, m.node_keep_fn, m.modality_fn = edge_keep, node_keep, lambda: int(g["modality"])
    _golden_model_checks(m, g, dev, 1e-4, 1e-7)
    res = torch.cat([m.result_user, m.result_item], 0).detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    # the default draws (device): exactly int(E (1 - rate)) pairs kept; dropped nodes lose every entry
    m.edge_keep_fn = m.node_keep_fn = m.modality_fn = None
    adj = m.random_graph_augment(0)
    assert int((adj.val[:m.n_edges] != 0).sum()) == int(m.n_edges * (1 - m.dropout_rate))
    assert torch.equal(adj.val[m.n_edges:], adj.val[:m.n_edges][m._lower])
    adj = m.random_graph_augment(1)
    dense = np.zeros((U + I, U + I), np.float32)
    rp, col = m.norm_adj.rowptr.cpu().numpy(), m.norm_adj.col.cpu().numpy()
    v = adj.val.cpu().numpy()
    for r in range(U + I):
        dense[r, col[rp[r]:rp[r + 1]]] = v[rp[r]:rp[r + 1]]
    assert np.array_equal(dense, dense.T)
    empty_rows = int(((dense != 0).sum(1) == 0).sum())
    assert empty_rows >= int(U * m.dropout_rate) + int(I * m.dropout_rate)
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    assert bool(torch.isfinite(loss))


def test_fkan_gcf_golden(dev):
    """Model/FKAN_GCF.py + kanlayer.py: L E through sparse.mm, the Fourier KAN layer's einsum as one GEMM over the cos / sin
    features; parameters in the reference's order, adjacency, loss, every gradient, tables, ranking.  Then node dropout (the
    family's sparse_dropout over the fixed structure) trains."""
    from chaorec_amd import graph
    from chaorec_amd.Model import FKAN_GCF
    g = load_golden("fkan_gcf_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = FKAN_GCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), float(g["reg"]), int(g["L"]), 0.0, 0.0,
                 int(g["G"]), dev).to(dev)
    assert np.array_equal(_csr_dense(m.norm_adj_matrix), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))
    _golden_model_checks(m, g, dev, 5e-5, 1e-8)
    res = torch.cat([m.user_emb_final, m.item_emb_final], 0).detach().cpu().numpy()
    assert np.abs(res - g["result"]).max() <= 5e-6 * np.abs(g["result"]).max()
    _check_rank(m.gene_ranklist(topk=int(g["topk"])).numpy(), g, g["result"][:U] @ g["result"][U:].T, U)
    m.node_dropout, m.message_dropout = 0.2, 0.1
    m.train()
    loss = m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    loss.backward()
    assert bool(torch.isfinite(loss)) and all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_lightgt_golden(dev):
    """Model/LightGT.py: the propagate's prefix means as running sums over sparse.mm, the two feature projections on the MFMA
    GEMM, the encoders in torch, with the reference's batch format (dataload.py:89-101) -- loss and every gradient (the
    unused template layers have none, as there) -- and the evaluation as ONE ranking call over concatenated tables against
    the reference's per-batch score matrices (:369-410)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGT
    g = load_golden("lightgt_small.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = LightGT(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), int(g["L"]), dev).to(dev)
    assert np.array_equal(_csr_dense(m.norm_adj_mat), _coo_dense(g["norm_idx"], g["norm_val"], (U + I, U + I)))
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), g["p_" + n]), n
    m.eval()
    users2 = torch.stack((torch.from_numpy(g["users"]), torch.from_numpy(g["users"])), 1)
    items2 = torch.stack((torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"])), 1)
    loss = m.loss(users2, items2, torch.from_numpy(g["mask"]), torch.from_numpy(g["user_item"]))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    unused = set(str(n) for n in g["no_grad"])
    assert unused and all("encoder_layer" in n for n in unused)
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None, n
            continue
        ref = g["g_" + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-8, n
    batches = [(torch.from_numpy(g[f"eval_users{k}"]), torch.from_numpy(g[f"eval_user_item{k}"]), torch.from_numpy(g[f"eval_mask{k}"]))
               for k in (0, 1)]
    with torch.no_grad():
        ut, it = m.user_tables(batches)
    sc = (ut @ it.T).cpu().numpy()
    assert np.abs(sc - g["scores"]).max() <= 1e-5 * np.abs(g["scores"]).max()
    rank = m.gene_ranklist(batches, topk=int(g["topk"])).numpy()
    ref_sc = g["scores"].copy()
    for u, items in graph.user_item_dict_from_edges(g["edges"]).items():
        ref_sc[u, np.asarray(items) - U] = 1e-5
    ok, why = tie_aware_rank_equal(rank, np.take_along_axis(ref_sc, rank - U, 1), g["rank"],
                                   np.take_along_axis(ref_sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why
    # the device-drawn sequences (every history here is shorter than src_len: the same sets, hence the same ranking)
    rank2 = m.gene_ranklist(topk=int(g["topk"])).numpy()
    ok, why = tie_aware_rank_equal(rank2, np.take_along_axis(ref_sc, rank2 - U, 1), g["rank"],
                                   np.take_along_axis(ref_sc, g["rank"] - U, 1), rtol=1e-4, atol=1e-7)
    assert ok, why


def test_history_sequences_are_the_reference_sampler(dev):
    """dataload.history_sequences against dataload.py:89-101: short histories whole, long ones a uniform subset of src_len
    (every item of a 200-item history kept about src_len / 200 of the time), padding and mask as there."""
    from chaorec_amd import dataload
    rowptr = torch.tensor([0, 3, 3, 203, 215], device=dev)
    col = torch.arange(215, dtype=torch.int32, device=dev)
    users = torch.tensor([2, 0, 1, 3], device=dev).repeat(500)
    gen = torch.Generator(device=dev).manual_seed(3)
    ui, mask = dataload.history_sequences((rowptr, col), users, 50, gen)
    assert ui.shape == (2000, 51) and bool((ui[:, 0] == -1).all())
    assert torch.equal(ui[1, 1:4].sort().values, torch.tensor([0, 1, 2], device=dev)) and bool((ui[1, 4:] == 0).all())
    assert mask[1].tolist() == [False] * 4 + [True] * 47 and mask[2].tolist() == [False] + [True] * 50
    assert not bool(mask[0].any()) and mask[3].tolist() == [False] * 13 + [True] * 38
    long = ui[0::4, 1:]                                    # user 2: 200 items (3 .. 202), 50 kept per draw
    assert bool(((long >= 3) & (long < 203)).all())
    assert all(len(set(r.tolist())) == 50 for r in long[:20])
    freq = torch.bincount(long.flatten(), minlength=203)[3:203].float() / 500
    assert float((freq - 0.25).abs().max()) < 0.08         # (binomial(500, 1/4): sd 0.019)
