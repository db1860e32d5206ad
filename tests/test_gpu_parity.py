"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the reference goldens.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Bars: bit-exact for SpMM / scoring / GEMM / sampler / top-K indices (order-defined arithmetic);
stated tolerances where libm or reduction order differs (BPR exp/log, atomics in the backward).
"""
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
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _rand_graph(U, I, deg_lo, deg_hi, seed, skew=False):
    rng = np.random.default_rng(seed)
    rows = []
    for u in range(U):
        k = int(rng.integers(deg_lo, deg_hi + 1))
        if skew and u % 97 == 0:
            k = min(I, 40 * k)
        for i in rng.choice(I, min(k, I), replace=False):
            rows.append((u, int(i) + U))
    return np.array(rows, dtype=np.int32)


def graph_dict(edges):
    from chaorec_amd import graph
    return graph.user_item_dict_from_edges(edges)


def _csr_to_dev(csr_np, n, dev, symmetric=True):
    from chaorec_amd.graph import CSR
    rowptr, col, val = csr_np
    return CSR(torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
               n, n, symmetric)


# ------------------------------------------------------------------------------------------ SpMM
@pytest.mark.parametrize("D", [4, 8, 32, 64, 128, 256, 384, 768])
def test_spmm_bit_exact_random_graph(dev, oracle, D):
    from chaorec_amd import ops
    U, I = 301, 157
    e = _rand_graph(U, I, 0, 9, seed=D, skew=True)  # empty rows, ragged rows, a few long rows
    N = U + I
    csr = oracle.lightgcn_csr(e, N)
    x = np.random.default_rng(D + 1).standard_normal((N, D)).astype(np.float32)
    want = oracle.spmm(csr, x)
    got = ops.spmm_raw(_csr_to_dev(csr, N, dev), torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(got, want)


def test_spmm_epilogues_bit_exact(dev, oracle):
    from chaorec_amd import ops
    U, I, D = 200, 120, 64
    e = _rand_graph(U, I, 1, 12, seed=3)
    N = U + I
    csr = oracle.lightgcn_csr(e, N)
    g = _csr_to_dev(csr, N, dev)
    rng = np.random.default_rng(0)
    x, z, a0 = (rng.standard_normal((N, D)).astype(np.float32) for _ in range(3))
    tx, tz = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    # alpha/beta epilogue
    want = oracle.spmm(csr, x, alpha=0.25, z=z, beta=1.0 / 3.0)
    got = ops.spmm_raw(g, tx, alpha=0.25, z=tz, beta=1.0 / 3.0).cpu().numpy()
    assert np.array_equal(got, want)
    # acc epilogue, first layer (acc_init) then running
    acc_w = np.float32(1.0 / 3.0)
    acc_np = np.zeros((N, D), np.float32)
    y1 = oracle.spmm(csr, x, acc=acc_np, acc_init=x, acc_w=acc_w)
    y2 = oracle.spmm(csr, y1, acc=acc_np, acc_w=acc_w)
    acc_t = torch.empty((N, D), dtype=torch.float32, device=dev)
    t1 = ops.spmm_raw(g, tx, acc=acc_t, acc_init=tx, acc_w=float(acc_w))
    t2 = ops.spmm_raw(g, t1, acc=acc_t, acc_w=float(acc_w))
    assert np.array_equal(t2.cpu().numpy(), y2)
    assert np.array_equal(acc_t.cpu().numpy(), acc_np)
    # acc only (y == NULL)
    acc2 = acc_t.clone()
    ops.spmm_raw(g, t2, acc=acc2, acc_w=0.5, want_y=False)
    acc_np2 = acc_np.copy()
    oracle.spmm(csr, y2, acc=acc_np2, acc_w=0.5, want_y=False)
    assert np.array_equal(acc2.cpu().numpy(), acc_np2)


def test_spmm_empty_and_errors(dev):
    from chaorec_amd import ops
    from chaorec_amd.graph import CSR
    rowptr = torch.zeros(5, dtype=torch.int64, device=dev)
    g = CSR(rowptr, torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, device=dev), 4, 4, True)
    y = ops.spmm_raw(g, torch.ones(4, 8, device=dev))
    assert torch.all(y == 0)
    with pytest.raises(RuntimeError):
        ops.spmm_raw(g, torch.ones(4, 6, device=dev))  # D % 4 != 0
    with pytest.raises(RuntimeError):
        ops.spmm_raw(g, torch.ones(4, 8))  # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------ LightGCN
def _make_lightgcn(g, U, I, edges, D, L, reg, x0, dev):
    from chaorec_amd.Model import LightGCN
    from chaorec_amd import graph
    m = LightGCN(U, I, edges, graph.user_item_dict_from_edges(edges), D, reg, L, "add", dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(x0[:U]))
        m.item_embedding.weight.copy_(torch.from_numpy(x0[U:]))
    return m.to(dev)


def test_lightgcn_tiny_golden(dev):
    g = load_golden("lightgcn_tiny.npz")
    U, I, D, L = int(g["U"]), int(g["I"]), int(g["D"]), int(g["L"])
    m = _make_lightgcn(g, U, I, g["edges"], D, L, float(g["reg"]), g["x0"], dev)
    loss = m.loss(torch.from_numpy(g["users"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"]))
    loss.backward()
    assert np.array_equal(m.result.detach().cpu().numpy(), g["result"])  # forward: bit-exact vs reference
    assert float(loss) == pytest.approx(float(g["loss"]), rel=2e-6)
    assert np.allclose(m.user_embedding.weight.grad.cpu().numpy(), g["g_user"], rtol=1e-4, atol=1e-7)
    assert np.allclose(m.item_embedding.weight.grad.cpu().numpy(), g["g_item"], rtol=1e-4, atol=1e-7)
    with torch.no_grad():
        emb = m.result
        u, p, n = (torch.from_numpy(g[k]).to(dev) for k in ("users", "pos", "neg"))
        assert float(m.bpr_loss(u, p - U, n - U, emb)) == pytest.approx(float(g["bpr"]), rel=2e-6)
        assert float(m.regularization_loss(u, p - U, n - U, emb)) == pytest.approx(float(g["reg_loss"]), rel=1e-5)
    rank = m.gene_ranklist(topk=int(g["topk"]))
    assert rank.dtype == torch.int64 and rank.device.type == "cpu"
    sc = g["result"][:U] @ g["result"][U:].T
    for u, items in graph_dict(g["edges"]).items():
        sc[u, np.asarray(items) - U] = 1e-6
    got_val = np.take_along_axis(sc, rank.numpy() - U, 1)
    ok, why = tie_aware_rank_equal(rank.numpy(), got_val, g["rank"], g["rank_val"], rtol=1e-5, atol=1e-7)
    assert ok, why


def test_lightgcn_baby_golden_and_metrics(dev, baby, oracle):
    from chaorec_amd import utils
    g = load_golden("lightgcn_baby.npz")
    U, I, D, L = baby["U"], baby["I"], int(g["D"]), int(g["L"])
    x0 = (np.random.default_rng(int(g["x0_seed"])).standard_normal((U + I, D)) * float(g["x0_scale"])).astype(np.float32)
    m = _make_lightgcn(g, U, I, baby["train"], D, L, float(g["reg"]), x0, dev)
    loss = m.loss(torch.from_numpy(g["users"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"]))
    loss.backward()
    rows = g["rows"]
    res = m.result.detach().cpu().numpy()
    assert np.array_equal(res[rows], g["result_rows"])
    assert res.astype(np.float64).sum() == pytest.approx(float(g["result_sum"]), rel=1e-9)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=2e-6)
    grad = np.concatenate([m.user_embedding.weight.grad.cpu().numpy(), m.item_embedding.weight.grad.cpu().numpy()], 0)
    assert np.allclose(grad[rows], g["g_rows"], rtol=2e-4, atol=1e-9)
    assert np.abs(grad).sum() == pytest.approx(float(g["g_abs_sum"]), rel=1e-4)
    # ranking: HIP top-50 == oracle top-50 exactly (same dot-product order), and == reference up to near-ties
    rank = m.gene_ranklist().numpy()
    hist = oracle.user_hist_csr(baby["train"], U)
    o_idx, o_val = oracle.gene_ranklist(res, U, I, hist, 1e-6, 50)
    assert np.array_equal(rank, o_idx)
    urows = g["urows"]
    ok, why = tie_aware_rank_equal(rank[urows], o_val[urows], g["rank_rows"].astype(np.int64), g["rank_val_rows"],
                                   rtol=2e-5, atol=1e-8)
    assert ok, why
    # Recall/NDCG within 1e-4 of the reference's numbers (north_star bar)
    k_list = [int(k) for k in g["k_list"]]
    names = list(g["metric_names"])
    for split, key in ((baby["val"], "val_metrics"), (baby["test"], "test_metrics")):
        mt = utils.gene_metrics(split, torch.from_numpy(rank), k_list)
        got = np.array([[mt[k][n] for n in names] for k in k_list])
        assert np.abs(got - g[key]).max() < 1e-4


# ------------------------------------------------------------------------------------------ BPR / sampler
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("D", [64, 128, 16])
def test_bpr_fwd_bwd_vs_oracle(dev, oracle, variant, D):
    from chaorec_amd import ops
    rng = np.random.default_rng(variant * 10 + D)
    U, I, B = 97, 55, 300  # duplicates inside the batch are certain
    tu = (rng.standard_normal((U, D)) * 0.3).astype(np.float32)
    ti = (rng.standard_normal((I, D)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    reg = 1e-3 if variant == 0 else 0.0
    out, coef = oracle.bpr_fwd(tu, ti, users, pos, neg, variant, reg)
    g_u, g_i = oracle.bpr_bwd(tu, ti, users, pos, neg, coef, reg, grad_out=0.7)
    a = torch.from_numpy(tu).to(dev).requires_grad_(True)
    b = torch.from_numpy(ti).to(dev).requires_grad_(True)
    res = ops.bpr_loss(a, b, *(torch.from_numpy(t).to(dev) for t in (users, pos, neg)), variant, reg)
    (res[0] * 0.7).backward()
    assert np.allclose(res.detach().cpu().numpy(), out, rtol=3e-6, atol=1e-8)
    assert np.allclose(a.grad.cpu().numpy(), g_u, rtol=2e-5, atol=1e-8)
    assert np.allclose(b.grad.cpu().numpy(), g_i, rtol=2e-5, atol=1e-8)


def test_bpr_single_table_offset(dev, oracle):
    from chaorec_amd import ops
    rng = np.random.default_rng(5)
    U, I, B, D = 40, 30, 64, 64
    tab = (rng.standard_normal((U + I, D)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    out, coef = oracle.bpr_fwd(tab[:U], tab[U:], users, pos, neg, 0, 1e-3)
    g_u, g_i = oracle.bpr_bwd(tab[:U], tab[U:], users, pos, neg, coef, 1e-3)
    t = torch.from_numpy(tab).to(dev).requires_grad_(True)
    res = ops.bpr_loss(t, None, *(torch.from_numpy(x).to(dev) for x in (users, pos, neg)), 0, 1e-3, item_offset=U)
    res[0].backward()
    assert np.allclose(res.detach().cpu().numpy(), out, rtol=3e-6)
    assert np.allclose(t.grad.cpu().numpy(), np.concatenate([g_u, g_i]), rtol=2e-5, atol=1e-8)


def test_bpr_loss_is_run_to_run_identical(dev):
    from chaorec_amd import ops
    rng = np.random.default_rng(1)
    tab = torch.from_numpy((rng.standard_normal((500, 64)) * 0.3).astype(np.float32)).to(dev)
    idx = [torch.from_numpy(rng.integers(0, 250, 1024)).to(dev) for _ in range(3)]
    a = ops.bpr_loss(tab, None, *idx, 0, 1e-3, item_offset=250)
    b = ops.bpr_loss(tab, None, *idx, 0, 1e-3, item_offset=250)
    assert torch.equal(a.detach(), b.detach())


def test_sampler_bit_exact_and_never_in_history(dev, oracle, baby):
    from chaorec_amd import ops
    U, I = baby["U"], baby["I"]
    hist = oracle.user_hist_csr(baby["train"], U)
    users = baby["train"][:8192, 0].astype(np.int64)
    want = oracle.sample_negatives(hist, users, I, seed=42, step=7, id_offset=U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    got = ops.sample_negatives(dh, torch.from_numpy(users).to(dev), I, 42, 7, U).cpu().numpy()
    assert np.array_equal(got, want)
    for u, n in zip(users[:2000], got[:2000]):
        row = hist[1][hist[0][u]:hist[0][u + 1]]
        assert (n - U) not in row and U <= n < U + I
    other = ops.sample_negatives(dh, torch.from_numpy(users).to(dev), I, 42, 8, U).cpu().numpy()
    assert (other != got).mean() > 0.99  # a new step is a new draw
    # the sample's SECOND draw (dataload.py:81-84, read by MCLN): its own stream of the same generator -- the oracle's bits,
    # never in the history, independent of the first (equal to it about once in I draws)
    want2 = oracle.sample_negatives(hist, users, I, seed=42, step=7, id_offset=U, second=True)
    got2 = ops.sample_negatives(dh, torch.from_numpy(users).to(dev), I, 42, 7, U, second=True).cpu().numpy()
    assert np.array_equal(got2, want2)
    for u, n in zip(users[:2000], got2[:2000]):
        assert (n - U) not in hist[1][hist[0][u]:hist[0][u + 1]] and U <= n < U + I
    assert (got2 == got).mean() < 5.0 / I + 0.002


def test_sampler_uniform_over_unseen(dev, oracle):
    from chaorec_amd import ops
    g = load_golden("sampler_tiny.npz")
    U, I = int(g["U"]), int(g["I"])
    hist = oracle.user_hist_csr(g["edges"], U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    users = np.repeat(np.arange(U), 4000).astype(np.int64)
    neg = ops.sample_negatives(dh, torch.from_numpy(users).to(dev), I, 1, 0, U).cpu().numpy() - U
    mine = np.zeros((U, I), np.int64)
    np.add.at(mine, (users, neg), 1)
    assert np.array_equal(mine > 0, g["neg_hist"] > 0)  # same support as the reference's sampler
    for u in range(U):
        allowed = mine[u] > 0
        exp = 4000 / allowed.sum()
        assert ((mine[u][allowed] - exp) ** 2 / exp).sum() < 30


# ------------------------------------------------------------------------------------------ scoring + top-K
def _hist_random(U, I, max_deg, seed):
    rng = np.random.default_rng(seed)
    rowptr = np.zeros(U + 1, np.int64)
    cols = []
    for u in range(U):
        k = int(rng.integers(0, max_deg + 1))
        c = np.sort(rng.choice(I, min(k, I), replace=False)).astype(np.int32)
        cols.append(c)
        rowptr[u + 1] = rowptr[u] + len(c)
    return rowptr, (np.concatenate(cols) if cols else np.zeros(0, np.int32)).astype(np.int32)


@pytest.mark.parametrize("U,I,D,K", [(1, 64, 64, 50), (33, 50, 64, 50), (70, 1000, 64, 50), (257, 4794, 64, 50),
                                     (40, 3000, 128, 10), (40, 2000, 32, 64), (5, 9000, 64, 1)])
def test_score_topk_bit_exact_vs_oracle(dev, oracle, U, I, D, K):
    from chaorec_amd import ops
    rng = np.random.default_rng(U * 7 + I)
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    hist = _hist_random(U, I, 20, seed=I)
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, id_offset=U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K, id_offset=U)
    assert np.array_equal(got_v.cpu().numpy(), want_v)   # same fmaf chain -> same bits
    assert np.array_equal(got_i.cpu().numpy(), want_i)


def test_score_topk_integer_ties_lowest_index_first(dev, oracle):
    """Integer-valued embeddings: every score is exact in any summation order, ties are massive."""
    from chaorec_amd import ops
    rng = np.random.default_rng(0)
    U, I, D, K = 64, 3000, 64, 50
    ue = rng.integers(-2, 3, (U, D)).astype(np.float32)
    ie = rng.integers(-1, 2, (I, D)).astype(np.float32)
    hist = _hist_random(U, I, 30, seed=1)
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, 0)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K)
    assert np.array_equal(got_v.cpu().numpy(), want_v)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    # and against a plain fp32 matmul + stable sort (what torch.matmul gives on exact integers)
    sc = ue @ ie.T
    for u in range(U):
        sc[u, hist[1][hist[0][u]:hist[0][u + 1]]] = 1e-6
    ref = np.argsort(-sc, axis=1, kind="stable")[:, :K]
    assert np.array_equal(got_i.cpu().numpy(), ref)


def test_score_topk_mask_quirk_and_no_hist(dev, oracle):
    """Q7: the mask value 1e-6 outranks negative scores; hist=None is the kNN use."""
    from chaorec_amd import ops
    rng = np.random.default_rng(3)
    U, I, D, K = 8, 200, 64, 50
    ue = np.abs(rng.standard_normal((U, D))).astype(np.float32)
    ie = -np.abs(rng.standard_normal((I, D))).astype(np.float32)  # all scores negative
    hist = _hist_random(U, I, 10, seed=2)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K)
    got_i, got_v = got_i.cpu().numpy(), got_v.cpu().numpy()
    for u in range(U):
        h = hist[1][hist[0][u]:hist[0][u + 1]]
        assert np.array_equal(got_i[u, :len(h)], h)           # history first, ascending index
        assert np.all(got_v[u, :len(h)] == np.float32(1e-6))
    want_i, want_v = oracle.score_topk(ue, ie, None, 0.0, 10, 0)
    gi, gv = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), None, 0.0, 10)
    assert np.array_equal(gi.cpu().numpy(), want_i) and np.array_equal(gv.cpu().numpy(), want_v)


def test_score_topk_sampled_threshold_fallback(dev, oracle):
    """Adversarial for the sampled threshold: the sampled tiles (every 10th at K=50) hold all the big scores, so
    tau0 is far too high, fewer than K items pass and the certification must send every user through the
    unthresholded fallback.  The answer is still the exact top-K."""
    from chaorec_amd import ops
    rng = np.random.default_rng(11)
    U, I, D, K = 100, 6400, 64, 50
    ue = np.abs(rng.standard_normal((U, D))).astype(np.float32) * 0.2
    ie = (rng.standard_normal((I, D)) * 0.01).astype(np.float32)
    tiles = np.arange(I) // 32
    hot = (tiles % 10 == 0) & (np.arange(I) % 32 < 3)          # 3 hot items in every sampled tile: 60 in total
    ie[hot] = np.abs(rng.standard_normal((hot.sum(), D))).astype(np.float32)
    hist = _hist_random(U, I, 20, seed=4)
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, 7)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    tu, ti = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    for precision in (0, 1, 2):
        got_i, got_v = ops.score_topk(tu, ti, dh, 1e-6, K, id_offset=7, precision=precision)
        assert np.array_equal(got_v.cpu().numpy(), want_v), precision
        assert np.array_equal(got_i.cpu().numpy(), want_i), precision


# ---- bf16 prefilter + exact fp32 re-score (precision 0 at >= 4096 items, D in {64, 128}) -------------------
def _check_all_precisions(dev, oracle, ue, ie, hist, mask, K, id_offset=0):
    from chaorec_amd import ops
    want_i, want_v = oracle.score_topk(ue, ie, hist, mask, K, id_offset)
    dh = None if hist is None else (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    tu, ti = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    for precision in (0, 2, 1):
        got_i, got_v = ops.score_topk(tu, ti, dh, mask, K, id_offset=id_offset, precision=precision)
        assert np.array_equal(got_v.cpu().numpy(), want_v), precision     # exact fp32 values, not bf16 ones
        assert np.array_equal(got_i.cpu().numpy(), want_i), precision
    # thresholds carried between calls (chaorec_score_topk_hinted_f32) never change the result, whatever they hold:
    # the previous call's own hints, and hints built to fail in every way (too high, too low, NaN, garbage)
    U = ue.shape[0]
    hint = torch.empty(U, dtype=torch.float32, device=dev)
    rng = np.random.default_rng(U)
    garbage = torch.from_numpy((rng.standard_normal(U) * 3).astype(np.float32)).to(dev)
    checks = [("first", None), ("carried", "keep"), ("+inf", torch.full_like(hint, float("inf"))),
              ("-inf", torch.full_like(hint, float("-inf"))), ("nan", torch.full_like(hint, float("nan"))),
              ("garbage", garbage), ("huge", torch.full_like(hint, 3e38)), ("zero", torch.zeros_like(hint)),
              ("carried again", "keep"), ("light carried", "keep"), ("light garbage", garbage), ("light after garbage", "keep")]
    counters = torch.zeros(4, dtype=torch.int32, device=dev)
    for name, h in checks:
        valid = h is not None
        if valid and not isinstance(h, str):
            hint.copy_(h)
        got_i, got_v = ops.score_topk(tu, ti, dh, mask, K, id_offset=id_offset, hint=hint, hint_valid=valid,
                                      light=name.startswith("light"), counters=counters)
        assert np.array_equal(got_v.cpu().numpy(), want_v), name
        assert np.array_equal(got_i.cpu().numpy(), want_i), name
        c = counters.tolist()
        assert 0 <= c[0] <= U and 0 <= c[1] <= U and c[3] <= c[0], (name, c)
        if name.startswith("light") and ie.shape[0] < 131072:     # (very long item ranges keep their retry pass)
            assert c[3] == c[0], (name, c)        # no retry pass: everything pass A queued went to the exact route


@pytest.mark.parametrize("U,I,D,K", [(150, 9000, 64, 50), (70, 8200, 128, 20), (33, 12000, 64, 64),
                                     (129, 8192, 64, 1), (300, 15207, 64, 50), (64, 150000, 64, 50),
                                     (40, 560000, 64, 50),      # > 16384 tiles: coarser sample, more sweep splits
                                     (16, 1100000, 64, 50)])    # > 32768 tiles: every 16th tile sampled
def test_score_topk_prefilter_bit_exact_vs_oracle(dev, oracle, U, I, D, K):
    rng = np.random.default_rng(U + I + D)
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    ie *= (0.3 + rng.pareto(3.0, (I, 1))).astype(np.float32)        # heavy-tailed item norms, like trained tables
    _check_all_precisions(dev, oracle, ue, ie, _hist_random(U, I, 40, seed=I), 1e-6, K, id_offset=U)
    if I >= 100000:
        # a long item range must not degrade the sampled thresholds (streaming top-r in the sampler): the prefilter
        # route has to certify (nearly) everyone itself
        from chaorec_amd import ops
        st = {}
        ie = (np.random.default_rng(1).standard_normal((I, D)) * 0.2).astype(np.float32)   # (no norm outliers here)
        ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), None, 0.0, K, stats=st)
        assert st["prefilter_users"] == U and st["fallback_users"] <= U // 8, st
        assert st["candidates"] / U < 600, st


def test_score_topk_prefilter_adversarial(dev, oracle):
    """Inputs built against each assumption of the prefilter; every one must still give the exact top-K
    (through the certification + fp32 fallback where the bf16 pass cannot decide)."""
    rng = np.random.default_rng(21)
    U, I, D, K = 96, 9600, 64, 50
    base_u = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    base_i = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    hist = _hist_random(U, I, 25, seed=9)
    # (a) massive exact ties: integer embeddings
    _check_all_precisions(dev, oracle, rng.integers(-2, 3, (U, D)).astype(np.float32),
                          rng.integers(-1, 2, (I, D)).astype(np.float32), hist, 1e-6, K)
    # (b) all scores equal (zero users) and all-zero tables
    ue = base_u.copy()
    ue[::3] = 0.0
    _check_all_precisions(dev, oracle, ue, base_i, hist, 1e-6, K)
    _check_all_precisions(dev, oracle, np.zeros((U, D), np.float32), np.zeros((I, D), np.float32), hist, 1e-6, K)
    # (c) the sampled tiles (every 4th) hold all the big scores: threshold far too high
    ie = (base_i * 0.05).astype(np.float32)
    tiles = np.arange(I) // 32
    hot = (tiles % 4 == 0) & (np.arange(I) % 32 < 2)
    ie[hot] = np.abs(rng.standard_normal((hot.sum(), D))).astype(np.float32)
    _check_all_precisions(dev, oracle, np.abs(base_u), ie, hist, 1e-6, K)
    # (d) the big scores sit in NO sampled tile and in one narrow item range: threshold far too low, lists overflow
    ie = (base_i * 0.05).astype(np.float32)
    ie[33:33 + 3 * 32] = np.abs(rng.standard_normal((96, D))).astype(np.float32) * 3
    _check_all_precisions(dev, oracle, np.abs(base_u), ie, hist, 1e-6, K)
    # (e) one item with a huge norm blows up the error bound: the 2m band swallows far more than 128 candidates
    ie = base_i.copy()
    ie[777] *= 1e4
    _check_all_precisions(dev, oracle, base_u, ie, hist, 1e-6, K)
    # (f) scores that differ only below bf16 resolution: items are tiny perturbations of one vector
    v = rng.standard_normal(D).astype(np.float32)
    ie = (v[None, :] * (1.0 + 1e-4 * rng.standard_normal((I, 1)))).astype(np.float32)
    _check_all_precisions(dev, oracle, base_u, ie, hist, 1e-6, K)
    # (g) negative scores everywhere with the positive mask value (Q7) and no history at all
    _check_all_precisions(dev, oracle, np.abs(base_u), -np.abs(base_i), hist, 1e-6, K)
    _check_all_precisions(dev, oracle, base_u, base_i, None, 0.0, K)


def test_score_topk_carried_thresholds_cut_the_work(dev, oracle):
    """Second call on slightly moved tables with the first call's hints: no user needs the retry pass or the exact
    route, about K * 1.6 + a band of candidates per user instead of the sampled threshold's several K; a call on
    tables that moved A LOT is still exact (retry pass)."""
    from chaorec_amd import ops
    rng = np.random.default_rng(5)
    U, I, D, K = 2000, 15207, 64, 50
    ue = (rng.standard_normal((U, D)) * 0.3).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.3).astype(np.float32)
    hist = _hist_random(U, I, 30, seed=2)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    tu, ti = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    hint = torch.empty(U, dtype=torch.float32, device=dev)
    st0, st1, st2 = {}, {}, {}
    ops.score_topk(tu, ti, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=False, stats=st0)
    ue2 = (ue + rng.standard_normal((U, D)).astype(np.float32) * 0.003).astype(np.float32)     # "one epoch later"
    ie2 = (ie + rng.standard_normal((I, D)).astype(np.float32) * 0.003).astype(np.float32)
    want_i, want_v = oracle.score_topk(ue2, ie2, hist, 1e-6, K, U)
    gi, gv = ops.score_topk(torch.from_numpy(ue2).to(dev), torch.from_numpy(ie2).to(dev), dh, 1e-6, K, id_offset=U,
                            hint=hint, hint_valid=True, stats=st1)
    assert np.array_equal(gi.cpu().numpy(), want_i) and np.array_equal(gv.cpu().numpy(), want_v)
    assert st1["fallback_users"] == 0, st1
    assert st1["candidates"] / U < 0.75 * st0["candidates"] / U and st1["candidates"] / U < 3 * K, (st0, st1)
    ue3 = (ue * 0.3 + rng.standard_normal((U, D)).astype(np.float32) * 0.2).astype(np.float32)   # a different model
    want_i, want_v = oracle.score_topk(ue3, ie2, hist, 1e-6, K, U)
    gi, gv = ops.score_topk(torch.from_numpy(ue3).to(dev), torch.from_numpy(ie2).to(dev), dh, 1e-6, K, id_offset=U,
                            hint=hint, hint_valid=True, stats=st2)
    assert np.array_equal(gi.cpu().numpy(), want_i) and np.array_equal(gv.cpu().numpy(), want_v)
    print("candidates per user: sampled", st0["candidates"] / U, "carried", st1["candidates"] / U, "stale", st2)


def test_score_topk_single_pass_equals_sampled(dev):
    from chaorec_amd import ops
    torch.manual_seed(3)
    U, I, D = 3000, 9000, 64
    emb_u, emb_i = torch.randn(U, D, device=dev) * 0.1, torch.randn(I, D, device=dev) * 0.1
    a = ops.score_topk(emb_u, emb_i, None, 0.0, 50, precision=0)
    b = ops.score_topk(emb_u, emb_i, None, 0.0, 50, precision=1)
    c = ops.score_topk(emb_u, emb_i, None, 0.0, 50, precision=2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


def test_score_topk_errors(dev):
    from chaorec_amd import ops
    a = torch.zeros(4, 64, device=dev)
    with pytest.raises(RuntimeError):
        ops.score_topk(a, torch.zeros(10, 64, device=dev), None, 0.0, 50)  # K > n_items, like torch.topk
    with pytest.raises(RuntimeError):
        ops.score_topk(torch.zeros(4, 48, device=dev), torch.zeros(100, 48, device=dev), None, 0.0, 10)


# ------------------------------------------------------------------------------------------ GEMM / Adam
@pytest.mark.parametrize("M,N,K", [(300, 64, 128), (1, 1, 1), (129, 65, 17), (515, 256, 320), (64, 64, 2040),
                                   (260, 130, 70)])
@pytest.mark.parametrize("tA,tB", [(False, True), (False, False), (True, False)])
def test_gemm_bit_exact(dev, oracle, M, N, K, tA, tB):
    from chaorec_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    want = oracle.gemm(A, B, bias=bias, transA=tA, transB=tB, act=1)
    got = ops.gemm_raw(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), transA=tA, transB=tB,
                       bias=torch.from_numpy(bias).to(dev), act=1)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("tA,tB", [(False, True), (True, False)])
def test_gemm_split_k_deterministic_and_close(dev, oracle, tA, tB):
    """Long reductions with few output tiles (the weight-gradient shape) go through split-K slabs summed in
    a fixed order: bit-identical run to run, equal to the oracle's single chain to fp32 rounding."""
    from chaorec_amd import ops, _lib
    M, N, K = 96, 200, 20000
    assert _lib.load().chaorec_gemm_workspace_bytes(M, N, K) > 0
    rng = np.random.default_rng(5)
    A = rng.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    want = oracle.gemm(A, B, bias=bias, transA=tA, transB=tB, act=1)
    a, b, bb = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(bias).to(dev)
    got = ops.gemm_raw(a, b, transA=tA, transB=tB, bias=bb, act=1)
    again = ops.gemm_raw(a, b, transA=tA, transB=tB, bias=bb, act=1)
    assert torch.equal(got, again)
    # fp32 tolerance: |sum| ~ sqrt(K) ~ 141, eps 6e-8 * sqrt(K) terms
    assert np.allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-3)


def test_linear_autograd_vs_torch(dev):
    from chaorec_amd import ops
    torch.manual_seed(0)
    x = torch.randn(333, 96, device=dev, requires_grad=True)
    w = torch.randn(64, 96, device=dev, requires_grad=True)
    b = torch.randn(64, device=dev, requires_grad=True)
    y = ops.linear(x, w, b, act=1)
    y.square().sum().backward()
    gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
    x.grad = w.grad = b.grad = None
    yr = torch.nn.functional.leaky_relu(torch.nn.functional.linear(x.double(), w.double(), b.double()))
    yr.square().sum().backward()
    assert torch.allclose(y.double(), yr, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gx.double(), x.grad.double(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(gw.double(), w.grad.double(), rtol=1e-4, atol=1e-3)
    assert torch.allclose(gb.double(), b.grad.double(), rtol=1e-4, atol=1e-3)


def test_adam_matches_oracle_and_torch(dev, oracle):
    from chaorec_amd import ops
    rng = np.random.default_rng(0)
    n = 10007
    p0 = rng.standard_normal(n).astype(np.float32)
    p_np, m_np, v_np = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    p_t = torch.from_numpy(p0.copy()).to(dev)
    m_t, v_t = torch.zeros_like(p_t), torch.zeros_like(p_t)
    p_ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    for step in range(1, 6):
        g = (rng.standard_normal(n) * 0.1).astype(np.float32)
        oracle.adam_step(p_np, g, m_np, v_np, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
        ops.adam_step(p_t, torch.from_numpy(g).to(dev), m_t, v_t, step)
        p_ref.grad = torch.from_numpy(g.copy())
        opt.step()
    assert np.allclose(p_t.cpu().numpy(), p_np, rtol=0, atol=2e-7)
    assert np.allclose(p_t.cpu().numpy(), p_ref.detach().numpy(), rtol=0, atol=5e-7)


# ------------------------------------------------------------------------------------------ ranking metrics
def test_rank_metrics_device_vs_reference_golden_and_oracle(dev, oracle, baby):
    """chaorec_rank_metrics_f64 against the reference's own numbers (real baby val/test lists, fixed rank list) and
    against the per-user python restatement on lists with every corner case: empty rows, duplicate positives,
    duplicate ranked ids, rows in arbitrary user order, k up to 64."""
    from chaorec_amd import utils
    from conftest import load_golden
    g = load_golden("metrics_baby_fixed_rank.npz")
    k_list = [int(k) for k in g["k_list"]]
    fixed_rank = np.stack([np.random.default_rng(1000 + u).permutation(baby["I"])[:50] + baby["U"]
                           for u in range(baby["U"])]).astype(np.int64)     # the rank list the golden was made with
    rank = torch.from_numpy(fixed_rank).to(dev)
    m = utils.gene_metrics_device(utils.EvalLists(baby["val"], dev), rank, k_list)
    got = np.array([[m[k][n] for n in g["metric_names"]] for k in k_list])
    assert np.allclose(got, g["val_metrics"], rtol=1e-12, atol=0)          # the reference's own numbers
    rng = np.random.default_rng(3)
    U, I, K = 300, 400, 64
    rank_np = np.stack([rng.permutation(I)[:K] for _ in range(U)]).astype(np.int64) + U
    rank_np[5, 3] = rank_np[5, 1]                       # duplicate id inside a ranked list
    rank_np[6, :4] = rank_np[6, 0]
    data = []
    for u in rng.permutation(U)[:250]:
        n = int(rng.integers(0, 6))
        pos = list((rng.integers(0, I, n) + U).tolist())
        if n and rng.random() < 0.3:
            pos.append(pos[0])                          # duplicate positive
        if rng.random() < 0.5 and n:
            pos[0] = int(rank_np[u, rng.integers(0, 10)])   # make hits likely
        data.append([int(u)] + pos)
    data.append([5, int(rank_np[5, 1])])
    data.append([6, int(rank_np[6, 0]), int(rank_np[6, 9])])
    ks = [1, 5, 10, 20, 50, 64]
    ref = oracle.gene_metrics(data, rank_np, ks)
    got = utils.gene_metrics_device(utils.EvalLists(data, dev), torch.from_numpy(rank_np).to(dev), ks)
    again = utils.gene_metrics_device(utils.EvalLists(data, dev), torch.from_numpy(rank_np).to(dev), ks)
    for k in ks:
        for n in ref[k]:
            assert got[k][n] == pytest.approx(ref[k][n], rel=1e-12, abs=1e-15), (k, n)
            assert got[k][n] == again[k][n]             # fixed-order reduction


def test_rank_metrics_errors(dev):
    from chaorec_amd import ops
    rank = torch.zeros((4, 10), dtype=torch.int64, device=dev)
    ru = torch.zeros(2, dtype=torch.int64, device=dev)
    rp = torch.tensor([0, 1, 1], dtype=torch.int64, device=dev)
    it = torch.zeros(1, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):
        ops.rank_metrics(rank, ru, rp, it, [20])         # k beyond the rank list
    with pytest.raises(RuntimeError):
        ops.rank_metrics(rank, ru, rp, it, list(range(1, 10)))   # more than 8 cut-offs


# ------------------------------------------------------------------------------------------ full-size properties
def test_sports_size_properties(dev):
    """BASELINE configs[1] sizes (U=28940, I=15207, E=158554, D=64): size-independent checks."""
    from chaorec_amd import ops, graph
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D = 28940, 15207, 158554, 64
    edges = synthetic_interactions(U, I, E, seed=42)
    N = U + I
    A = graph.lightgcn_csr(edges, N).to(dev)
    torch.manual_seed(0)
    x = torch.randn(N, D, device=dev)
    y = torch.randn(N, D, device=dev)
    # determinism: same launch twice is bit-identical
    a1, a2 = ops.spmm_raw(A, x), ops.spmm_raw(A, x)
    assert torch.equal(a1, a2)
    # linearity within rounding, symmetry <Ax, y> == <x, Ay>
    lhs = ops.spmm_raw(A, x + y)
    rhs = a1 + ops.spmm_raw(A, y)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-5)
    d1 = (a1.double() * y.double()).sum()
    d2 = (x.double() * ops.spmm_raw(A, y).double()).sum()
    assert abs(float(d1 - d2)) <= 1e-6 * abs(float(d1)) + 1e-6
    # spectral bound: ||A x|| <= ||x|| for the symmetric-normalised adjacency
    assert float(a1.norm()) <= float(x.norm()) * (1 + 1e-5)
    # top-K: sorted descending, indices unique and in range, history never above an unmasked positive score
    hist = tuple(t.to(dev) for t in graph.user_hist_csr_from_edges(edges, U))
    emb = torch.randn(N, D, device=dev) * 0.1
    idx, val = ops.score_topk(emb[:U], emb[U:], hist, 1e-6, 50, id_offset=U)
    assert torch.all(val[:, :-1] >= val[:, 1:])
    assert int(idx.min()) >= U and int(idx.max()) < N
    srt = torch.sort(idx, dim=1).values
    assert torch.all(srt[:, 1:] != srt[:, :-1])
    # spot-check 64 users against a dense fp32 matmul + topk
    sel = torch.arange(0, U, U // 64, device=dev)[:64]
    sc = emb[:U][sel] @ emb[U:].T
    rp, col = hist
    for k, u in enumerate(sel.tolist()):
        sc[k, col[rp[u]:rp[u + 1]].long()] = 1e-6
    tv, ti = torch.topk(sc, 50)
    assert torch.allclose(tv, val[sel], rtol=1e-5, atol=1e-7)
    assert (ti + U == idx[sel]).float().mean() > 0.999


# ------------------------------------------------------------------------------------------ fused Adam / hipGraph step
def test_graphed_step_equals_eager_and_torch_adam(dev):
    """FusedAdam + the captured hipGraph step follow the same trajectory as eager torch.optim.Adam."""
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, L, B = 1500, 800, 7000, 64, 2, 256
    edges = synthetic_interactions(U, I, E, seed=2)
    uid = graph.user_item_dict_from_edges(edges)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(6):
        b = rng.choice(E, B, replace=False)
        batches.append((torch.from_numpy(edges[b, 0].astype(np.int64)).to(dev),
                        torch.from_numpy(edges[b, 1].astype(np.int64)).to(dev),
                        torch.from_numpy(rng.integers(U, U + I, B)).to(dev)))
    results = {}
    for mode in ("torch", "fused", "graph"):
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, uid, D, 1e-3, L, "add", dev).to(dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-2) if mode == "torch" else FusedAdam(m.parameters(), lr=1e-2)
        losses = []
        if mode == "graph":
            w0 = [p.detach().clone() for p in m.parameters()]
            g = GraphedTrainStep(m, opt, batches[0], warmup=2)
            # the capture's warm-up steps must not count as training
            assert all(torch.equal(p.detach(), w) for p, w in zip(m.parameters(), w0))
            assert int(opt._step_dev.item()) == 0
            assert all(float(st["exp_avg"].abs().max()) == 0.0 for st in opt.state.values())
            for b in batches:
                losses.append(float(g(*b)))
            assert g.replays == len(batches)
        else:
            for b in batches:
                opt.zero_grad()
                loss = m.loss(*b)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        results[mode] = (losses, torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy())
    for mode in ("fused", "graph"):
        assert np.allclose(results[mode][0], results["torch"][0], rtol=1e-5), mode
        assert np.allclose(results[mode][1], results["torch"][1], rtol=0, atol=5e-6), mode
    assert np.array_equal(results["graph"][1], results["fused"][1]) or \
        np.allclose(results["graph"][1], results["fused"][1], rtol=0, atol=1e-6)


def test_graphed_step_with_in_graph_sampling(dev, oracle):
    """batch_fn captured inside the hipGraph: every replay draws a fresh batch from the device counter, and the
    draws are the oracle's (seed, step) function of that counter."""
    from chaorec_amd import graph, ops
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, B = 900, 500, 4000, 64, 128
    edges = synthetic_interactions(U, I, E, seed=6)
    torch.manual_seed(0)
    m = LightGCN(U, I, edges, graph.user_item_dict_from_edges(edges), D, 1e-3, 2, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-2)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    seen = {}

    def draw():
        sel = torch.randint(0, E, (B,), device=dev)
        counter.add_(1)
        users, pos = edges_dev[sel, 0], edges_dev[sel, 1]
        neg = ops.sample_negatives(m.hist, users, I, 11, 0, U, step_dev=counter)
        seen["users"], seen["neg"] = users, neg
        return users, pos, neg

    g = GraphedTrainStep(m, opt, batch_fn=draw, warmup=2)
    c0 = int(counter.item())
    hist = oracle.user_hist_csr(edges, U)
    losses, negs = [], []
    for k in range(4):
        losses.append(float(g()))
        assert int(counter.item()) == c0 + k + 1
        u_np, n_np = seen["users"].cpu().numpy(), seen["neg"].cpu().numpy()
        want = oracle.sample_negatives(hist, u_np, I, seed=11, step=c0 + k + 1, id_offset=U)
        assert np.array_equal(n_np, want)
        negs.append(n_np.copy())
    assert g.replays == 4 and len(set(losses)) == 4
    assert not np.array_equal(negs[0], negs[1])


def test_draw_batch_one_launch(dev, oracle, baby):
    """chaorec_draw_batch: edges are real training edges, local ids, negatives follow the sampler's stream."""
    from chaorec_amd import ops
    U, I = baby["U"], baby["I"]
    edges = torch.from_numpy(baby["train"].astype(np.int64)).to(dev)
    hist = oracle.user_hist_csr(baby["train"], U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    B = 4096
    counter = torch.tensor([5], dtype=torch.int64, device=dev)
    u, p, n = ops.draw_batch(edges, dh, B, U, I, 42, 2, step_dev=counter)          # effective step 7
    u, p, n = u.cpu().numpy(), p.cpu().numpy(), n.cpu().numpy()
    train = set(map(tuple, baby["train"].tolist()))
    assert all((int(a), int(b) + U) in train for a, b in zip(u[:500], p[:500]))    # picked pairs are training edges
    assert np.array_equal(n, oracle.sample_negatives(hist, u, I, seed=42, step=7, id_offset=0))
    assert len(np.unique(u)) > B // 4                                               # spread over the edge list
    # global item ids straight from the launch (item_offset = num_user), and Model.loss()'s shift back + row list in one
    ug, pg, ng = ops.draw_batch(edges, dh, B, U, I, 42, 2, step_dev=counter, item_offset=U)
    assert np.array_equal(ug.cpu().numpy(), u) and np.array_equal(pg.cpu().numpy(), p + U) and np.array_equal(ng.cpu().numpy(), n + U)
    pl, nl, rows = ops.shift_cat(pg, ng, U)
    assert np.array_equal(pl.cpu().numpy(), p) and np.array_equal(nl.cpu().numpy(), n)
    assert np.array_equal(rows.cpu().numpy(), np.concatenate([p, n])) and rows.data_ptr() == pl.data_ptr()
    u2, _, _ = ops.draw_batch(edges, dh, B, U, I, 42, 8)
    assert (u2.cpu().numpy() != u).mean() > 0.9
    # uniform over edges: users appear in proportion to their degree (chi-square on degree buckets)
    big = torch.cat([ops.draw_batch(edges, dh, 65536, U, I, 1, s)[0] for s in range(4)]).cpu().numpy()
    deg = np.bincount(baby["train"][:, 0], minlength=U)
    got = np.bincount(deg[big], minlength=deg.max() + 1).astype(np.float64)
    exp = np.bincount(deg, weights=deg.astype(np.float64), minlength=deg.max() + 1) / deg.sum() * len(big)
    keep = exp > 50
    assert ((got[keep] - exp[keep]) ** 2 / exp[keep]).sum() < 3 * keep.sum()


# ---- edge dropout + renormalisation, weighted edge sampling ---------------------------------------
@pytest.mark.parametrize("p", [0.0, 0.2, 0.7])
def test_edge_dropout_norm_bit_exact(dev, oracle, p):
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 3000, 1700, 40000
    edges = synthetic_interactions(U, I, E, seed=4)
    s = graph.ngcf_structure(edges, U + I).to(dev)
    er, col, te = (t.cpu().numpy() for t in (s.entry_row, s.col, s.transpose_entry))
    step_dev = torch.tensor([5], dtype=torch.int64, device=dev)
    val, val_t = ops.edge_dropout_norm(s, p, seed=1234, step=2, step_dev=step_dev, salt=3)
    keep = oracle.edge_dropout_keep(s.nnz, p, 1234, 7, 3) | (er == col)
    want, want_t = oracle.edge_dropout_norm(er, col, te, U + I, keep)
    assert np.array_equal(val.cpu().numpy(), want)
    assert np.array_equal(val_t.cpu().numpy(), want_t)
    real = er != col
    assert abs(keep[real].mean() - (1 - p)) < 0.01
    if p == 0.0:       # no dropout: the static D^-1/2 (A+I) D^-1/2 of BasicGCN, to the ulp of pow(-0.5) vs 1/sqrt
        assert np.allclose(want, s.val.cpu().numpy(), rtol=3e-7, atol=0)
    # an externally drawn mask replaces the generator
    ext = (np.random.default_rng(0).random(s.nnz) < 0.5)
    v2, v2t = ops.edge_dropout_norm(s, p, seed=0, keep=torch.from_numpy(ext.astype(np.uint8)).to(dev))
    w2, w2t = oracle.edge_dropout_norm(er, col, te, U + I, ext | (er == col))
    assert np.array_equal(v2.cpu().numpy(), w2) and np.array_equal(v2t.cpu().numpy(), w2t)
    # val_t really is the transpose: (A x) . y == x . (A^T y)
    x = torch.randn(U + I, 64, device=dev, dtype=torch.float64).float()
    y = torch.randn(U + I, 64, device=dev, dtype=torch.float64).float()
    ax = ops.spmm_raw(s.with_values(val), x).double()
    aty = ops.spmm_raw(s.with_values(val_t), y).double()
    assert float((ax * y.double()).sum()) == pytest.approx(float((x.double() * aty).sum()), rel=1e-5)


def test_spmm_values_matches_oracle_spmm(dev, oracle):
    """The schedule-less SpMM route the per-step values take, against the ordered CPU SpMM."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 900, 500, 9000
    edges = synthetic_interactions(U, I, E, seed=6)
    s = graph.ngcf_structure(edges, U + I).to(dev)
    val, _ = ops.edge_dropout_norm(s, 0.3, seed=9)
    x = np.random.default_rng(1).standard_normal((U + I, 64)).astype(np.float32)
    want = oracle.spmm((s.rowptr.cpu().numpy(), s.col.cpu().numpy(), val.cpu().numpy()), x)
    got = ops.spmm_raw(s.with_values(val), torch.from_numpy(x).to(dev))
    assert np.array_equal(got.cpu().numpy(), want)


def test_gemm_leaky_02_accumulate_bit_exact(dev, oracle):
    from chaorec_amd import ops
    rng = np.random.default_rng(8)
    A, B = rng.standard_normal((300, 64)).astype(np.float32), rng.standard_normal((64, 64)).astype(np.float32)
    C = rng.standard_normal((300, 64)).astype(np.float32)
    want = oracle.gemm(A, B, transB=True, C=C.copy(), act=2)
    out = torch.from_numpy(C.copy()).to(dev)
    ops.gemm_raw(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), transB=True, out=out, accumulate=True, act=2)
    assert np.array_equal(out.cpu().numpy(), want) and (want < 0).any()


@pytest.mark.parametrize("n,k", [(1000, 1), (1000, 1000), (50000, 40000), (3_000_000, 2_400_000)])
def test_weighted_sample_keep_is_the_k_smallest_keys(dev, oracle, n, k):
    from chaorec_amd import ops
    rng = np.random.default_rng(n)
    w = (rng.random(n) ** 2 + 1e-3).astype(np.float32)
    w[::97] = 0.0                                            # never selected
    if k == n:
        w[:] = np.maximum(w, 1e-3)
    wt = torch.from_numpy(w).to(dev)
    keep, keys = ops.weighted_sample_keep(wt, k, seed=11, step=4, return_keys=True)
    keep2 = ops.weighted_sample_keep(wt, k, seed=11, step_dev=torch.tensor([4], device=dev), step=0)
    keep, keys = keep.cpu().numpy().astype(bool), keys.cpu().numpy().view(np.uint64)
    assert np.array_equal(keep, keep2.cpu().numpy().astype(bool))          # deterministic, step == step_dev
    assert keep.sum() == k
    kth = np.partition(keys, k - 1)[k - 1]
    assert np.array_equal(keep, keys <= kth)                                 # exactly the k smallest keys
    assert not keep[w == 0].any()
    # the keys are the exponential race of the header: |log u| / w in the high word, hash bits in the low word
    u, low = oracle.race_uniform(n, 11, 4)
    pos = w > 0
    hi = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)[pos], low[pos])
    assert np.allclose(hi[pos], np.abs(np.log(u[pos].astype(np.float64))) / w[pos], rtol=1e-5)
    assert (keys[~pos] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    other = ops.weighted_sample_keep(wt, k, seed=12, step=4).cpu().numpy().astype(bool)
    assert k == n or not np.array_equal(other, keep)


def test_weighted_sample_inclusion_probabilities(dev):
    """Sampling 2 of 4 weights without replacement: the pair frequencies over 3000 independent draws (one `step`
    each) follow the sequential-draw law p(i then j) = w_i/W * w_j/(W - w_i), torch.multinomial's."""
    from chaorec_amd import ops
    w = np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32)
    wt = torch.from_numpy(w).to(dev)
    counts = np.zeros((4, 4))
    T = 3000
    for t in range(T):
        k = ops.weighted_sample_keep(wt, 2, seed=5, step=t).cpu().numpy().astype(bool)
        i, j = np.nonzero(k)[0]
        counts[i, j] += 1
    W = w.sum()
    for i in range(4):
        for j in range(i + 1, 4):
            pij = w[i] / W * w[j] / (W - w[i]) + w[j] / W * w[i] / (W - w[j])
            assert abs(counts[i, j] / T - pij) < 4 * np.sqrt(pij * (1 - pij) / T) + 1e-3, (i, j)


@pytest.mark.parametrize("n,D", [(1, 4), (1000, 64), (777, 128), (300, 20), (65, 384), (40, 1024)])
def test_row_cosine_scale_vs_oracle_and_torch_autograd(dev, oracle, n, D):
    """chaorec_row_cosine_scale_fwd/bwd against the fp64 restatement and against torch's own
    F.cosine_similarity + einsum autograd in fp64 (tolerance: fp32 rounding of 3 row reductions)."""
    from chaorec_amd import ops
    rng = np.random.default_rng(n + D)
    y = rng.standard_normal((n, D)).astype(np.float32)
    e = rng.standard_normal((n, D)).astype(np.float32)
    if n > 10:
        y[3] = 0.0            # |y| below eps: w = 0, and the clamp branch of the gradient
        e[5] = 0.0
        y[7] = e[7]           # w = 1
    go = rng.standard_normal((n, D)).astype(np.float32)
    ty = torch.from_numpy(y).to(dev).requires_grad_(True)
    te = torch.from_numpy(e).to(dev).requires_grad_(True)
    out = ops.row_cosine_scale(ty, te)
    out.backward(torch.from_numpy(go).to(dev))
    want, w = oracle.row_cosine_scale(y, e)
    assert np.allclose(out.detach().cpu().numpy(), want, rtol=2e-6, atol=1e-6)
    dy = torch.from_numpy(y).double().requires_grad_(True)
    de = torch.from_numpy(e).double().requires_grad_(True)
    ref = torch.einsum('a,ab->ab', torch.nn.functional.cosine_similarity(dy, de, dim=-1), dy)
    ref.backward(torch.from_numpy(go).double())
    assert np.allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-6)
    for got, r in ((ty.grad, dy.grad), (te.grad, de.grad)):
        scale = float(r.abs().max()) + 1e-12
        assert float((got.cpu().double() - r).abs().max()) <= 5e-6 * scale


@pytest.mark.parametrize("M,N", [(1, 1), (20000, 64), (513, 320), (70000, 7), (3, 1000)])
def test_col_sum_and_mean_all_deterministic_and_replay_safe(dev, M, N):
    """chaorec_colsum_f32 / chaorec_sum_f32: equal to fp64 sums to fp32 rounding, bit-identical run to run, and --
    the reason they exist -- still right on every replay of a captured graph when the input changes in between
    (torch's own multi-block x.sum(0) / x.mean() go stale from the second replay on, checked here too as a canary)."""
    from chaorec_amd import ops
    g0 = torch.Generator(device=dev)
    g0.manual_seed(M * 31 + N)
    x = torch.randn(M, N, device=dev, generator=g0)
    cs, ma = ops.col_sum(x), ops.mean_all(x)
    assert torch.equal(cs, ops.col_sum(x)) and torch.equal(ma, ops.mean_all(x))
    assert torch.allclose(cs.double(), x.double().sum(0), rtol=1e-5, atol=1e-4 * max(1.0, M ** 0.5))
    assert float(ma) == pytest.approx(float(x.double().mean()), abs=1e-5)
    xr = x.clone().requires_grad_(True)
    ops.mean_all(xr * xr).backward()
    assert torch.allclose(xr.grad, 2 * x / x.numel(), rtol=1e-6, atol=1e-12)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            ops.col_sum(x), ops.mean_all(x)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        c_out, m_out, t_out = ops.col_sum(x), ops.mean_all(x), x.sum(0)
    torch_stale = False
    for r in range(3):
        x.copy_(torch.randn(M, N, device=dev, generator=g0) * (r + 2))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(c_out, ops.col_sum(x)) and torch.equal(m_out, ops.mean_all(x)), r
        torch_stale |= not torch.allclose(t_out, x.sum(0), rtol=1e-4, atol=1e-3)
    if M >= 20000:
        # not an assertion on torch: if this starts passing the workaround may be dropped
        print("torch x.sum(0) stale under replay:", torch_stale)


@pytest.mark.parametrize("U,D,kind", [(40, 64, "ties"), (600, 64, "ties"), (70, 128, "hot_range")])
def test_score_topk_grouped_fallback_on_long_item_ranges(dev, oracle, U, D, kind):
    """Past 128 k items the users the prefilter cannot certify are ranked 32 at a time by the f32 MFMA sweep over a
    COMPACT user set (device-side queue -> user_map), the overflow of that set (> 512 users) by the per-user exact
    kernel.  Inputs that push every user (ties) or a strided subset (hot item range) into the queue must still give
    the exact top-K, written to the right rows."""
    from chaorec_amd import ops
    rng = np.random.default_rng(U + D)
    I, K = 140000, 50
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    if kind == "ties":
        ie[:] = ie[0]                       # every score of a user is the same number: lists overflow for everybody
    else:
        ie *= 0.05
        ie[40000:40000 + 3 * 32] = np.abs(rng.standard_normal((96, D))).astype(np.float32) * 3
        ue[::3] = np.abs(ue[::3])           # a third of the users see a hot, unsampled range: overflow
    hist = _hist_random(U, I, 30, seed=U)
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    st = {}
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K, id_offset=U,
                                  stats=st)
    assert np.array_equal(got_v.cpu().numpy(), want_v)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    if kind == "ties":
        assert st["fallback_users"] == U, st            # all of them went through the queue (U = 600: both routes)


@pytest.mark.parametrize("variant,joined", [(0, True), (1, False), (2, True)])
def test_bpr_fused_draw_equals_draw_then_bpr(dev, variant, joined):
    """chaorec_bpr_fwd_drawn_f32 (batch drawn inside the forward launch) against chaorec_draw_batch followed by
    chaorec_bpr_fwd_f32 for the same (seed, step, step_dev): identical ids, bit-identical loss parts and gradients."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, B = 3000, 1700, 30000, 64, 1024
    edges = synthetic_interactions(U, I, E, seed=3)
    hist = tuple(t.to(dev) for t in graph.user_hist_csr_from_edges(edges, U))
    ed = torch.from_numpy(edges.astype(np.int64)).to(dev)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(variant)
    tab = (torch.randn(U + I, D, device=dev, generator=g0) * 0.3)
    step_dev = torch.tensor([7], dtype=torch.int64, device=dev)

    def run(fused):
        t = tab.clone().requires_grad_(True)
        tu, ti, off = (t, None, U) if joined else (t[:U], t[U:], 0)
        if fused:
            out, users, pos, neg = ops.bpr_loss_drawn(tu, ti, ed, hist, B, U, I, 99, 5, variant, 1e-3, item_offset=off,
                                                      step_dev=step_dev)
        else:
            users, pos, neg = ops.draw_batch(ed, hist, B, U, I, 99, 5, step_dev=step_dev)
            out = ops.bpr_loss(tu, ti, users, pos, neg, variant, 1e-3, item_offset=off)
        out[0].backward()
        return users, pos, neg, out.detach().clone(), t.grad.clone()

    a, b = run(True), run(False)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert torch.equal(a[3], b[3])
    # (the backward scatters with fp32 atomics: equal up to the order of duplicate rows inside the batch)
    assert torch.allclose(a[4], b[4], rtol=1e-5, atol=1e-9)
    assert int(a[1].min()) >= 0 and int(a[1].max()) < I and int(a[2].max()) < I
    # advance=True moves the device counter on AFTER the draw: same ids as step_dev = 7, counter 8 afterwards
    ctr = torch.tensor([7], dtype=torch.int64, device=dev)
    _, u2, p2, n2 = ops.bpr_loss_drawn(tab, None, ed, hist, B, U, I, 99, 5, variant, 1e-3, item_offset=U, step_dev=ctr,
                                       advance=True)
    assert int(ctr) == 8 and torch.equal(u2, a[0]) and torch.equal(p2, a[1]) and torch.equal(n2, a[2])


@pytest.mark.parametrize("M,N,K,act,bias", [(11384, 64, 4096, 0, True), (1000, 64, 384, 1, True), (777, 100, 320, 2, False),
                                            (300, 256, 128, 0, True), (4097, 64, 70, 0, False), (128, 64, 8192, 0, True)])
def test_gemm_nt_bf16x3_fp32_grade(dev, M, N, K, act, bias):
    """The split-bf16 projection GEMM (three bf16 planes per fp32 operand, six MFMA products): error against an fp64
    product bounded by 1e-6 * sum_k |a||b| per element (stated tolerance; an fp32 chain's own bound is K * 6e-8 of
    that), and at least as close to it as the exact-chain f32 kernel is on the long reductions."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev, generator=g) * torch.exp(torch.randn(M, 1, device=dev, generator=g))
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g) if bias else None
    got = ops.gemm_nt_bf16x3(x, w, bias=b, act=act)
    ref = x.double() @ w.double().t()
    mass = x.double().abs() @ w.double().abs().t()
    if b is not None:
        ref = ref + b.double()
        mass = mass + b.double().abs()
    slope = {0: 1.0, 1: 0.01, 2: 0.2}[act]
    ref_act = torch.where(ref > 0, ref, ref * slope) if act else ref
    err = (got.double() - ref_act).abs()
    assert bool((err <= 1e-6 * mass + 1e-30).all()), float((err / (mass + 1e-30)).max())
    f32 = ops.gemm_raw(x, w, transB=True, bias=b, act=act)
    assert float(err.mean()) <= 1.5 * float((f32.double() - ref_act).abs().mean()) + 1e-12
    # deterministic
    assert torch.equal(got, ops.gemm_nt_bf16x3(x, w, bias=b, act=act))


@pytest.mark.parametrize("M,N,K", [(64, 64, 47001), (64, 768, 9000), (768, 768, 5000), (70, 130, 4099), (33, 61, 1234),
                                   (256, 128, 640), (128, 4096, 2048)])
def test_gemm_tn_bf16x3_fp32_grade(dev, M, N, K):
    """The split-bf16 weight-gradient GEMM gy^T x (reduction over the rows of both operands, slabs along K): error
    against an fp64 product bounded by 1e-6 * sum_k |a||b| per element, on average no worse than 1.5x the f32 kernel's,
    bit-identical run to run; ragged sizes take the scalar fetch path."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(M + N + K)
    gy = torch.randn(K, M, device=dev, generator=g) * torch.exp(torch.randn(K, 1, device=dev, generator=g))
    x = torch.randn(K, N, device=dev, generator=g) * 0.05
    got = ops.gemm_tn_bf16x3(gy, x)
    ref = gy.double().t() @ x.double()
    mass = gy.double().abs().t() @ x.double().abs()
    err = (got.double() - ref).abs()
    assert bool((err <= 1e-6 * mass + 1e-30).all()), float((err / (mass + 1e-30)).max())
    f32 = ops.gemm_raw(gy, x, transA=True)
    assert float(err.mean()) <= 1.5 * float((f32.double() - ref).abs().mean()) + 1e-12
    assert torch.equal(got, ops.gemm_tn_bf16x3(gy, x))


@pytest.mark.parametrize("M,N,K", [(47001, 64, 64), (9000, 320, 64), (5000, 768, 768), (4099, 130, 70), (1234, 61, 33),
                                   (640, 128, 256), (300, 4096, 64)])
def test_gemm_nn_bf16x3_fp32_grade(dev, M, N, K):
    """The split-bf16 input-gradient GEMM gy W (W [K = out, N = in] as nn.Linear stores it, no transposed copy): error
    against an fp64 product bounded by 1e-6 * sum_k |a||b| per element, on average no worse than 1.5x the f32 kernel's,
    bit-identical run to run; also reading / writing column slices of wider buffers in place."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(M + N + K)
    gy = torch.randn(M, K, device=dev, generator=g) * torch.exp(torch.randn(M, 1, device=dev, generator=g))
    w = torch.randn(K, N, device=dev, generator=g) * 0.05
    got = ops.gemm_nn_bf16x3(gy, w)
    ref = gy.double() @ w.double()
    mass = gy.double().abs() @ w.double().abs()
    err = (got.double() - ref).abs()
    assert bool((err <= 1e-6 * mass + 1e-30).all()), float((err / (mass + 1e-30)).max())
    f32 = ops.gemm_raw(gy, w)
    assert float(err.mean()) <= 1.5 * float((f32.double() - ref).abs().mean()) + 1e-12
    assert torch.equal(got, ops.gemm_nn_bf16x3(gy, w))
    if K % 4 == 0 and N % 4 == 0:
        wide_in = torch.full((M, K + 8), float("nan"), device=dev)
        wide_in[:, 4:4 + K] = gy
        wide_out = torch.full((M, N + 12), -7.0, device=dev)
        ops.gemm_nn_bf16x3(wide_in[:, 4:4 + K], w, out=wide_out[:, 8:8 + N])
        assert torch.equal(wide_out[:, 8:8 + N], got)
        assert bool((wide_out[:, :8] == -7.0).all()) and bool((wide_out[:, 8 + N:] == -7.0).all())


@pytest.mark.parametrize("M,K,N1,N2", [(60499, 64, 64, 64), (5000, 256, 256, 64), (777, 128, 64, 128), (300, 64, 4, 60)])
def test_gemm_bf16x3_dual_products(dev, M, K, N1, N2):
    """The two-segment GEMMs (two Linears over the same input as ONE product each way): the forward is bit-identical to the two
    separate products (the same k-ordered accumulation per output element); the input gradient [g1 | g2] [w1; w2] and the
    weight gradients [g1 | g2]^T x are fp32-grade against fp64 (other associations than two products + an add)."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(M + K + N1)
    x = torch.randn(M, K, device=dev, generator=g)
    w1, w2 = torch.randn(N1, K, device=dev, generator=g) * 0.1, torch.randn(N2, K, device=dev, generator=g) * 0.1
    b1, b2 = torch.randn(N1, device=dev, generator=g), torch.randn(N2, device=dev, generator=g)
    y1, y2 = ops.gemm_nt_bf16x3_dual(x, w1, w2, b1, b2, 0, 1)
    assert torch.equal(y1, ops.gemm_nt_bf16x3(x, w1, bias=b1, act=0)) and torch.equal(y2, ops.gemm_nt_bf16x3(x, w2, bias=b2, act=1))
    g1, g2 = torch.randn(M, N1, device=dev, generator=g), torch.randn(M, N2, device=dev, generator=g)
    gx = ops.gemm_nn_bf16x3_dual(g1, g2, w1, w2)
    ref = g1.double() @ w1.double() + g2.double() @ w2.double()
    mass = g1.double().abs() @ w1.double().abs() + g2.double().abs() @ w2.double().abs()
    assert bool(((gx.double() - ref).abs() <= 1e-6 * mass + 1e-30).all())
    d1, d2 = ops.gemm_tn_bf16x3_dual(g1, g2, x)
    for got, gg in ((d1, g1), (d2, g2)):
        ref = gg.double().t() @ x.double()
        mass = gg.double().abs().t() @ x.double().abs()
        assert bool(((got.double() - ref).abs() <= 1e-6 * mass + 1e-30).all())
    assert torch.equal(gx, ops.gemm_nn_bf16x3_dual(g1, g2, w1, w2))              # deterministic
    d1b, d2b = ops.gemm_tn_bf16x3_dual(g1, g2, x)
    assert torch.equal(d1, d1b) and torch.equal(d2, d2b)


def test_linear_forward_pipes_agree(dev):
    """ops.linear on either pipe: same autograd contract, outputs equal to fp32 rounding, identical backward kernels."""
    from chaorec_amd import ops
    torch.manual_seed(1)
    x = torch.randn(3000, 384, device=dev, requires_grad=True)
    lin = torch.nn.Linear(384, 64).to(dev)
    outs = {}
    for pipe in ("bf16x3", "f32"):
        ops.LINEAR_FORWARD = pipe
        x.grad = None
        lin.zero_grad()
        y = ops.linear(x, lin.weight, lin.bias, act=1)
        y.square().sum().backward()
        outs[pipe] = (y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    ops.LINEAR_FORWARD = "bf16x3"
    for a, b in zip(outs["bf16x3"], outs["f32"]):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()))


def test_ngcf_elementwise_backward_kernels_equal_torch(dev):
    """chaorec_leaky_bwd_f32 / chaorec_mul_pair_bwd_f32 against the torch expressions they replace, bit for bit
    (separately rounded product and sum), through ops.ngcf_layer's autograd and directly on the Linear's slope too."""
    from chaorec_amd import ops
    N, D = 4099, 64
    s = torch.randn(N, D, device=dev, requires_grad=True)
    x = torch.randn(N, D, device=dev, requires_grad=True)
    w1 = (torch.randn(D, D, device=dev) * 0.1).requires_grad_()
    w2 = (torch.randn(D, D, device=dev) * 0.1).requires_grad_()
    gy = torch.randn(N, D, device=dev)
    y = ops.ngcf_layer(s, x, w1, w2)
    y.backward(gy)
    # the same chain with torch's elementwise ops around the same GEMM launches
    g = torch.where(y.detach() > 0, gy, gy * 0.2)
    gs_ref = ops.gemm_raw(g, w1.detach()) + ops.gemm_raw(g, w2.detach()) * x.detach()
    gx_ref = ops.gemm_raw(g, w2.detach()) * s.detach()
    assert torch.equal(s.grad, gs_ref) and torch.equal(x.grad, gx_ref)
    assert torch.equal(w1.grad, ops.gemm_raw(g, s.detach(), transA=True))
    assert torch.equal(w2.grad, ops.gemm_raw(g, (s * x).detach(), transA=True))
    # forward value: leaky_relu_0.2 of the two products (fp32 association of the accumulate epilogue: tolerance)
    ref = torch.nn.functional.leaky_relu(s.detach() @ w1.detach().t() + (s * x).detach() @ w2.detach().t(), 0.2)
    assert torch.allclose(y.detach(), ref, rtol=1e-4, atol=1e-5)
