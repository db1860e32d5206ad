"""Pins the CPU oracle (oracle/) against outputs of the reference's own classes
(tests/golden/*.npz, produced by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, tie_aware_rank_equal


def test_csr_equals_edge_scatter(oracle):
    g = load_golden("lightgcn_tiny.npz")
    N = int(g["U"]) + int(g["I"])
    src, dst = oracle.bidirectional_edges(g["edges"])
    w = oracle.sym_norm_weights(src, dst, N)
    csr = oracle.csr_from_edges(src, dst, w, N)
    a = oracle.spmm(csr, g["x0"])
    b = oracle.scatter_edges(src, dst, w, g["x0"], N)
    assert np.array_equal(a, b)
    # isolated item 4 (global node 10) stays exactly zero
    assert np.all(a[10] == 0)


def test_lightgcn_tiny_forward_bit_exact(oracle):
    g = load_golden("lightgcn_tiny.npz")
    N = int(g["U"]) + int(g["I"])
    csr = oracle.lightgcn_csr(g["edges"], N)
    final, layers = oracle.lightgcn_forward(g["x0"], csr, int(g["L"]))
    for l, x in enumerate(layers):
        assert np.array_equal(x, g["layers"][l]), f"layer {l}"
    assert np.array_equal(final, g["result"])


def test_lightgcn_tiny_loss_grad_rank(oracle):
    g = load_golden("lightgcn_tiny.npz")
    U, I = int(g["U"]), int(g["I"])
    csr = oracle.lightgcn_csr(g["edges"], U + I)
    out, grad = oracle.lightgcn_loss(g["x0"], csr, int(g["L"]), U, g["users"], g["pos"] - U, g["neg"] - U,
                                     float(g["reg"]))
    assert out[0] == pytest.approx(float(g["loss"]), rel=1e-6)
    assert out[1] == pytest.approx(float(g["bpr"]), rel=1e-6)
    assert out[2] == pytest.approx(float(g["reg_loss"]), rel=1e-5)
    ref = np.concatenate([g["g_user"], g["g_item"]], 0)
    assert np.allclose(grad, ref, rtol=1e-4, atol=1e-7)
    hist = oracle.user_hist_csr(g["edges"], U)
    idx, val = oracle.gene_ranklist(g["result"], U, I, hist, 1e-6, int(g["topk"]))
    ok, why = tie_aware_rank_equal(idx, val, g["rank"], g["rank_val"], rtol=1e-5, atol=1e-7)
    assert ok, why


def test_lightgcn_baby(oracle, baby):
    g = load_golden("lightgcn_baby.npz")
    U, I, D, L = baby["U"], baby["I"], int(g["D"]), int(g["L"])
    x0 = (np.random.default_rng(int(g["x0_seed"])).standard_normal((U + I, D)) * float(g["x0_scale"])).astype(np.float32)
    csr = oracle.lightgcn_csr(baby["train"], U + I)
    final, layers = oracle.lightgcn_forward(x0, csr, L)
    rows = g["rows"]
    # torch's CPU scatter_add_ order is the edge order: the restatement is bit-exact on real data too
    for l in range(L + 1):
        assert np.array_equal(layers[l][rows], g["layer_rows"][l]), f"layer {l}"
    assert np.array_equal(final[rows], g["result_rows"])
    assert final.astype(np.float64).sum() == pytest.approx(float(g["result_sum"]), rel=1e-9)
    out, grad = oracle.lightgcn_loss(x0, csr, L, U, g["users"], g["pos"] - U, g["neg"] - U, float(g["reg"]))
    assert out[0] == pytest.approx(float(g["loss"]), rel=2e-6)
    assert np.allclose(grad[rows], g["g_rows"], rtol=2e-4, atol=1e-9)
    assert np.abs(grad).sum() == pytest.approx(float(g["g_abs_sum"]), rel=1e-4)
    # ranking on a user subset
    hist = oracle.user_hist_csr(baby["train"], U)
    urows = g["urows"]
    sub_rowptr = np.zeros(len(urows) + 1, np.int64)
    cols = []
    for k, u in enumerate(urows):
        c = hist[1][hist[0][u]:hist[0][u + 1]]
        cols.append(c)
        sub_rowptr[k + 1] = sub_rowptr[k] + len(c)
    sub_hist = (sub_rowptr, np.concatenate(cols).astype(np.int32))
    idx, val = oracle.score_topk(final[:U][urows], final[U:], sub_hist, 1e-6, 50, U)
    ok, why = tie_aware_rank_equal(idx, val, g["rank_rows"].astype(np.int64), g["rank_val_rows"], rtol=2e-5, atol=1e-8)
    assert ok, why


def test_metrics_pinned(oracle, baby):
    g = load_golden("metrics_baby_fixed_rank.npz")
    U, I = baby["U"], baby["I"]
    fixed_rank = np.stack([np.random.default_rng(1000 + u).permutation(I)[:50] + U for u in range(U)])
    k_list = [int(k) for k in g["k_list"]]
    m = oracle.gene_metrics(baby["val"], fixed_rank, k_list)
    got = np.array([[m[k][n] for n in g["metric_names"]] for k in k_list])
    assert np.allclose(got, g["val_metrics"], rtol=1e-12, atol=0)


def test_metrics_of_reference_ranklist(oracle, baby):
    """Recall/NDCG of the oracle's rank list vs the reference's numbers on the reference's list."""
    g = load_golden("lightgcn_baby.npz")
    U, I, D, L = baby["U"], baby["I"], int(g["D"]), int(g["L"])
    x0 = (np.random.default_rng(int(g["x0_seed"])).standard_normal((U + I, D)) * float(g["x0_scale"])).astype(np.float32)
    csr = oracle.lightgcn_csr(baby["train"], U + I)
    final, _ = oracle.lightgcn_forward(x0, csr, L)
    hist = oracle.user_hist_csr(baby["train"], U)
    idx, _ = oracle.gene_ranklist(final, U, I, hist, 1e-6, 50)
    k_list = [int(k) for k in g["k_list"]]
    names = list(g["metric_names"])
    for split, key in ((baby["val"], "val_metrics"), (baby["test"], "test_metrics")):
        m = oracle.gene_metrics(split, idx, k_list)
        got = np.array([[m[k][n] for n in names] for k in k_list])
        assert np.abs(got - g[key]).max() < 1e-4  # north_star tolerance on Recall/NDCG@20


def test_user_item_dict_rule(oracle, baby):
    d = oracle.user_item_dict_from_edges(baby["train"])
    assert list(d.keys()) == sorted(d.keys()) and len(d) == baby["U"]
    rowptr, col = oracle.user_hist_csr(baby["train"], baby["U"])
    for u in (0, 1, 777, baby["U"] - 1):
        assert sorted(i - baby["U"] for i in d[u]) == col[rowptr[u]:rowptr[u + 1]].tolist()


def test_sampler_matches_reference_support(oracle):
    """dataload.py:74-79: negatives are uniform over the items the user never touched."""
    g = load_golden("sampler_tiny.npz")
    U, I = int(g["U"]), int(g["I"])
    hist = oracle.user_hist_csr(g["edges"], U)
    ref_hist = g["neg_hist"]
    users = np.repeat(np.arange(U), 4000).astype(np.int64)
    neg = oracle.sample_negatives(hist, users, I, seed=42, step=0, id_offset=U) - U
    mine = np.zeros((U, I), np.int64)
    np.add.at(mine, (users, neg), 1)
    assert np.array_equal(mine > 0, ref_hist > 0)  # same support as the reference sampler
    for u in range(U):
        allowed = mine[u] > 0
        exp = 4000 / allowed.sum()
        chi2 = ((mine[u][allowed] - exp) ** 2 / exp).sum()
        assert chi2 < 30, (u, chi2)  # dof <= 4: p(chi2 > 30) ~ 5e-6


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_ngcf_oracle_vs_reference_golden(oracle, tag):
    """The edge-wise NGCF restatement against the reference model's own output (tests/golden/gen_golden.py:gen_ngcf;
    the dropout masks the reference drew are inputs of the fixture)."""
    g = load_golden(f"ngcf_small_{tag}.npz")
    U, I, L = int(g["U"]), int(g["I"]), int(g["L"])
    x0 = np.concatenate([g["p_user_embedding.weight"], g["p_item_embedding.weight"]])
    w1 = [g[f"p_conv_layers.{l}.W1.weight"] for l in range(L)]
    w2 = [g[f"p_conv_layers.{l}.W2.weight"] for l in range(L)]
    masks = g["keep_masks"] if tag == "drop" else None
    out = oracle.ngcf_forward(x0, w1, w2, g["edges"], U + I, masks)
    assert np.allclose(out, g["result"], rtol=1e-6, atol=1e-7)
    terms, _ = oracle.bpr_fwd(out[:U], out[U:], g["users"], g["pos"] - U, g["neg"] - U, 0, float(g["reg"]))
    assert float(terms[0]) == pytest.approx(float(g["loss"]), rel=2e-6)
    if tag == "drop":
        assert masks.shape == (L, 2 * len(g["edges"])) and 0.5 < masks.mean() < 0.9


def test_ordered_bpr_backward_restatement(oracle):
    """oracle_bpr_bwd_ordered_f32 (the float, role-major, batch-ordered index_add the product's ordered backward launch is held to
    bit for bit on the GPU): a plain numpy walk in the same order gives the same bits; the double-precision index_add of
    oracle_bpr_bwd_f32 (pinned by the reference's golden gradients above) agrees to float accuracy; a joined table (items as
    rows item_offset.. of the one table) is the two-table result, row for row."""
    rng = np.random.default_rng(3)
    U, I, D, B = 23, 17, 64, 400
    tu = (rng.standard_normal((U, D)) * 0.3).astype(np.float32)
    ti = (rng.standard_normal((I, D)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    _, coef64 = oracle.bpr_fwd(tu, ti, users, pos, neg, 0, 1e-3)
    coef = coef64.astype(np.float32)
    go = np.float32(0.7)
    g_u, g_i = oracle.bpr_bwd_ordered(tu, ti, users, pos, neg, coef, 1e-3, grad_out=go)
    r2 = np.float32(np.float32(2.0) * np.float32(1e-3) / (np.float32(B) * np.float32(D))) * go
    wu, wi = np.zeros_like(tu), np.zeros_like(ti)
    for role in range(3):
        for b in range(B):
            c = coef[b] * np.float32(go * np.float32(1.0))
            u, p, n = tu[users[b]], ti[pos[b]], ti[neg[b]]
            if role == 0:
                wu[users[b]] += c * (p - n) + r2 * u
            elif role == 1:
                wi[pos[b]] += c * u + r2 * p
            else:
                wi[neg[b]] += -c * u + r2 * n
    assert np.array_equal(g_u, wu) and np.array_equal(g_i, wi)
    d_u, d_i = oracle.bpr_bwd(tu, ti, users, pos, neg, coef64, 1e-3, grad_out=0.7)
    assert np.allclose(g_u, d_u, rtol=2e-5, atol=1e-7) and np.allclose(g_i, d_i, rtol=2e-5, atol=1e-7)
    joined, _ = oracle.bpr_bwd_ordered(np.concatenate([tu, ti]), None, users, pos, neg, coef, 1e-3, grad_out=go, item_offset=U)
    assert np.array_equal(joined[:U], g_u) and np.array_equal(joined[U:], g_i)
