"""The CPU oracle against the reference's outputs on the REAL sports graph and over a 10-step training trajectory
(tests/golden/gen_fullsize.py; BASELINE.json configs[1]).  CPU only, seconds."""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_interactions, tie_aware_rank_equal


def seeded_lightgcn_tables(U, I, D, seed):
    """The reference constructor's initialisation (Model/LightGCN.py:67-70) under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    ue, ie = torch.nn.Embedding(U, D), torch.nn.Embedding(I, D)
    torch.nn.init.xavier_uniform_(ue.weight)
    torch.nn.init.xavier_uniform_(ie.weight)
    return torch.cat((ue.weight, ie.weight), 0).detach().numpy().copy()


def sub_hist(hist, urows):
    rowptr = np.zeros(len(urows) + 1, np.int64)
    cols = []
    for k, u in enumerate(urows):
        c = hist[1][hist[0][u]:hist[0][u + 1]]
        cols.append(c)
        rowptr[k + 1] = rowptr[k] + len(c)
    return rowptr, np.concatenate(cols).astype(np.int32)


def metrics_table(oracle, data, rank, g):
    k_list = [int(k) for k in g["k_list"]]
    m = oracle.gene_metrics(data, rank, k_list)
    return np.array([[m[k][n] for n in g["metric_names"]] for k in k_list])


def test_lightgcn_sports_oracle_vs_reference(oracle):
    g = load_golden("lightgcn_sports.npz")
    d = load_interactions("sports")
    U, I, D, L = d["U"], d["I"], int(g["D"]), int(g["L"])
    x0 = seeded_lightgcn_tables(U, I, D, int(g["init_seed"]))
    rows = g["rows"]
    assert np.array_equal(x0[rows], g["x0_rows"])
    assert x0.astype(np.float64).sum() == pytest.approx(float(g["x0_sum"]), rel=1e-12)
    csr = oracle.lightgcn_csr(d["train"], U + I)
    final, layers = oracle.lightgcn_forward(x0, csr, L)
    for l in range(L + 1):
        assert np.array_equal(layers[l][rows], g["layer_rows"][l]), f"layer {l}"
    assert np.array_equal(final[rows], g["result_rows"])
    out, grad = oracle.lightgcn_loss(x0, csr, L, U, g["users"], g["pos"] - U, g["neg"] - U, float(g["reg"]))
    assert out[0] == pytest.approx(float(g["loss"]), rel=2e-6)
    assert out[1] == pytest.approx(float(g["bpr"]), rel=2e-6)
    assert out[2] == pytest.approx(float(g["reg_loss"]), rel=1e-5)
    assert np.allclose(grad[rows], g["g_rows"], rtol=2e-4, atol=1e-10)
    assert np.abs(grad).sum() == pytest.approx(float(g["g_abs_sum"]), rel=1e-4)
    hist = oracle.user_hist_csr(d["train"], U)
    urows = g["urows"]
    idx, val = oracle.score_topk(final[:U][urows], final[U:], sub_hist(hist, urows), 1e-6, 50, U)
    ok, why = tie_aware_rank_equal(idx, val, g["rank_rows"].astype(np.int64), g["rank_val_rows"], rtol=2e-5, atol=1e-9)
    assert ok, why


@pytest.mark.parametrize("name", ["baby", "sports"])
def test_training_trajectory_oracle_vs_reference(oracle, name):
    """T Adam steps of the reference loop restated with the oracle's pieces (forward, BPR, ordered backward SpMM,
    oracle_adam_step), then the evaluation on the stale result (Q4)."""
    g = load_golden(f"lightgcn_trajectory_{name}.npz")
    d = load_interactions(name)
    U, I, D, L, T = d["U"], d["I"], int(g["D"]), int(g["L"]), int(g["T"])
    x = seeded_lightgcn_tables(U, I, D, int(g["init_seed"]))
    csr = oracle.lightgcn_csr(d["train"], U + I)
    m, v = np.zeros_like(x), np.zeros_like(x)
    final = None
    for t in range(T):
        users, pos, neg = (g["batches"][t, k].astype(np.int64) for k in range(3))
        final, _ = oracle.lightgcn_forward(x, csr, L)
        out, grad = oracle.lightgcn_loss(x, csr, L, U, users, pos - U, neg - U, float(g["reg"]))
        assert out[0] == pytest.approx(float(g["losses"][t]), rel=5e-6), t
        oracle.adam_step(x, np.ascontiguousarray(grad, np.float32), m, v, float(g["lr"]), 0.9, 0.999, 1e-8, 0.0, t + 1)
    rows = g["rows"]
    # Adam divides by sqrt(v) + 1e-8 with v ~ g^2: an entry whose gradient is ~1e-8 moves by a visibly different
    # amount for a 1e-4 relative difference in g; everything else agrees to the last digits
    assert np.allclose(x[rows], g["weight_rows"], rtol=0, atol=2e-5)
    assert np.abs(x[rows] - g["weight_rows"]).mean() < 2e-7
    assert np.allclose(final[rows], g["result_rows"], rtol=0, atol=1e-5)
    hist = oracle.user_hist_csr(d["train"], U)
    idx, _ = oracle.gene_ranklist(final, U, I, hist, 1e-6, 50)
    for split, key in ((d["val"], "val_metrics"), (d["test"], "test_metrics")):
        assert np.abs(metrics_table(oracle, split, idx, g) - g[key]).max() < 1e-4     # north_star tolerance
