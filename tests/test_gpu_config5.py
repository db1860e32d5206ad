"""BASELINE.json configs[4] on one GPU: ONE rank's share of the synthetic 10 M x 2 M / 200 M-edge graph
(1.25 M users x 2 M items, 25 M interactions -> 50 M directed edges, dim 128, 3 layers; the 1.66 GB embedding table is
6.5x the Infinity Cache: the HBM-bound regime).  The oracle cannot restate the whole shard in seconds, so: SpMM rows
against the oracle (bit-exact, ordered sums) and an fp64 row reference on a row subset that includes the heaviest
rows; full-rank top-50 of sampled + heaviest-history users over all 2 M items against oracle.score_topk (bit-exact);
size-independent properties (linearity, symmetry <Ax, y> = <x, Ay>, determinism, sortedness, uniqueness, no history
items); the captured training step against the eager one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D, L = 128, 3


@pytest.fixture(scope="module")
def shard():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib, graph
    from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
    _lib.load()
    dev = torch.device("cuda:0")
    U, I, E = DATASET_SHAPES["config5_shard"]
    edges = synthetic_interactions(U, I, E, seed=42)
    csr_host = graph.lightgcn_csr(edges, U + I)
    hist = graph.user_hist_csr_from_edges(edges, U)
    g = torch.Generator().manual_seed(5)
    a = float(np.sqrt(6.0 / (U + I + D)))
    x = (torch.rand(U + I, D, generator=g) * 2 - 1) * a          # xavier_uniform-shaped table
    return dict(dev=dev, U=U, I=I, edges=edges, csr_host=csr_host, csr=csr_host.to(dev),
                hist_host=hist, hist=(hist[0].to(dev), hist[1].to(dev)), x_host=x, x=x.to(dev))


def test_spmm_rows_vs_oracle_and_fp64(shard, oracle):
    from chaorec_amd import ops
    csr, x = shard["csr"], shard["x"]
    y = ops.spmm_raw(csr, x)
    y2 = ops.spmm_raw(csr, x)
    assert torch.equal(y, y2)                                    # deterministic
    rp = shard["csr_host"].rowptr.numpy()
    col = shard["csr_host"].col.numpy()
    val = shard["csr_host"].val.numpy()
    deg = rp[1:] - rp[:-1]
    rng = np.random.default_rng(0)
    rows = np.unique(np.concatenate([rng.choice(len(deg), 4000, replace=False), np.argsort(deg)[-24:],
                                     np.nonzero(deg == 0)[0][:8]]))
    assert deg[rows].max() > 1000                                # the heavy tail is in the sample
    # sub-CSR of those rows over the full column range -> ordered sums by the oracle (C, -ffp-contract=off)
    sub_rp = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(deg[rows], out=sub_rp[1:])
    take = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
    xh = shard["x_host"].numpy()
    want = oracle.spmm((sub_rp, col[take], val[take]), xh)
    got = y[torch.from_numpy(rows).to(y.device)].cpu().numpy()
    assert np.array_equal(got, want)
    # fp64 row reference
    # fp64 row reference, with the rigorous bound of a sequential fp32 sum: (n + 1) u sum_e |val_e x_e|
    ref = np.zeros((len(rows), D))
    mass = np.zeros((len(rows), D))
    for k in range(len(rows)):
        s, e = sub_rp[k], sub_rp[k + 1]
        terms = val[take[s:e]].astype(np.float64)[:, None] * xh[col[take[s:e]]].astype(np.float64)
        ref[k], mass[k] = terms.sum(0), np.abs(terms).sum(0)
    bound = (deg[rows][:, None] + 1) * 2.0 ** -24 * mass
    assert np.all(np.abs(got - ref) <= bound + 1e-30)
    light = deg[rows] <= 64
    assert np.abs(got - ref)[light].max() <= 2e-6 * np.abs(ref)[light].max()
    # linearity and symmetry on the whole shard (A is symmetric: <Ax, z> = <x, Az>)
    gz = torch.Generator(device=x.device).manual_seed(7)          # (seeded: the bound below is tight for the hub rows)
    z = torch.randn(x.shape, generator=gz, device=x.device) * 0.01
    yz = ops.spmm_raw(csr, z)
    lin = ops.spmm_raw(csr, x + z)
    # three ordered fp32 sums of up to 71 k terms each + the rounding of x + z: a few 1e-6 of the row scale
    assert float((lin - (y + yz)).abs().max()) <= 5e-6 * float(lin.abs().max())
    lhs, rhs = float((y.double() * z.double()).sum()), float((x.double() * yz.double()).sum())
    assert lhs == pytest.approx(rhs, rel=1e-6)
    # layer-mean epilogue == the reference's accumulation, on the sampled rows
    acc = torch.empty_like(x)
    ops.spmm_raw(csr, x, acc=acc, acc_init=x, acc_w=0.25)
    w = np.float32(0.25)
    assert np.array_equal(acc[torch.from_numpy(rows).to(acc.device)].cpu().numpy(), (w * xh[rows]) + (w * got))


def test_full_rank_vs_oracle(shard, oracle):
    from chaorec_amd import ops
    U, I, dev = shard["U"], shard["I"], shard["dev"]
    # a propagated table (scores with structure), as gene_ranklist sees it
    res = ops.layer_mean_propagate(shard["x"], shard["csr"], L)
    st = {}
    idx, val = ops.score_topk(res[:U], res[U:], shard["hist"], 1e-6, 50, id_offset=U, stats=st)
    assert st["prefilter_users"] == U                            # the bf16-prefilter route ran
    rp, hc = shard["hist_host"][0].numpy(), shard["hist_host"][1].numpy()
    hdeg = rp[1:] - rp[:-1]
    rng = np.random.default_rng(1)
    users = np.unique(np.concatenate([rng.choice(U, 320, replace=False), np.argsort(hdeg)[-16:]]))
    sub_rp = np.zeros(len(users) + 1, np.int64)
    np.cumsum(hdeg[users], out=sub_rp[1:])
    sub_col = np.concatenate([hc[rp[u]:rp[u + 1]] for u in users]).astype(np.int32)
    res_h = res.cpu().numpy()
    wi, wv = oracle.score_topk(res_h[:U][users], res_h[U:], (sub_rp, sub_col), 1e-6, 50, U)
    sel = torch.from_numpy(users).to(dev)
    assert np.array_equal(idx[sel].cpu().numpy(), wi)
    assert np.array_equal(val[sel].cpu().numpy(), wv)
    # properties over ALL 1.25 M users: sorted values, distinct items in range, no history item unless it ranks on 1e-6
    assert bool((val[:, 1:] <= val[:, :-1]).all())
    assert int(idx.min()) >= U and int(idx.max()) < U + I
    srt = idx.sort(1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    loc = (idx - U).to(torch.int64)
    hr, hcd = shard["hist"]
    for u in users[:64]:
        mine = set(loc[u].tolist())
        seen = set(hcd[hr[u]:hr[u + 1]].tolist())
        masked = [k for k in range(50) if int(loc[u, k]) in seen]
        assert all(float(val[u, k]) == np.float32(1e-6) for k in masked)
        assert len(mine) == 50
    assert torch.equal(idx, ops.score_topk(res[:U], res[U:], shard["hist"], 1e-6, 50, id_offset=U)[0])


def test_captured_step_equals_eager(shard):
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    U, I, dev, edges = shard["U"], shard["I"], shard["dev"], shard["edges"]

    class _NoDict(dict):          # user_item_dict of 1.25 M python lists is not needed: the history CSR is shared
        pass

    def make():
        m = LightGCN.__new__(LightGCN)
        torch.nn.Module.__init__(m)
        m.result, m.device, m.num_user, m.num_item = None, dev, U, I
        m.aggr_mode, m.user_item_dict, m.reg_weight, m.dim_embedding, m.n_layers = "add", _NoDict(), 1e-3, D, L
        m.edge_index = None
        m.graph, m.hist = shard["csr"], shard["hist"]
        m.user_embedding, m.item_embedding = torch.nn.Embedding(U, D), torch.nn.Embedding(I, D)
        with torch.no_grad():
            m.user_embedding.weight.copy_(shard["x_host"][:U])
            m.item_embedding.weight.copy_(shard["x_host"][U:])
        m._flat = None
        m._join_tables()
        m = m.to(dev)
        return m, FusedAdam(m.parameters(), lr=1e-3)

    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    eager, oe = make()
    cap, oc = make()
    counter = torch.zeros(1, dtype=torch.int64, device=dev)

    def drawn():
        return cap.loss_drawn(edges_dev, 1024, 42, 0, step_dev=counter, advance=True)

    step = GraphedTrainStep(cap, oc, batch_fn=lambda: (), loss_fn=drawn)
    counter.zero_()
    for it in range(6):
        oe.zero_grad()
        le = eager.loss_drawn(edges_dev, 1024, 42, it)
        le.backward()
        oe.step()
        lc = step()
        assert float(lc) == pytest.approx(float(le.detach()), rel=1e-5), it
    assert torch.allclose(cap.user_embedding.weight, eager.user_embedding.weight, rtol=0, atol=2e-6)
