"""Round 6 (GPU): the ORDERED BPR backward -- atomic-free, bit-exact against the oracle's float index_add, reproducible run to
run -- and what it makes checkable: a training step that is the same bits every time, on one stream or two."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("D,B", [(64, 300), (128, 257), (16, 64), (256, 100), (64, 4096)])
@pytest.mark.parametrize("variant", [0, 2])
def test_bpr_ordered_backward_is_the_oracles_float_index_add_bit_for_bit(dev, oracle, D, B, variant):
    """chaorec_bpr_bwd_ordered_f32 against oracle_bpr_bwd_ordered_f32 (role-major, batch order, every addend rounded to float --
    one running sum per row over the gradients of emb[users] / emb[pos] / emb[neg], Model/LightGCN.py:113-121): the same bits, with heavy
    duplication inside the batch (dozens of addends per row), a non-unit grad_out and the L2 term; five repetitions give the
    same bits again (the atomic launch does not promise that)."""
    from chaorec_amd import ops
    assert ops.BPR_ORDERED
    rng = np.random.default_rng(D + B + variant)
    U, I = 37, 23
    tu = (rng.standard_normal((U, D)) * 0.3).astype(np.float32)
    ti = (rng.standard_normal((I, D)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    reg = 1e-3 if variant == 0 else 0.0
    ids = [torch.from_numpy(t).to(dev) for t in (users, pos, neg)]
    first = None
    for rep in range(5):
        a = torch.from_numpy(tu).to(dev).requires_grad_(True)
        b = torch.from_numpy(ti).to(dev).requires_grad_(True)
        res = ops.bpr_loss(a, b, *ids, variant, reg)
        (res[0] * 0.7).backward()
        got = (a.grad.cpu().numpy(), b.grad.cpu().numpy())
        if first is None:
            first = got
        else:
            assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), rep
    # the coefficients the forward launch leaves (float, on the device) feed the oracle's backward: the backward alone is compared
    a = torch.from_numpy(tu).to(dev).requires_grad_(True)
    b = torch.from_numpy(ti).to(dev).requires_grad_(True)
    res = ops.bpr_loss(a, b, *ids, variant, reg)
    coef = res.loss.grad_fn.saved_tensors[5].cpu().numpy()
    g_u, g_i = oracle.bpr_bwd_ordered(tu, ti, users, pos, neg, coef, reg, grad_out=np.float32(0.7))
    assert np.array_equal(first[0], g_u)
    assert np.array_equal(first[1], g_i)
    # ... and the double-precision oracle within the tolerance the atomic launch is held to
    out, coef64 = oracle.bpr_fwd(tu, ti, users, pos, neg, variant, reg)
    g_u64, g_i64 = oracle.bpr_bwd(tu, ti, users, pos, neg, coef64, reg, grad_out=0.7)
    assert np.allclose(first[0], g_u64, rtol=2e-5, atol=1e-7) and np.allclose(first[1], g_i64, rtol=2e-5, atol=1e-7)


def test_bpr_ordered_backward_on_one_joined_table(dev, oracle):
    """tab_i = None (LightGCN / MMGCN: users and items are rows of ONE table, one gradient buffer): the user and item slots
    are one group of 3 B slots in the ordered launch."""
    from chaorec_amd import ops
    rng = np.random.default_rng(5)
    U, I, D, B = 31, 29, 64, 500
    tab = (rng.standard_normal((U + I, D)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    t = torch.from_numpy(tab).to(dev).requires_grad_(True)
    res = ops.bpr_loss(t, None, *(torch.from_numpy(x).to(dev) for x in (users, pos, neg)), 0, 1e-3, item_offset=U)
    coef = res.loss.grad_fn.saved_tensors[5].cpu().numpy()
    res[0].backward()
    g, _ = oracle.bpr_bwd_ordered(tab, None, users, pos, neg, coef, 1e-3, item_offset=U)
    assert np.array_equal(t.grad.cpu().numpy(), g)


def test_bpr_multi_ordered_backward_equals_the_atomic_sums_and_repeats(dev, monkeypatch):
    """chaorec_bpr_multi_bwd_ordered_f32 (FREEDOM's three terms, two of them over gathered row blocks whose gradient is also
    scattered into the full table's row gradient): the atomic launch's sums up to the order of the additions, the same bits on
    every repetition."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    U, I, D, B = 60, 40, 64, 700
    tab_u = torch.randn(U, D, device=dev, generator=g) * 0.2
    users = torch.randint(0, U, (B,), device=dev, generator=g)
    wvec = torch.tensor([1.0, 1e-3, 1e-3], device=dev)

    def run(ordered):
        monkeypatch.setattr(ops, "BPR_ORDERED", ordered)
        tu = tab_u.clone().requires_grad_(True)
        gg = torch.Generator(device=dev)
        gg.manual_seed(12)
        terms, leaves = [], []
        for rows in (I, 2 * B, 2 * B):
            t = (torch.randn(rows, D, device=dev, generator=gg) * 0.1).requires_grad_(True)
            leaves.append(t)
            terms.append((t, torch.randint(0, rows, (B,), device=dev, generator=gg), torch.randint(0, rows, (B,), device=dev, generator=gg)))
        loss = ops.bpr_loss_multi(tu, users, ops.VARIANT_LOGSIGMOID, terms, wvec)
        (loss * 1.5).backward()
        return [tu.grad.clone()] + [t.grad.clone() for t in leaves]

    ref = run(False)
    first = run(True)
    for a, b in zip(first, ref):
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * float(b.abs().max()) + 1e-12)
    for _ in range(4):
        again = run(True)
        assert all(torch.equal(a, b) for a, b in zip(again, first))


def _mmgcn_six_steps(dev, streams, sharded=False):
    from chaorec_amd import graph, ops
    from chaorec_amd.Model import MMGCN
    import importlib
    mm = importlib.import_module("chaorec_amd.Model.MMGCN")
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 6000, 2500, 40000, 512
    edges = synthetic_interactions(U, I, E, seed=3)
    uid = graph.user_item_dict_from_edges(edges)
    g = torch.Generator().manual_seed(4)
    v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 256, generator=g)
    old = mm.BRANCH_STREAMS
    mm.BRANCH_STREAMS = streams
    try:
        torch.manual_seed(21)
        m = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-3)
        edges_dev = torch.from_numpy(np.stack([edges[:, 0], edges[:, 1]], 1).astype(np.int64)).to(dev)
        counter = torch.zeros(1, dtype=torch.int64, device=dev)

        def draw():
            counter.add_(1)
            u, pos, neg = ops.draw_batch(edges_dev, m.hist, B, U, I, 42, 0, step_dev=counter, item_offset=U)
            return torch.stack((u, u), 1), torch.stack((pos, neg), 1)

        step = GraphedTrainStep(m, opt, batch_fn=draw)
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        return {n: p.detach().clone() for n, p in m.named_parameters()}
    finally:
        mm.BRANCH_STREAMS = old


def test_mmgcn_steps_are_the_same_bits_every_run_on_one_stream_and_on_two(dev):
    """VERDICT r5 #3.  Six captured MMGCN training steps, repeated: EVERY parameter the same bits -- run against run on one
    stream (the round-5 code was not: the fp32 atomic adds of the BPR backward are applied in an order that moves with the load
    on the chip, tools/stream_stress.py) and the two-stream step (visual branch on a side stream) against the one-stream step."""
    ref = _mmgcn_six_steps(dev, False)
    for rep in range(3):
        for streams in (False, True):
            got = _mmgcn_six_steps(dev, streams)
            for n in ref:
                assert torch.equal(got[n], ref[n]), (rep, streams, n, float((got[n] - ref[n]).abs().max()))


@pytest.mark.parametrize("n_rows", [100_003, 262_147, 1_048_575 + 33])
def test_frontier_pack_prefix_over_several_blocks_of_words(dev, n_rows):
    """the multi-workgroup form of the frontier prefix (csrc/exchange.hip bits_prefix_kernel: a workgroup scans 1 024
    bitmap words, later workgroups first sum the words before theirs) on bitmaps that straddle several blocks, with
    n_rows % 32 != 0 and stray bits past the end: prefix[] against numpy's popcount / cumsum, the packed rows in bitmap order,
    the inverse copy."""
    from chaorec_amd import ops
    rng = np.random.default_rng(n_rows)
    D = 8
    n_words = (n_rows + 31) // 32
    rows = np.unique(rng.integers(0, n_rows, n_rows // 7))
    # dense stretches and empty stretches, so that whole 1 024-word blocks are all-ones / all-zeros
    rows = np.unique(np.concatenate([rows, np.arange(40_000, 40_000 + 70_000) % n_rows]))
    rows = rows[(rows < 5_000) | (rows > 38_000)]
    words = np.zeros(n_words + 1, dtype=np.uint32)
    np.bitwise_or.at(words, rows >> 5, np.uint32(1) << (rows & 31).astype(np.uint32))
    stray = words.copy()
    if n_rows % 32:
        stray[n_words - 1] |= np.uint32(0xFFFFFFFF) << np.uint32(n_rows % 32)          # every bit past the last row
    bits = torch.from_numpy(stray.view(np.int32)).to(dev)
    tab = torch.randn(n_rows, D, device=dev)
    compact = torch.full((len(rows) + 5, D), float("nan"), device=dev)
    prefix = torch.zeros(n_words + 1, dtype=torch.int32, device=dev)
    ops.frontier_pack(tab, bits, prefix, compact)
    pop = np.array([bin(int(w)).count("1") for w in words[:n_words]], dtype=np.int64)
    want_prefix = np.concatenate([[0], np.cumsum(pop)])
    assert np.array_equal(prefix.cpu().numpy().astype(np.int64), want_prefix)
    r = torch.from_numpy(rows).to(dev)
    assert torch.equal(compact[:len(rows)], tab[r]) and float(compact[len(rows):].abs().max()) == 0.0
    back = torch.zeros(n_rows, D, device=dev)
    ops.frontier_unpack(back, bits, prefix, compact)
    rest = torch.ones(n_rows, dtype=torch.bool, device=dev)
    rest[r] = False
    assert torch.equal(back[r], tab[r]) and float(back[rest].abs().max()) == 0.0


def test_edge_dot_refuses_wrong_index_types(dev):
    """chaorec_edge_dot_f32 reads 4-byte indices -- an int64 tensor must be an error, not pairs of int32."""
    from chaorec_amd import ops
    a, b = torch.randn(10, 8, device=dev), torch.randn(12, 8, device=dev)
    er, col = torch.tensor([0, 3, 9], dtype=torch.int32, device=dev), torch.tensor([1, 2, 11], dtype=torch.int32, device=dev)
    got = ops.edge_dot_raw(er, col, a, b)
    assert torch.allclose(got, (a[er.long()] * b[col.long()]).sum(1), rtol=1e-6, atol=1e-6)
    with pytest.raises(TypeError, match="int32"):
        ops.edge_dot_raw(er.long(), col, a, b)
    with pytest.raises(TypeError, match="int32"):
        ops.edge_dot_raw(er, col.long(), a, b)
    with pytest.raises(ValueError, match="n_entries"):
        ops.edge_dot_raw(er, col, a, b, n_entries=4)
    with pytest.raises(ValueError, match="same width"):
        ops.edge_dot_raw(er, col, a, torch.randn(12, 4, device=dev))


@pytest.mark.parametrize("M,N,K", [(4099, 260, 2100), (1500, 772, 4001), (70000, 128, 300), (768, 772, 60499)])
def test_large_bf16x3_products_stay_inside_their_bound(dev, M, N, K):
    """The split-bf16 products after round 6's changes -- the three-bytes split (h = x & 0xFFFF0000, m = (x - h) & 0xFFFF0000,
    l = x - h - m), the slab count chosen by rounds, the tile order with the short side fastest -- at sizes where those matter
    (ragged, several rounds of workgroups, MMGCN's own weight-gradient shape): NT with bias + leaky-relu, TN, NN with the
    accumulate epilogue against torch's fp64 products within 1e-6 of sum |a||b|, and the same bits on a second call."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(M + N + K)
    if M * K > 3e8 or N * K > 3e8:
        pytest.skip("operand too large for the fp64 reference")
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    gy = torch.randn(M, N, device=dev, generator=g)
    base = torch.randn(M, K, device=dev, generator=g)

    def run():
        y = ops.gemm_nt_bf16x3(x, w, bias=b, act=1)
        gw = ops.gemm_tn_bf16x3(gy, x)                       # [N, K]
        gx = ops.gemm_nn_bf16x3(gy, w, out=base.clone(), accumulate=True)
        return y, gw, gx

    y, gw, gx = run()
    y2, gw2, gx2 = run()
    assert torch.equal(y, y2) and torch.equal(gw, gw2) and torch.equal(gx, gx2)
    xd, wd, gd = x.double(), w.double(), gy.double()
    ref_y = torch.nn.functional.leaky_relu(xd @ wd.t() + b.double())
    assert bool(((y.double() - ref_y).abs() <= 1e-6 * (xd.abs() @ wd.abs().t() + b.double().abs()) + 1e-12).all())
    assert bool(((gw.double() - gd.t() @ xd).abs() <= 1e-6 * (gd.abs().t() @ xd.abs())).all())
    assert bool(((gx.double() - (base.double() + gd @ wd)).abs() <= 1e-6 * (gd.abs() @ wd.abs() + base.double().abs())).all())


@pytest.mark.parametrize("capture", [True, False])
def test_fused_lightgcn_step_with_the_ordered_backward_is_reproducible_and_is_the_autograd_step(dev, capture, monkeypatch):
    """CHAOREC_BPR_ORDERED=2: the fused LightGCN step's gradient rows are added by the ordered launch too (one launch more per step).
    Twelve steps on baby with batches of 1 024 (hundreds of repeated rows per batch, where the atomic launch's sums depend on their
    order): two fused runs leave the SAME BITS in both tables, and they are the bits of the ordinary step it replaces
    (LightGCN.loss_drawn -> backward -> FusedAdam.step, whose BPR node runs the ordered backward by default)."""
    from conftest import load_interactions
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    monkeypatch.setenv("CHAOREC_BPR_ORDERED", "2")
    d = load_interactions("baby")
    U, I, edges = d["U"], d["I"], d["train"]
    uid = graph.user_item_dict_from_edges(edges)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)

    def model():
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, uid, 64, 1e-3, 3, "add", dev).to(dev)
        return m, FusedAdam(m.parameters(), lr=1e-3)

    def fused():
        m, o = model()
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        step = FusedLightGCNStep(m, o, batch_size=1024, edges=edges_dev, seed=7, step_dev=counter, capture=capture)
        for _ in range(12):
            step()
        torch.cuda.synchronize()
        return torch.cat((m.user_embedding.weight, m.item_embedding.weight)).detach().clone()

    a, b = fused(), fused()
    assert torch.equal(a, b)
    m, o = model()
    for it in range(12):
        o.zero_grad()
        m.loss_drawn(edges_dev, 1024, 7, it).backward()
        o.step()
    w = torch.cat((m.user_embedding.weight, m.item_embedding.weight)).detach()
    assert torch.equal(a, w), float((a - w).abs().max())


@pytest.mark.parametrize("D", [64, 128, 256])
def test_rowlist_striped_rows_are_the_dense_rows_bit_for_bit(dev, D, monkeypatch):
    """A list launch's very long rows go to D / 32 workgroups each, one per 128-byte column stripe (csrc/spmm.hip,
    spmm_rowlist_long_ws_kernel): the per-element chain of adds is untouched, so the rows -- with alpha / beta z and with the
    layer-mean epilogue -- are the dense launch's bit for bit.  A graph with a few items of 1 500-6 000 entries, the stripe
    threshold lowered to 1 000: rows below (long, one workgroup), above (striped) and short rows in one launch."""
    from chaorec_amd import graph, ops
    monkeypatch.setenv("CHAOREC_ROWLIST_STRIPE_T", "1000")
    rng = np.random.default_rng(5)
    U, I = 7000, 300
    heavy = {0: 6000, 1: 3100, 2: 1500, 3: 1001, 4: 1000, 5: 700, 6: 300}
    pairs = [np.stack([rng.choice(U, d, replace=False), np.full(d, i)], 1) for i, d in heavy.items()]
    pairs.append(np.stack([rng.integers(0, U, 9000), rng.integers(7, I, 9000)], 1))
    e = np.unique(np.concatenate(pairs), axis=0)
    edges = np.stack([e[:, 0], e[:, 1] + U], 1).astype(np.int32)
    csr = graph.lightgcn_csr(edges, U + I).to(dev)
    N = U + I
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(N, D, device=dev, generator=gen) * 0.1
    z = torch.randn(N, D, device=dev, generator=gen) * 0.1
    t0 = torch.randn(N, D, device=dev, generator=gen) * 0.1
    rows = np.unique(np.concatenate([U + np.arange(7), rng.integers(0, N, 150)])).astype(np.int32)
    lst = torch.from_numpy(rows).to(dev)
    n = torch.tensor([len(rows)], dtype=torch.int32, device=dev)
    long_rows = ops.long_row_buffers(csr, 256)
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).cpu().numpy()
    assert (deg[rows] > 1000).sum() >= 4 and ((deg[rows] > 256) & (deg[rows] <= 1000)).sum() >= 3
    want = ops.spmm_raw(csr, x, y=torch.empty_like(x), alpha=0.7, z=z, beta=0.3)
    got = torch.full_like(x, float("nan"))
    ops.spmm_rowlist_raw(csr, x, got, lst, n, alpha=0.7, z=z, beta=0.3, long_rows=long_rows)
    torch.cuda.synchronize()
    assert int(long_rows[1].abs().sum()) == 0                                   # counters left clean (the striped one too)
    sel = torch.from_numpy(rows.astype(np.int64)).to(dev)
    assert torch.equal(got[sel], want[sel])
    rest = torch.ones(N, dtype=torch.bool, device=dev)
    rest[sel] = False
    assert bool(torch.isnan(got[rest]).all())
    # the layer-mean epilogue: ((w t0 + w x) + w (A x)) in the rows of the list
    w = 1.0 / 3.0
    mean_want = torch.empty_like(x)
    ops.spmm_mean_raw(csr, x, [t0, x], w, mean_want)
    mean_got = torch.full_like(x, float("nan"))
    ops.spmm_rowlist_raw(csr, x, None, lst, n, mean_out=mean_got, mean_terms=[t0, x], mean_w=w, long_rows=long_rows)
    torch.cuda.synchronize()
    assert torch.equal(mean_got[sel], mean_want[sel])
    # a LONG list (capacity above 65 536 rows: stripes from four times the threshold on -- here the 6 000-entry row only)
    big = torch.zeros(70000, dtype=torch.int32, device=dev)
    big[:len(rows)] = lst
    longl = torch.full_like(x, float("nan"))
    ops.spmm_rowlist_raw(csr, x, longl, big, n, alpha=0.7, z=z, beta=0.3, long_rows=long_rows)
    torch.cuda.synchronize()
    assert torch.equal(longl[sel], want[sel]) and int(long_rows[1].abs().sum()) == 0
    # and without stripes: the same bits
    monkeypatch.setenv("CHAOREC_ROWLIST_STRIPE_T", "0")
    again = torch.full_like(x, float("nan"))
    ops.spmm_rowlist_raw(csr, x, again, lst, n, alpha=0.7, z=z, beta=0.3, long_rows=long_rows)
    torch.cuda.synchronize()
    assert torch.equal(again[sel], got[sel])


@pytest.mark.parametrize("M,N,K", [(1000, 772, 768), (60499, 772, 96), (700, 200, 5000), (129, 65, 40000)])
def test_bf16x3_products_do_not_depend_on_the_workgroup_order(dev, M, N, K, monkeypatch):
    """The XCD regrouping of the bf16x3 GEMMs' workgroups (csrc/gemm_bf16x3.hip: dispatch id d works on item
    (d mod 8) (n / 8) + d / 8 of the slab-major tile list, bijective form) is a permutation of the work: grids whose size
    is no multiple of 8, with and without k-slabs, give the dispatch-order launch's bits -- a map that were not a
    bijection would leave tiles of the output unwritten (the outputs start as NaN)."""
    from chaorec_amd import ops
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(M, K, device=dev, generator=gen)
    w = torch.randn(N, K, device=dev, generator=gen) * 0.05
    gy = torch.randn(M, N, device=dev, generator=gen)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CHAOREC_X3_XCD", mode)
        y = torch.full((M, N), float("nan"), device=dev)
        gw = torch.full((N, K), float("nan"), device=dev)
        gx = torch.full((M, K), float("nan"), device=dev)
        ops.gemm_nt_bf16x3(x, w, out=y)
        ops.gemm_tn_bf16x3(gy, x, out=gw)
        ops.gemm_nn_bf16x3(gy, w, out=gx)
        torch.cuda.synchronize()
        outs[mode] = (y, gw, gx)
    for a, b in zip(outs["1"], outs["0"]):
        assert not torch.isnan(a).any() and torch.equal(a, b)


def test_scoring_sweep_does_not_depend_on_the_workgroup_order(dev, monkeypatch):
    """... and the same for the prefilter sweep's (user workgroup, item split) grid (CHAOREC_SWEEP_XCD): the ranking is the
    dispatch-order launch's, ids and scores, and went through the prefilter route."""
    from chaorec_amd import ops
    gen = torch.Generator(device=dev).manual_seed(3)
    U, I, D, K = 5003, 9001, 64, 50
    ue = torch.randn(U, D, device=dev, generator=gen) * 0.1
    ie = torch.randn(I, D, device=dev, generator=gen) * 0.1
    hist = (torch.arange(U + 1, dtype=torch.int64, device=dev),                    # one masked item per user
            torch.randint(0, I, (U,), device=dev, generator=gen).to(torch.int32))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CHAOREC_SWEEP_XCD", mode)
        stats = {}
        idx, val = ops.score_topk(ue, ie, hist, 1e-6, K, stats=stats)
        torch.cuda.synchronize()
        res[mode] = (idx.clone(), val.clone(), stats)
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    assert res["1"][2], "no prefilter statistics: the call did not take the sweep's route"
