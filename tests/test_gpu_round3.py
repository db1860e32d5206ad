"""Round-3 additions on the GPU: graph builders that run on the device equal the host builders bit for bit; a full-rank
call whose workspace is cut into user ranges equals the one-piece call; (further down) the block-joint selection."""
import os

import numpy as np
import pytest
import torch

from conftest import load_interactions

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def test_device_built_csr_and_history_equal_the_host_builders(dev):
    """graph.lightgcn_csr / user_hist_csr_from_edges on a CUDA edge list (config 5's path) against the numpy path on the
    real sports interactions: same rowptr, same entry order, same normalisation bits."""
    from chaorec_amd import graph
    d = load_interactions("sports")
    U, I = d["U"], d["I"]
    host = graph.lightgcn_csr(d["train"], U + I)
    on_dev = graph.lightgcn_csr(torch.from_numpy(d["train"]).to(dev), U + I)
    assert on_dev.col.is_cuda and on_dev.symmetric
    for name in ("rowptr", "col", "val"):
        assert torch.equal(getattr(host, name), getattr(on_dev, name).cpu()), name
    h0 = graph.user_hist_csr_from_edges(d["train"], U)
    h1 = graph.user_hist_csr_from_edges(torch.from_numpy(d["train"]).to(dev), U)
    assert torch.equal(h0[0], h1[0].cpu()) and torch.equal(h0[1], h1[1].cpu())


def test_synthetic_generator_on_the_device(dev):
    from chaorec_amd.synthetic import synthetic_interactions_torch
    U, I, E = 300_000, 60_000, 6_000_000
    e = synthetic_interactions_torch(U, I, E, seed=1, device=dev, chunk_users=70_000)
    assert e.is_cuda and e.dtype == torch.int32 and abs(len(e) - E) <= 0.002 * E
    u, i = e[:, 0].long(), e[:, 1].long()
    assert bool((u[1:] >= u[:-1]).all()) and int(i.min()) >= U and int(i.max()) < U + I
    key = u * I + (i - U)
    assert key.unique().numel() == key.numel()                   # no duplicate interaction
    deg = torch.bincount(u, minlength=U)
    assert int(deg.min()) >= 3 and int(deg.max()) <= 256
    ideg = torch.bincount(i - U, minlength=I)
    assert int(ideg.max()) > 40 * float(ideg.float().mean())      # heavy-tailed item popularity


@pytest.mark.parametrize("hinted", [False, True])
def test_score_topk_in_user_chunks_equals_one_piece(dev, hinted):
    from chaorec_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    U, I, D = 20_000, 9_000, 64
    ue = torch.randn(U, D, generator=g, device=dev) * 0.2
    ie = torch.randn(I, D, generator=g, device=dev) * 0.2
    cnt = torch.randint(0, 12, (U,), generator=g, device=dev)
    rowptr = torch.zeros(U + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    col = torch.cat([torch.sort(torch.randperm(I, generator=g, device=dev)[:int(c)]).values for c in cnt[:64].tolist()]
                    + [torch.arange(int(c), device=dev) * 7 for c in cnt[64:].tolist()]).to(torch.int32)
    hist = (rowptr, col)
    kw = {}
    if hinted:
        hint = torch.empty(U, device=dev)
        ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False)
        kw = dict(hint=hint.clone(), hint_valid=True, counters=torch.zeros(4, dtype=torch.int32, device=dev))
    want_i, want_v = ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U, **kw)
    one = ops._lib.load().chaorec_score_topk_workspace_bytes(U, I, 50, D)
    old = os.environ.get("CHAOREC_SCORE_WS_LIMIT")
    os.environ["CHAOREC_SCORE_WS_LIMIT"] = str(one // 3)
    try:
        st = {}
        if hinted:
            kw["hint"] = hint.clone()
        got_i, got_v = ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U, stats=st, **kw)
    finally:
        if old is None:
            del os.environ["CHAOREC_SCORE_WS_LIMIT"]
        else:
            os.environ["CHAOREC_SCORE_WS_LIMIT"] = old
    assert st["user_chunks"] >= 3 and st["prefilter_users"] == U
    assert torch.equal(got_i, want_i) and torch.equal(got_v, want_v)


@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("capture", [False, True])
def test_fused_sharded_step_one_rank_equals_the_unsharded_fused_step(dev, L, capture):
    """dist.FusedShardedLightGCNStep on a single shard that holds every user == optim.FusedLightGCNStep: the joined
    shard graph is the whole graph, the exchanges are no-ops, and the item rows (layer mean by chaorec_rows_mean_f32,
    Adam by one fused launch) must come out like the rows the unsharded step updates in its SpMM epilogues."""
    from chaorec_amd import dist as cdist, graph
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, B, T = 3000, 1200, 20000, 64, 256, 4
    edges = synthetic_interactions(U, I, E, seed=2)
    torch.manual_seed(3)
    ref = LightGCN(U, I, edges, None, D, 1e-3, L, "add", dev).to(dev)
    shard = cdist.UserShard(edges, U, I, 1, 0, dev)
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, L, dev, seed=1).to(dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(ref.user_embedding.weight)
        m.item_embedding.weight.copy_(ref.item_embedding.weight)
    # the joined shard graph IS the unsharded normalised graph
    j = cdist.joined_shard_csr(shard)
    assert torch.equal(j.rowptr, ref.graph.rowptr) and torch.equal(j.col, ref.graph.col) and torch.equal(j.val, ref.graph.val)
    s_ref = FusedLightGCNStep(ref, FusedAdam(ref.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=False)
    s_sh = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True,
                                          capture=capture)
    rng = np.random.default_rng(5)
    for t in range(T):
        sel = rng.choice(E, B, replace=False)
        users = torch.from_numpy(edges[sel, 0].astype(np.int64)).to(dev)
        pos = torch.from_numpy(edges[sel, 1].astype(np.int64)).to(dev)
        neg = torch.from_numpy(rng.integers(U, U + I, B)).to(dev)
        l0, l1 = float(s_ref(users, pos, neg)), float(s_sh(users, pos, neg))
        assert l1 == pytest.approx(l0, rel=1e-6), t
    assert float(s_sh.G.abs().max()) == 0.0
    # (the BPR backward's float atomics: last-bit noise between any two runs, ten-fold by lr 1e-2 / eps-sized moments)
    assert torch.allclose(m.user_embedding.weight, ref.user_embedding.weight, rtol=0, atol=5e-6)
    assert torch.allclose(m.item_embedding.weight, ref.item_embedding.weight, rtol=0, atol=5e-6)
    assert torch.allclose(m.result_u, ref.result[:U], rtol=0, atol=5e-6) and torch.allclose(m.result_i, ref.result[U:], rtol=0, atol=5e-6)


def test_rows_mean_matches_the_layer_mean_association(dev):
    from chaorec_amd import ops
    g = torch.Generator(device=dev).manual_seed(4)
    terms = [torch.randn(5000, 64, generator=g, device=dev) for _ in range(4)]
    w = 0.25
    out = ops.rows_mean(terms, w, torch.empty_like(terms[0]))
    a = np.float32(w) * terms[0].cpu().numpy()
    for t in terms[1:]:
        a = a + np.float32(w) * t.cpu().numpy()
    assert np.array_equal(out.cpu().numpy(), a)


def test_captured_step_refuses_a_stale_graph_and_optimizer_step_survives_state_dict(dev):
    """ADVICE (round 2): a captured step whose model re-allocated its graph must fail loudly, not replay dead addresses;
    FusedAdam's device step counter travels in state_dict (bias corrections and lazy-row stamps depend on it)."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 1500, 700, 9000
    edges = synthetic_interactions(U, I, E, seed=4)
    torch.manual_seed(0)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 2, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    sel = rng.choice(E, 128, replace=False)
    batch = (torch.from_numpy(edges[sel, 0].astype(np.int64)), torch.from_numpy(edges[sel, 1].astype(np.int64)),
             torch.from_numpy(rng.integers(U, U + I, 128)))
    step = GraphedTrainStep(m, opt, example_batch=batch)
    for _ in range(3):
        step(*batch)
    m.graph_generation = getattr(m, "graph_generation", 0) + 1          # what FREEDOM / LayerGCN do when they rebind their graph
    with pytest.raises(RuntimeError, match="graph_generation"):
        step(*batch)
    # a rebind BEFORE a capture is harmless (the old flag made the first replay of such a step raise)
    step2 = GraphedTrainStep(m, opt, example_batch=batch)
    step2(*batch)
    sd = opt.state_dict()
    assert sd["chaorec_step"] == 4
    import copy
    opt2 = FusedAdam(m.parameters(), lr=1e-3)
    opt2.load_state_dict(copy.deepcopy(sd))       # (as from a file: torch's load_state_dict keeps tensors that already fit)
    assert int(opt2._step_dev.item()) == 4
    w0 = m._flat.detach().clone()
    for o in (opt, opt2):                                   # the same fourth step from either optimizer
        with torch.no_grad():
            m._flat.copy_(w0)
        o.zero_grad()
        m.loss(*batch).backward()
        o.step()
        if o is opt:
            w_a = m.user_embedding.weight.detach().clone()
    assert torch.allclose(m.user_embedding.weight, w_a, rtol=0, atol=1e-6)


def test_freedom_result_is_fresh_under_a_captured_step(dev):
    """FREEDOM.result is concatenated lazily from the forward's two halves; under GraphedTrainStep those halves are the
    hipGraph's static buffers and no Python runs per replay -- the concatenation must therefore never be cached (a cached
    one would make every evaluation after the first rank the tables of the first)."""
    from conftest import load_golden
    from chaorec_amd import graph
    from chaorec_amd.Model import FREEDOM
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = load_golden("freedom_small_nodrop.npz")
    U, I = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = FREEDOM(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["D"]), float(g["reg"]), float(g["dropout"]),
                int(g["L"]), int(g["mm_layers"]), int(g["knn"]), float(g["w"]), dev).to(dev)
    m.pre_epoch_processing()
    opt = FusedAdam(m.parameters(), lr=1e-2)
    batch = tuple(torch.from_numpy(g[k]) for k in ("users", "pos", "neg"))
    step = GraphedTrainStep(m, opt, example_batch=batch)
    step(*batch)
    r1 = m.result.clone()
    for _ in range(5):
        step(*batch)
    r2 = m.result
    assert not torch.equal(r1, r2)                      # five Adam steps at lr 1e-2 moved the tables
    # what an eager forward at the weights BEFORE the last update gives is what the last replay left (stale-result quirk):
    # check against a forward at the current weights up to one step's movement, and exactly against the parts
    assert torch.equal(r2[:U], m._result_parts[0]) and torch.equal(r2[U:], m._result_parts[1])
    with torch.no_grad():
        fu, fi = m.forward(m.masked_adj)
    assert float((torch.cat((fu, fi), 0) - r2).abs().max()) < 0.05


def test_bpr_multi_one_launch_equals_term_by_term(dev, monkeypatch):
    """chaorec_bpr_multi_{fwd,bwd}_f32 (all of FREEDOM's three BPR terms in one launch each way, the weighted sum in the
    finalize) against T calls of chaorec_bpr_fwd_f32 / chaorec_bpr_bwd_f32 and the weighted sum in torch: per-term losses
    bit-identical, total to one rounding, gradients to the atomics' order."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    U, I, B, D = 3000, 2000, 1024, 64
    tab_u = torch.randn(U, D, device=dev, generator=g) * 0.1
    users = torch.randint(0, U, (B,), device=dev, generator=g)
    wvec = torch.tensor([1.0, 1e-3, 1e-3], device=dev)

    def run(limit):
        monkeypatch.setattr(ops, "BPR_MULTI_MAX", limit)
        tu = tab_u.clone().requires_grad_(True)
        gg = torch.Generator(device=dev)
        gg.manual_seed(12)
        terms, leaves = [], []
        for k, rows in enumerate((I, 2 * B, 2 * B)):
            t = (torch.randn(rows, D, device=dev, generator=gg) * 0.1).requires_grad_(True)
            leaves.append(t)
            terms.append((t, torch.randint(0, rows, (B,), device=dev, generator=gg), torch.randint(0, rows, (B,), device=dev, generator=gg)))
        loss = ops.bpr_loss_multi(tu, users, ops.VARIANT_LOGSIGMOID, terms, wvec)
        (loss * 1.5).backward()
        return loss.detach(), tu.grad, [t.grad for t in leaves]

    l1, gu1, gi1 = run(4)
    l0, gu0, gi0 = run(0)
    assert float((l1 - l0).abs()) <= 2e-7 * float(l0.abs())
    assert torch.allclose(gu1, gu0, rtol=0, atol=1e-6 * float(gu0.abs().max()))          # (a few ulp of the largest addend)
    for a, b in zip(gi1, gi0):
        assert torch.allclose(a, b, rtol=0, atol=1e-6 * float(b.abs().max()) + 1e-12)
    # deterministic forward
    assert torch.equal(run(4)[0], l1)


def test_mmgcn_branches_on_two_streams_train_like_one_stream(dev, monkeypatch):
    """CHAOREC_MMGCN_STREAMS (the visual branch on a side stream, forward and backward): six CAPTURED training steps at a size
    that takes the split-bf16 pipe and the dual products -- every parameter the same bits as the one-stream run; repeated
    three times (a race between the streams would show as a varying error)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import MMGCN
    import sys
    mm = sys.modules["chaorec_amd.Model.MMGCN"]          # (the package re-exports the class under the module's name)
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 6000, 2500, 40000, 512
    edges = synthetic_interactions(U, I, E, seed=3)
    uid = graph.user_item_dict_from_edges(edges)
    g = torch.Generator().manual_seed(4)
    v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 256, generator=g)
    rng = np.random.default_rng(9)
    batches = []
    for _ in range(6):
        sel = rng.choice(E, B, replace=False)
        u = torch.from_numpy(edges[sel, 0].astype(np.int64))
        pos = torch.from_numpy(edges[sel, 1].astype(np.int64))
        neg = torch.from_numpy(rng.integers(U, U + I, B))
        batches.append((torch.stack((u, u), 1).to(dev), torch.stack((pos, neg), 1).to(dev)))

    def run(streams):
        monkeypatch.setattr(mm, "BRANCH_STREAMS", streams)
        torch.manual_seed(21)
        m = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-3)
        step = GraphedTrainStep(m, opt, example_batch=batches[0])
        for b in batches:
            step(*b)
        torch.cuda.synchronize()
        return {n: p.detach().clone() for n, p in m.named_parameters()}

    ref = run(False)
    for rep in range(3):
        got = run(True)
        for n in ref:
            # (round 6: the BPR backward adds its rows in a fixed order -- no atomics --, so the comparison is exact)
            assert torch.equal(got[n], ref[n]), (rep, n, float((got[n] - ref[n]).abs().max()))


def test_fused_adam_early_tables_equal_the_in_step_update(dev):
    """FusedAdam.early_tables (a claimed table's dense update launched by submit() on a side stream, step offset 1, joined by
    step() before the counter moves): four eager steps of a Linear over a claimed table, bit-identical to the in-step update."""
    from chaorec_amd import ops
    from chaorec_amd.optim import FusedAdam
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    t0 = torch.randn(3000, 256, device=dev, generator=g)
    w0 = torch.randn(64, 256, device=dev, generator=g) * 0.05
    rows = [torch.randperm(3000, device=dev, generator=g)[:512] for _ in range(4)]     # (distinct: index_add_ stays deterministic)
    tgt = torch.randn(512, 64, device=dev, generator=g)

    def run(early):
        table = torch.nn.Parameter(t0.clone())
        table._chaorec_projected_only = True
        lin = torch.nn.Linear(256, 64, bias=True).to(dev)
        with torch.no_grad():
            lin.weight.copy_(w0)
            lin.bias.zero_()
        opt = FusedAdam([table, lin.weight, lin.bias], lr=1e-2)
        opt.early_tables = early
        for r in rows:
            opt.zero_grad()
            y = ops.linear_rows(table, r, lin.weight, lin.bias)
            ((y - tgt) ** 2).mean().backward()
            opt.step()
        torch.cuda.synchronize()
        return table.detach().clone(), lin.weight.detach().clone()

    ta, wa = run(False)
    tb, wb = run(True)
    assert torch.equal(ta, tb) and torch.equal(wa, wb)


def test_gene_ranklist_reaches_the_light_mode(dev, monkeypatch):
    """ranking.gene_ranklist consumes the previous call's queue counters ONCE per call (RankState.use_hints): from the third
    evaluation of unchanged tables on, the call carries thresholds AND runs without the retry pass (light), and the lists
    stay identical."""
    from chaorec_amd import ops, ranking
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    U, I, D = 3000, 5000, 64
    res = torch.randn(U + I, D, device=dev, generator=g) * 0.1
    hist = (torch.arange(U + 1, dtype=torch.int64, device=dev) * 0, torch.zeros(1, dtype=torch.int32, device=dev))
    seen = []
    real = ops.score_topk

    def spy(*a, **k):
        seen.append((bool(k.get("hint_valid")), bool(k.get("light"))))
        return real(*a, **k)

    monkeypatch.setattr(ops, "score_topk", spy)
    state = ranking.RankState()
    lists = []
    for _ in range(4):
        lists.append(ranking.gene_ranklist(res, U, I, hist, 1e-6, 50, to_cpu=False, state=state).clone())
        torch.cuda.synchronize()
    assert seen[0] == (False, False) and seen[1][0] and seen[-1] == (True, True), seen
    assert all(torch.equal(lists[0], x) for x in lists[1:])
