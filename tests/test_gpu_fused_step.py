"""FusedLightGCNStep (2L+2 launches, Adam in the last SpMM's epilogue, no zero fill, no autograd) against the ordinary
step it replaces (LightGCN.loss_drawn -> backward -> FusedAdam.step), on the same batch stream."""
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


def _pair(dev, U, I, edges, D, L, seed=0):
    from chaorec_amd import graph
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam
    uid = graph.user_item_dict_from_edges(edges)
    out = []
    for _ in range(2):
        torch.manual_seed(seed)
        m = LightGCN(U, I, edges, uid, D, 1e-3, L, "add", dev).to(dev)
        out.append((m, FusedAdam(m.parameters(), lr=1e-3)))
    return out


@pytest.mark.parametrize("D,L,capture,ordered", [(64, 3, True, "1"), (64, 3, False, "1"), (64, 2, True, "2"), (64, 1, True, "2"),
                                                 (128, 3, True, "1"), (32, 2, False, "2")])
def test_fused_step_equals_ordinary_step(dev, D, L, capture, ordered, monkeypatch):
    """ordered "1": the product's default -- the fused launch adds its gradient rows with fp32 atomics (the tolerance below is
    theirs); "2": through the ordered launch."""
    from chaorec_amd.optim import FusedLightGCNStep
    monkeypatch.setenv("CHAOREC_BPR_ORDERED", ordered)
    d = load_interactions("baby")
    U, I, edges = d["U"], d["I"], d["train"]
    (ref, oref), (fus, ofus) = _pair(dev, U, I, edges, D, L)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    step = FusedLightGCNStep(fus, ofus, batch_size=1024, edges=edges_dev, seed=7, step_dev=counter, loss_accum=acc,
                             capture=capture)
    assert int(counter) == 0 and float(acc) == 0.0 and int(ofus._step_dev) == 0       # warm-up rolled back
    total = 0.0
    for it in range(12):
        oref.zero_grad()
        lr_ = ref.loss_drawn(edges_dev, 1024, 7, it)
        lr_.backward()
        oref.step()
        lf = step()
        assert float(lf) == pytest.approx(float(lr_.detach()), rel=2e-6), it
        total += float(lf)
        assert not bool(step.G.any()), it                        # the batch-gradient buffer is all-zero again
        assert torch.equal(step.ids[0], ref.batch[0]) and torch.equal(step.ids[2], ref.batch[2])
    assert int(counter) == 12 and int(ofus._step_dev) == int(oref._step_dev) == 12
    assert float(acc) == pytest.approx(total, rel=1e-5)
    wr = torch.cat((ref.user_embedding.weight, ref.item_embedding.weight)).detach()
    wf = torch.cat((fus.user_embedding.weight, fus.item_embedding.weight)).detach()
    # same kernels and rounding; only the order of the batch's float atomics differs from run to run
    assert float((wr - wf).abs().max()) <= 2e-5 and float((wr - wf).abs().mean()) <= 1e-8
    assert torch.allclose(fus.result, ref.result.detach(), rtol=0, atol=1e-6)
    for k in ("exp_avg", "exp_avg_sq"):
        a, b = oref.state[ref.user_embedding.weight][k], ofus.state[fus.user_embedding.weight][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-9)
    # the ordinary path keeps working on the same optimizer state afterwards (short last batch of an epoch)
    ofus.zero_grad()
    loss = fus.loss_drawn(edges_dev, 300, 7, 100)
    loss.backward()
    ofus.step()
    assert int(ofus._step_dev) == 13 and torch.isfinite(loss)


def test_fused_step_epoch_permutation_and_rank(dev):
    """The training loop's form: batches from an epoch permutation (perm / perm_pos), then gene_ranklist on the
    result the fused step left behind."""
    from chaorec_amd.optim import FusedLightGCNStep
    d = load_interactions("baby")
    U, I, edges = d["U"], d["I"], d["train"]
    (ref, oref), (fus, ofus) = _pair(dev, U, I, edges, 64, 2)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    perm = torch.randperm(len(edges), device=dev, generator=g)
    pos_f, pos_r = (torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2))
    cnt_f, cnt_r = (torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2))
    step = FusedLightGCNStep(fus, ofus, batch_size=1024, edges=edges_dev, seed=5, step_dev=cnt_f, perm=perm, perm_pos=pos_f)
    for it in range(len(edges) // 1024):
        oref.zero_grad()
        l0 = ref.loss_drawn(edges_dev, 1024, 5, 0, step_dev=cnt_r, advance=True, perm=perm, perm_pos=pos_r)
        l0.backward()
        oref.step()
        l1 = step()
        assert float(l1) == pytest.approx(float(l0.detach()), rel=5e-6), it
    assert int(pos_f) == int(pos_r) == (len(edges) // 1024) * 1024
    ra, rb = fus.gene_ranklist(), ref.gene_ranklist()
    assert float((ra != rb).float().mean()) < 2e-3          # atomics-order noise can flip a near-tie, nothing more


@pytest.mark.parametrize("D,L", [(64, 3), (64, 2), (64, 1), (128, 2), (128, 3), (64, 5), (8, 3)])
def test_layer_mean_in_last_epilogue_is_bit_identical(dev, D, L):
    """ops.forward_layers: the whole layer mean formed in the LAST propagate's epilogue (few layers) must equal the
    per-layer acc epilogue bit for bit (both restate the reference's final += w * x_l order)."""
    from chaorec_amd import graph, ops
    d = load_interactions("baby")
    N = d["U"] + d["I"]
    csr = graph.lightgcn_csr(d["train"], N).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(D * 10 + L)
    x0 = torch.randn(N, D, device=dev, generator=g) * 0.1
    w = 1.0 / (L + 1)
    want = torch.empty_like(x0)
    x = x0
    for l in range(L):
        y = torch.empty_like(x0)
        ops.spmm_raw(csr, x, y=y, acc=want, acc_init=x0 if l == 0 else None, acc_w=w)
        x = y
    got = torch.full_like(x0, float("nan"))
    ops.forward_layers(csr, x0, L, got, [torch.empty_like(x0) for _ in range(L - 1)])
    assert torch.equal(got, want)
    assert torch.equal(ops.layer_mean_propagate(x0, csr, L), want)


def test_multi_step_replay_equals_single_steps(dev):
    from chaorec_amd.optim import FusedLightGCNStep
    d = load_interactions("baby")
    U, I, edges = d["U"], d["I"], d["train"]
    (a, oa), (b, ob) = _pair(dev, U, I, edges, 64, 3)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    ca, cb = (torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2))
    la, lb = (torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(2))
    sa = FusedLightGCNStep(a, oa, batch_size=1024, edges=edges_dev, seed=9, step_dev=ca, loss_accum=la, steps_per_replay=1)
    sb = FusedLightGCNStep(b, ob, batch_size=1024, edges=edges_dev, seed=9, step_dev=cb, loss_accum=lb, steps_per_replay=4)
    sa.run(11)
    sb.run(11)                      # 2 replays of 4 + 3 single steps
    assert int(ca) == int(cb) == 11 and int(oa._step_dev) == int(ob._step_dev) == 11
    assert float(la) == pytest.approx(float(lb), rel=1e-5)
    wa = torch.cat((a.user_embedding.weight, a.item_embedding.weight)).detach()
    wb = torch.cat((b.user_embedding.weight, b.item_embedding.weight)).detach()
    assert float((wa - wb).abs().max()) <= 2e-5 and float((wa - wb).abs().mean()) <= 1e-8
