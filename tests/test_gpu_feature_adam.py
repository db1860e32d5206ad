"""chaorec_adam_lowrank_f32 (Adam over a feature table with the gradient gy W, never materialised) and the FREEDOM
training path built on it (ops.linear_rows + optim.FusedAdam's claimed tables): Model/FREEDOM.py:59-60, 209-213 with
torch.optim.Adam (main.py:397)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _gy_sequence(rng, n, R, steps, frac):
    out = []
    for _ in range(steps):
        gy = np.zeros((n, R), np.float32)
        rows = rng.choice(n, max(1, int(n * frac)), replace=False)
        gy[rows] = (rng.standard_normal((rows.size, R)) * 0.05).astype(np.float32)
        out.append(gy)
    return out


@pytest.mark.parametrize("n,K,R,wd", [(301, 520, 64, 0.0), (97, 4096, 64, 0.0), (513, 384, 24, 0.0), (200, 260, 64, 1e-2)])
def test_adam_lowrank_dense_equals_adam_on_the_materialised_gradient(dev, oracle, n, K, R, wd):
    """mode 0 == chaorec_adam_step_f32 on g = gy W (the k-ascending chain of the f32 GEMM), bit for bit on p, m, v over
    several steps; and the CPU oracle's GEMM + Adam within its usual 2e-7."""
    from chaorec_amd import ops
    rng = np.random.default_rng(n + K)
    p0 = rng.standard_normal((n, K)).astype(np.float32)
    W = (rng.standard_normal((R, K)) * 0.1).astype(np.float32)
    seq = _gy_sequence(rng, n, R, 5, 0.2)
    p_a, p_b = (torch.from_numpy(p0.copy()).to(dev) for _ in range(2))
    m_a, v_a, m_b, v_b = (torch.zeros(n, K, device=dev) for _ in range(4))
    Wt = torch.from_numpy(W).to(dev)
    p_np, m_np, v_np = p0.copy(), np.zeros((n, K), np.float32), np.zeros((n, K), np.float32)
    for step, gy in enumerate(seq, 1):
        gyt = torch.from_numpy(gy).to(dev)
        ops.adam_lowrank(p_a, gyt, Wt, m_a, v_a, step, weight_decay=wd)
        g = ops.gemm_raw(gyt, Wt)
        ops.adam_step(p_b, g, m_b, v_b, step, weight_decay=wd)
        g_np = oracle.gemm(gy, W)
        assert np.array_equal(g.cpu().numpy(), g_np)
        oracle.adam_step(p_np, g_np, m_np, v_np, 1e-3, 0.9, 0.999, 1e-8, wd, step)
    for a, b in ((p_a, p_b), (m_a, m_b), (v_a, v_b)):
        assert torch.equal(a, b)
    assert np.allclose(p_a.cpu().numpy(), p_np, rtol=0, atol=2e-7)


@pytest.mark.parametrize("n,K,R,wd", [(301, 520, 64, 0.0), (1000, 384, 64, 0.0), (200, 260, 32, 1e-2)])
def test_adam_lowrank_lazy_then_flush_equals_dense(dev, n, K, R, wd):
    """mode 1 (only the batch rows, after replaying the zero-gradient steps they sat out) followed by a flush gives the
    same bits as updating every row every step; a catch-up of chosen rows makes exactly those rows current."""
    from chaorec_amd import ops
    rng = np.random.default_rng(7 * n + K)
    p0 = rng.standard_normal((n, K)).astype(np.float32)
    Wt = torch.from_numpy((rng.standard_normal((R, K)) * 0.1).astype(np.float32)).to(dev)
    seq = _gy_sequence(rng, n, R, 9, 0.1)
    betas = (0.9, 0.999)
    table = ops.adam_bias_table(6, betas, dev)              # shorter than the run: later steps computed in the launch
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    p_d, p_l = (torch.from_numpy(p0.copy()).to(dev) for _ in range(2))
    m_d, v_d, m_l, v_l = (torch.zeros(n, K, device=dev) for _ in range(4))
    last = torch.zeros((ops.adam_lowrank_strips(K), n), dtype=torch.int32, device=dev)
    for step, gy in enumerate(seq, 1):
        gyt = torch.from_numpy(gy).to(dev)
        ops.adam_lowrank(p_d, gyt, Wt, m_d, v_d, step, weight_decay=wd)
        if step == 5:                                       # rows about to be read: current for step 4 afterwards
            rows = torch.tensor([0, 3, n - 1], device=dev)
            flags = torch.zeros((n, 1), device=dev)
            flags[rows] = 1.0
            ref_rows = p_d_prev[rows]
            ops.adam_lowrank(p_l, flags, None, m_l, v_l, 0, weight_decay=wd, step_dev=step_dev, mode=3, last=last,
                             bc_table=table)
            assert torch.equal(p_l[rows], ref_rows)
            assert bool((last[:, rows] == 4).all())
        step_dev.fill_(step)
        ops.adam_lowrank(p_l, gyt, Wt, m_l, v_l, 0, weight_decay=wd, step_dev=step_dev, mode=1, last=last, bc_table=table)
        p_d_prev = p_d.clone()
    touched_ever = torch.from_numpy(np.any([np.any(g != 0, 1) for g in seq], 0)).to(dev)
    assert not torch.equal(p_l, p_d) or bool(touched_ever.all())      # stale rows exist before the flush
    ops.adam_lowrank(p_l, None, None, m_l, v_l, 0, weight_decay=wd, step_dev=step_dev, mode=2, last=last, bc_table=table)
    for a, b in ((p_l, p_d), (m_l, m_d), (v_l, v_d)):
        assert torch.equal(a, b)
    assert bool((last == len(seq)).all())


def test_adam_lowrank_row_list_path_equals_dense(dev):
    """The lazy modes driven by chaorec_unique_rows' list of the batch's distinct rows (duplicates in the batch, ids the
    gradient is zero for, repeated listing between steps) == every row every step, after the flush."""
    from chaorec_amd import ops
    n, K, R = 700, 1028, 64
    rng = np.random.default_rng(11)
    p0 = rng.standard_normal((n, K)).astype(np.float32)
    Wt = torch.from_numpy((rng.standard_normal((R, K)) * 0.1).astype(np.float32)).to(dev)
    table = ops.adam_bias_table(64, (0.9, 0.999), dev)
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    p_d, p_l = (torch.from_numpy(p0.copy()).to(dev) for _ in range(2))
    m_d, v_d, m_l, v_l = (torch.zeros(n, K, device=dev) for _ in range(4))
    last = torch.zeros((ops.adam_lowrank_strips(K), n), dtype=torch.int32, device=dev)
    claim, stamp = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    cap = 300
    rowlist, rowcount = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    for step in range(1, 13):
        batch = rng.integers(0, n, cap)                       # duplicates on purpose
        batch[:5] = batch[5:10]
        if step == 3:                                         # ids outside the table are dropped, not dereferenced
            bad = torch.from_numpy(np.concatenate([batch[:10], [n, n + 7, -1]])).to(dev)
            ops.unique_rows(bad, claim, stamp, rowlist, rowcount)
            assert int(rowcount[0]) == np.unique(batch[:10]).size
        gy = np.zeros((n, R), np.float32)
        hot = np.unique(batch)[3:]                            # three listed rows get a zero gradient row
        gy[hot] = (rng.standard_normal((hot.size, R)) * 0.05).astype(np.float32)
        gyt, bt = torch.from_numpy(gy).to(dev), torch.from_numpy(batch).to(dev)
        ops.adam_lowrank(p_d, gyt, Wt, m_d, v_d, step)
        for _ in range(2):                                    # listing twice gives the same list again
            ops.unique_rows(bt, claim, stamp, rowlist, rowcount)
            k = int(rowcount[0])
            assert k == np.unique(batch).size
            assert np.array_equal(np.sort(rowlist[:k].cpu().numpy()), np.unique(batch))
        before = p_l.clone()
        ops.adam_lowrank(p_l, None, None, m_l, v_l, 0, step_dev=step_dev, mode=3, last=last, bc_table=table,
                         rowlist=(rowlist, rowcount))          # step_dev = step - 1: current for the steps so far
        others = torch.ones(n, dtype=torch.bool, device=dev)
        others[bt] = False
        assert torch.equal(p_l[others], before[others])       # only listed rows were touched
        assert bool((last[:, bt] == step - 1).all())
        step_dev.fill_(step)
        ops.adam_lowrank(p_l, gyt, Wt, m_l, v_l, 0, step_dev=step_dev, mode=1, last=last, bc_table=table,
                         rowlist=(rowlist, rowcount))
        assert torch.equal(p_l[bt], p_d[bt]) and torch.equal(m_l[bt], m_d[bt])
    ops.adam_lowrank(p_l, None, None, m_l, v_l, 0, step_dev=step_dev, mode=2, last=last, bc_table=table)
    for a, b in ((p_l, p_d), (m_l, m_d), (v_l, v_d)):
        assert torch.equal(a, b)


def _freedom(dev, claim):
    from test_gpu_models import _make_freedom
    g = load_golden("freedom_small_nodrop.npz")
    m, U, I = _make_freedom(g, dev)
    if not claim:
        del m.image_embedding.weight._chaorec_projected_only, m.text_embedding.weight._chaorec_projected_only
    m.pre_epoch_processing()
    return m, g, U, I


def _batches(g, U, I, n, dev):
    rng = np.random.default_rng(5)
    B = int(g["users"].shape[0])
    return [(torch.from_numpy(rng.integers(0, U, B)).to(dev), torch.from_numpy(rng.integers(U, U + I, B)).to(dev),
             torch.from_numpy(rng.integers(U, U + I, B)).to(dev)) for _ in range(n)]


def _train(m, opt, batches):
    losses = []
    for b in batches:
        opt.zero_grad()
        loss = m.loss(*b)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


def test_freedom_training_with_claimed_feature_tables(dev):
    """Eight training steps of FREEDOM three ways: (a) dense feature gradients + FusedAdam (the path the goldens check),
    (b) FusedAdam with the feature tables claimed (no [I, K] gradient, every row updated every step), (c) the same with
    lazily updated rows + flush.  All three agree to rounding: (a) vs (b) differ in the rounding of gy W (split-bf16
    GEMM vs fp32 chain), (b) vs (c) only in the order of the float atomics of the two runs."""
    from chaorec_amd.optim import FusedAdam
    out = {}
    for tag, claim, lazy in (("dense_grad", False, False), ("claimed", True, False), ("lazy", True, True)):
        m, g, U, I = _freedom(dev, claim)
        opt = FusedAdam(m.parameters(), lr=1e-3, lazy_rows=lazy)
        assert bool(opt._claimed) == claim
        losses = _train(m, opt, _batches(g, U, I, 8, dev))
        if claim:
            assert m.image_embedding.weight.grad is None and m.text_embedding.weight.grad is None
        if lazy:
            stale = m.image_embedding.weight.detach().clone()
            opt.flush()
            assert not torch.equal(stale, m.image_embedding.weight)      # some rows were behind
        out[tag] = (losses, {k: v.detach().clone() for k, v in m.named_parameters()},
                    {k: {n: opt.state[p][n].clone() for n in ("exp_avg", "exp_avg_sq")} for k, p in m.named_parameters()})
    # (two runs are not bit-comparable: the BPR backward and the scatter of gy add with float atomics; the bitwise
    # lazy == dense statement is test_adam_lowrank_lazy_then_flush_equals_dense's)
    for k in out["claimed"][1]:
        assert torch.allclose(out["claimed"][1][k], out["lazy"][1][k], rtol=0, atol=5e-7), k
        for n in out["claimed"][2][k]:
            assert torch.allclose(out["claimed"][2][k][n], out["lazy"][2][k][n], rtol=1e-4, atol=1e-9), (k, n)
        assert torch.allclose(out["claimed"][1][k], out["dense_grad"][1][k], rtol=0, atol=2e-6), k
    assert np.allclose(out["claimed"][0], out["lazy"][0], rtol=1e-6)
    assert np.allclose(out["claimed"][0], out["dense_grad"][0], rtol=1e-6)
    # the untouched-row updates really happened in the claimed run: rows outside every batch moved by momentum only if
    # they had been in an earlier batch; all tables differ from their initial values somewhere
    assert not torch.equal(out["claimed"][1]["image_embedding.weight"].cpu(), torch.from_numpy(g["v_feat"]))


def test_freedom_claimed_step_captured_equals_eager(dev):
    """GraphedTrainStep over the claimed (and the lazy) path replays the same updates as the eager loop."""
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    for lazy in (False, True):
        res = []
        for captured in (False, True):
            m, g, U, I = _freedom(dev, True)
            opt = FusedAdam(m.parameters(), lr=1e-3, lazy_rows=lazy)
            batches = _batches(g, U, I, 6, dev)
            if captured:
                step = GraphedTrainStep(m, opt, example_batch=batches[0])
                for b in batches:
                    step(*b)
            else:
                _train(m, opt, batches)
            opt.flush()
            torch.cuda.synchronize()
            res.append({k: v.detach().clone() for k, v in m.named_parameters()})
        for k in res[0]:
            assert torch.allclose(res[0][k], res[1][k], rtol=0, atol=5e-7), (lazy, k)


@pytest.mark.parametrize("lazy", [False, True])
def test_sharded_freedom_trains_like_freedom_with_claimed_tables(dev, lazy):
    """dist.ShardedFREEDOM (one rank) + FusedAdam: the claimed tables' gy goes through sync_grads()' reduction hook, after
    which the update finds its rows by scanning the summed gy (lazy: mode 1 without a list) -- same parameters as the
    single-process FREEDOM + FusedAdam after six steps."""
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    m, g, U, I = _freedom(dev, True)
    sh = cdist.ShardedFREEDOM(m, [0, U], 1, 0, dev)
    opt_m, opt_s = FusedAdam(m.parameters(), lr=1e-3), FusedAdam(sh.parameters(), lr=1e-3, lazy_rows=lazy)
    assert len(opt_s._claimed) == 2
    batches = _batches(g, U, I, 6, dev)
    _train(m, opt_m, batches)
    for b in batches:
        sh.zero_grad()
        loss = sh.loss(b[0], b[1] - U, b[2] - U)
        loss.backward()
        assert sh.image_embedding.weight.grad is None
        assert not any(p is sh.image_embedding.weight for p in sh.replicated_parameters())
        sh.sync_grads()
        opt_s.step()
    opt_s.flush()
    opt_m.flush()                       # (lazy rows are the default: a reader of the tables' rows flushes first)
    ref = dict(m.named_parameters())
    for n, p in sh.named_parameters():
        assert torch.allclose(p, ref[n], rtol=0, atol=2e-6), n


def test_adam_multi_equals_one_launch_per_tensor(dev):
    """chaorec_adam_multi_f32 (one launch, pointers in the kernel argument) == chaorec_adam_step_f32 per tensor, bit for
    bit: sizes below / at / above a block, a size that is no multiple of 4, views at 4-byte (not 16-byte) alignment, an
    empty tensor; and FusedAdam, which batches its small tensors through it, == torch.optim.Adam to rounding."""
    from chaorec_amd import ops
    from chaorec_amd.optim import FusedAdam
    torch.manual_seed(3)
    sizes = [1, 3, 64, 4096, 4097, 8192 + 5, 100_003, 0, 64 * 64, 12]
    flat = torch.randn(sum(sizes) + 16, device=dev)
    tensors, ref, o = [], [], 1                        # offset 1: every view is only 4-byte aligned
    for n in sizes:
        p = flat[o:o + n]
        o += n
        g = torch.randn(n, device=dev) * 0.1
        m, v = torch.rand(n, device=dev) * 0.01, torch.rand(n, device=dev) * 0.001
        tensors.append((p, g, m, v, n))
        ref.append((p.clone(), g, m.clone(), v.clone(), n))
    aligned = [(torch.randn(n, device=dev), torch.randn(n, device=dev), torch.zeros(n, device=dev),
                torch.zeros(n, device=dev), n) for n in (7, 640, 5000)]
    ref += [(p.clone(), g, m.clone(), v.clone(), n) for p, g, m, v, n in aligned]
    tensors += aligned
    assert len(tensors) <= ops.adam_multi_max()
    for step in (1, 2, 7):
        ops.adam_multi(tensors, step, weight_decay=1e-3)
        for p, g, m, v, n in ref:
            if n:
                ops.adam_step(p, g, m, v, step, weight_decay=1e-3)
    for a, b in zip(tensors, ref):
        for x, y in ((a[0], b[0]), (a[2], b[2]), (a[3], b[3])):
            assert torch.equal(x, y)
    # through the optimizer: 60 small parameters (two launches) against torch.optim.Adam
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in [33, 64, 4096, 5000] * 15]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fa, ta = FusedAdam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2)
    for _ in range(3):
        for p, q in zip(ps, qs):
            p.grad = torch.randn_like(p) * 0.1
            q.grad = p.grad.clone()
        fa.step()
        ta.step()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=0, atol=1e-6)


def test_mgcn_training_with_claimed_feature_tables(dev):
    """MGCN projects its trainable feature tables as a WHOLE (ops.linear over every item row, Model/MGCN.py:80-83, 125-
    126): gy is dense, the gradient gy W still has rank 64.  Five FusedAdam steps with the tables claimed (no [I, K]
    gradient, no input-gradient GEMM) against five with the dense gradients: same parameters to rounding."""
    from chaorec_amd.Model import MGCN
    from chaorec_amd import graph
    from chaorec_amd.optim import FusedAdam
    g = load_golden("mgcn_small.npz")
    U, I = int(g["U"]), int(g["I"])
    rng = np.random.default_rng(9)
    B = int(g["users"].shape[0])
    batches = [(torch.from_numpy(rng.integers(0, U, B)), torch.from_numpy(rng.integers(U, U + I, B)),
                torch.from_numpy(rng.integers(U, U + I, B))) for _ in range(5)]
    out = {}
    for claim in (False, True):
        torch.manual_seed(0)
        m = MGCN(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                 torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]), 2, "add", float(g["ssl_temp"]),
                 float(g["ssl_alpha"]), dev).to(dev)
        if not claim:
            del m.image_embedding.weight._chaorec_projected_only, m.text_embedding.weight._chaorec_projected_only
        opt = FusedAdam(m.parameters(), lr=1e-3)
        assert len(opt._claimed) == (2 if claim else 0)
        _train(m, opt, batches)
        assert (m.image_embedding.weight.grad is None) == claim
        out[claim] = {k: v.detach().clone() for k, v in m.named_parameters()}
    for k in out[True]:
        assert torch.allclose(out[True][k], out[False][k], rtol=0, atol=2e-6), k
    assert not torch.equal(out[True]["image_embedding.weight"].cpu(), torch.from_numpy(g["v_feat"]))


def test_feature_adam_edge_cases(dev):
    """Empty and minimal inputs, and the argument checks of the new entry points: an empty table and an empty batch are
    no-ops; K = 4 / R = 1 is the smallest table the kernels take; bad shapes fail loudly instead of launching."""
    from chaorec_amd import ops, _lib
    lib = _lib.load()
    z = torch.zeros(0, 8, device=dev)
    ops.adam_lowrank(z, torch.zeros(0, 2, device=dev), torch.zeros(2, 8, device=dev), z.clone(), z.clone(), 1)   # no rows
    ops.adam_multi([], 1)
    claim, stamp = torch.zeros(5, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    lst, cnt = torch.zeros(4, dtype=torch.int32, device=dev), torch.full((1,), 7, dtype=torch.int32, device=dev)
    ops.unique_rows(torch.zeros(0, dtype=torch.int64, device=dev), claim, stamp, lst, cnt)
    assert int(cnt[0]) == 0 and int(stamp[0]) == 1
    # smallest table: K = 4, R = 1
    p, m, v = torch.ones(3, 4, device=dev), torch.zeros(3, 4, device=dev), torch.zeros(3, 4, device=dev)
    gy, W = torch.tensor([[1.0], [0.0], [-2.0]], device=dev), torch.tensor([[0.5, -1.0, 0.0, 2.0]], device=dev)
    q, mq, vq = p.clone(), m.clone(), v.clone()
    ops.adam_lowrank(p, gy, W, m, v, 1)
    ops.adam_step(q, gy @ W, mq, vq, 1)
    assert torch.equal(p, q) and torch.equal(m, mq) and torch.equal(v, vq)
    assert torch.equal(p[1], torch.ones(4, device=dev))                      # zero gradient, zero moments: untouched
    # argument checks
    for bad in (dict(K=6), dict(R=65), dict(mode=4)):
        rc = lib.chaorec_adam_lowrank_f32(p.data_ptr(), gy.data_ptr(), W.data_ptr(), m.data_ptr(), v.data_ptr(), 3,
                                          bad.get("K", 4), bad.get("R", 1), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None,
                                          bad.get("mode", 0), None, None, 0, None, None, 0, 0, None)
        assert rc != 0, bad
    rc = lib.chaorec_adam_lowrank_f32(p.data_ptr(), gy.data_ptr(), W.data_ptr(), m.data_ptr(), v.data_ptr(), 3, 4, 1, 1e-3,
                                      0.9, 0.999, 1e-8, 0.0, 1, None, 1, None, None, 0, None, None, 0, 0, None)
    assert rc != 0                                                            # lazy mode without `last`
    with pytest.raises(ValueError):
        ops.adam_lowrank(p, gy, torch.zeros(1, 8, device=dev), m, v, 1)       # W's width is not the table's
    with pytest.raises(RuntimeError):
        ops.adam_lowrank(torch.ones(3, 4), gy.cpu(), W.cpu(), torch.zeros(3, 4), torch.zeros(3, 4), 1)   # no CPU path
