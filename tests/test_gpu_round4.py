"""Round-4 additions on the GPU: the early-Adam ordering fix for dense readers, the ranking tail, the
self-launching bench (two ranks on this one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def test_early_adam_with_a_dense_reader_equals_the_in_step_update(dev):
    """FusedAdam.early_tables with a table that ops.linear reads WHOLE (VBPR's / MGCN's feature tables: submit(...,
    dense_reader=True)): the node's weight gradient reads every row of the table, so it must be queued before submit()
    starts the table's in-place update on the side stream (it was queued after).  Four steps, eager and
    captured (GraphedTrainStep), bit-identical to the in-step update."""
    from chaorec_amd import ops
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    I, K, R = 6000, 1024, 64
    t0 = torch.randn(I, K, device=dev, generator=g)
    w0 = torch.randn(R, K, device=dev, generator=g) * 0.03
    tgt = torch.randn(I, R, device=dev, generator=g)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(t0.clone())
            self.table._chaorec_projected_only = True
            self.lin = torch.nn.Linear(K, R, bias=True)
            with torch.no_grad():
                self.lin.weight.copy_(w0)
                self.lin.bias.zero_()

        def loss(self):
            y = ops.linear(self.table, self.lin.weight, self.lin.bias)
            return ops.mean_all((y - tgt) ** 2)

    def run(early, captured):
        net = Net().to(dev)
        net.lin.weight.data.copy_(w0)
        opt = FusedAdam(list(net.parameters()), lr=1e-2)
        opt.early_tables = early
        if captured:
            step = GraphedTrainStep(net, opt, batch_fn=lambda: (), loss_fn=net.loss)
            opt.early_tables = early
            for _ in range(4):
                step()
        else:
            for _ in range(4):
                opt.zero_grad()
                net.loss().backward()
                opt.step()
        torch.cuda.synchronize()
        return net.table.detach().clone(), net.lin.weight.detach().clone()

    ref = run(False, False)
    for early, captured in ((True, False), (True, True), (False, True)):
        got = run(early, captured)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), (early, captured)


def _run_bench(extra, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        env["CHAOREC_BENCH_DETAIL"] = os.path.join(tmp, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                           timeout=timeout)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        detail = json.load(open(env["CHAOREC_BENCH_DETAIL"]))
    # the LAST stdout line is the compact record (benchlib/record.py: <= 4 KB, a projection of the detail); the tests below read
    # the detail, which holds everything measured
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last.encode()) <= 4096 and line["metric"] == detail["metric"] and "roofline" in line and "cpu_baseline" in line
    assert abs(line["value"] - detail["value"]) <= 1e-5 * abs(detail["value"]) and "[bench detail]" not in r.stdout
    return detail


def test_bench_self_launches_two_ranks_on_this_gpu(dev):
    """`python bench.py --gpus 2` with no launcher around it: two ranks share this one GPU (gloo for the
    collectives -- the line must say it is not an RCCL measurement), the N > 1 line carries the `hbm_regime` and `models`
    sub-records next to the headline, and every record trained (finite loss)."""
    line = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--hbm-steps", "2", "--no-cpu-baseline"], {})
    assert line["n_gpus"] == 2 and line["multi_rank_rccl_measured"] is False and line["scaling"] == "weak"
    assert line["config"]["ranks_share_devices"] is True
    assert np.isfinite(line["loss_mean"]) and line["value"] > 0
    h = line["hbm_regime"]
    assert "error" not in h and h["ms_per_step"] > 0 and np.isfinite(h["loss_mean"]) and "split" in h["launch"]
    for name in ("MMGCN", "FREEDOM"):
        m = line["models"][name]
        assert "error" not in m and m["ms_per_step"] > 0 and m["config"]["exchange_bytes_per_step_per_rank"] > 0, m


@pytest.mark.parametrize("name", ["MMGCN", "FREEDOM"])
def test_bench_model_records_count_their_spmm_work(dev, name):
    """`bench.py --model X` on one GPU: the record's `value` divides the SpMM messages of one step -- counted by patching
    `ops.spmm_raw` for one eager step -- by the step time; every caller must therefore reach the SpMM through the `ops` facade
    (a module that binds the function at import time drops out of the count: MMGCN's record read 0 after the ops split)."""
    line = _run_bench(["--model", name, "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], {})
    assert line["value"] > 0 and line["config"]["spmm_nnz_per_step_all_ranks"] > 0 and line["ms_per_step"] > 0


def _sharded_mmgcn_streams_worker(rank, world, port, tmp, streams):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_FORCE_COLLECTIVES"] = "1"          # a 1-rank group still issues every exchange through RCCL
    os.environ["CHAOREC_DIST_MMGCN_STREAMS"] = "1" if streams else "0"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from chaorec_amd import dist as cdist, graph, ops
    from chaorec_amd.Model import MMGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 6000, 2500, 40000, 512
    edges = synthetic_interactions(U, I, E, seed=3)
    g = torch.Generator().manual_seed(4)
    v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 256, generator=g)
    torch.manual_seed(21)
    full = MMGCN(U, I, edges, graph.user_item_dict_from_edges(edges), v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
    shard = cdist.UserShard(edges, U, I, world, rank, dev, self_loops=True)
    m = cdist.ShardedMMGCN(full, shard, dev)
    del full
    opt = FusedAdam(m.parameters(), lr=1e-3)
    edges_dev = torch.from_numpy(shard.local_edges.astype(np.int64)).to(dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)

    def draw():
        counter.add_(1)
        u, pos, neg = ops.draw_batch(edges_dev, m.hist, B, U, I, 42, 0, step_dev=counter, item_offset=U)
        return torch.stack((u, u), 1), torch.stack((pos, neg), 1)

    before = dict(cdist.STATS)
    step = GraphedTrainStep(m, opt, batch_fn=draw, after_backward=m.sync_grads)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    out = {n: p.detach().cpu().numpy() for n, p in m.named_parameters()}
    out["__exchanges"] = np.array(cdist.STATS["exchanges"] - before["exchanges"])
    np.savez(os.path.join(tmp, f"mm_streams_{int(streams)}.npz"), **out)
    dist.destroy_process_group()


def test_sharded_mmgcn_two_streams_captured_with_rccl(dev):
    """dist.ShardedMMGCN with its two modality branches on two streams, the exchanges of both really issued
    through RCCL (1-rank group, forced) and the whole step captured in one hipGraph: six replayed steps leave the SAME BITS
    in every parameter as the one-stream run -- in two fresh processes each.  (Round 5 saw this comparison off by 2e-4 once in
    ~10 runs and marked it xfail; round 6 found the cause in the BPR backward's fp32 atomic adds, whose order -- and with it
    the last bit of a few gradient elements, which Adam's first steps blow up -- moves with the load on the chip, on ONE
    stream as often as on two: tools/stream_stress.py, profiles/r06_stream_bisect.txt.  The backward launch is ordered now.)"""
    import tempfile
    import torch.multiprocessing as mp
    from test_gpu_dist2 import _free_port
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for rep in range(2):
            for streams in (False, True):
                mp.spawn(_sharded_mmgcn_streams_worker, args=(1, _free_port(), tmp, streams), nprocs=1, join=True)
                out[(rep, streams)] = dict(np.load(os.path.join(tmp, f"mm_streams_{int(streams)}.npz")))
    assert int(out[(0, True)]["__exchanges"]) > 0
    ref = out[(0, False)]
    for key, got in out.items():
        for n, r in ref.items():
            if not n.startswith("__"):
                assert np.array_equal(got[n], r), (key, n, float(np.abs(got[n] - r).max()))


def test_bpr_multi_backward_scatters_gathered_terms_itself(dev):
    """ops.bpr_loss_multi(..., gathered=...) (FREEDOM.loss): the backward launch adds the gradient of a projected row block
    also into its [I, D] row gradient, which ops.linear_rows' backward then takes instead of a zero fill + index_add_ of
    its own; ops.split_rows' backward returns a view when its two gradients already lie back to back.  Same loss, same
    gradients as the path without `gathered` (up to the order of the atomic adds), duplicated rows included."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    U, I, D, B, Kt = 900, 700, 64, 512, 256
    x0 = torch.randn(U + I, D, device=dev, generator=g) * 0.1
    table0 = torch.randn(I, Kt, device=dev, generator=g) * 0.1
    w0, b0 = torch.randn(D, Kt, device=dev, generator=g) * 0.05, torch.randn(D, device=dev, generator=g) * 0.01
    users = torch.randint(0, U, (B,), device=dev, generator=g)
    pos = torch.randint(0, I, (B,), device=dev, generator=g)
    neg = torch.randint(0, 40, (B,), device=dev, generator=g)          # few distinct negatives: rows repeat in the batch
    rows = torch.cat((pos, neg))
    idx = torch.arange(B, device=dev)
    wvec = torch.tensor([1.0, 0.3], device=dev)

    def run(gathered):
        x = x0.clone().requires_grad_(True)
        table, w, b = (t.clone().requires_grad_(True) for t in (table0, w0, b0))
        xu, xi = ops.split_rows(x, U)
        proj = ops.linear_rows(table, rows, w, b)
        token = proj._chaorec_row_scatter
        loss = ops.bpr_loss_multi(xu, users, ops.VARIANT_LOGSIGMOID, [(xi, pos, neg), (proj, idx, idx + B)], wvec,
                                  gathered=[None, (rows, I)] if gathered else None)
        loss.backward()
        torch.cuda.synchronize()
        # the hand-over went through the gathering node's own token (no process-wide dict): put once by the BPR
        # backward, taken by linear_rows' backward, nothing left behind
        assert token.puts == token.hits == (1 if gathered else 0) and token._held is None
        return loss.detach(), x.grad, table.grad, w.grad, b.grad

    ref, got = run(False), run(True)
    assert torch.equal(ref[0], got[0])
    for a, c in zip(ref[1:], got[1:]):
        assert torch.allclose(a, c, rtol=0, atol=2e-6 * float(a.abs().max()) + 1e-12)
    assert float(got[2].abs().sum()) > 0 and not hasattr(ops, "_SCATTERED")


def test_fused_step_migrates_existing_adam_moments(dev):
    """an optimizer that already holds moments in separate tensors (trained the tables before, or came out of
    load_state_dict) used to be refused by the fused steps ("not adjacent").  FusedAdam.make_moments_adjacent copies them
    into one buffer: a fused step built on such an optimizer continues exactly like one built on the untouched optimizer."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 1500, 700, 9000, 128
    edges = synthetic_interactions(U, I, E, seed=4)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(4):
        sel = rng.choice(E, B, replace=False)
        batches.append((torch.from_numpy(edges[sel, 0].astype(np.int64)).to(dev), torch.from_numpy(edges[sel, 1].astype(np.int64)).to(dev),
                        torch.from_numpy(rng.integers(U, U + I, B)).to(dev)))

    def run(scatter_state):
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, None, 64, 1e-3, 2, "add", dev).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-2)
        for b in batches[:2]:                          # two ordinary steps: the optimizer now has moments
            opt.zero_grad()
            m.loss(*b).backward()
            opt.step()
        if scatter_state:                              # ... and they no longer sit back to back (as after load_state_dict)
            for p in m.parameters():
                for k in ("exp_avg", "exp_avg_sq"):
                    opt.state[p][k] = opt.state[p][k].clone()
        step = FusedLightGCNStep(m, opt, batch_size=B, given_batch=True, capture=False)
        for b in batches[2:]:
            step(*b)
        torch.cuda.synchronize()
        return m._flat.detach().clone(), opt.state[m.user_embedding.weight]["exp_avg"].detach().clone()

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=0, atol=2e-6)


def test_bench_sharded_path_with_rccl_on_one_rank(dev):
    """The N > 1 code path of bench.py with RCCL itself (a 1-rank group, every exchange forced through it): the node probe
    runs in a child job before the parent touches the GPU, the step is captured with its collectives, the line says what was
    used."""
    line = _run_bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--no-hbm-regime", "--no-models", "--no-cpu-baseline"],
                      {"CHAOREC_FORCE_SHARDED": "1", "CHAOREC_FORCE_COLLECTIVES": "1"})
    cfg = line["config"]
    assert cfg["node_probe"]["graph"] is True and cfg["node_probe"]["stages"]["allreduce_replay"] is True
    assert "captured hipGraph" in cfg["launch"] and "over nccl" in cfg["parallelism"]
    assert line["multi_rank_rccl_measured"] is False and np.isfinite(line["loss_mean"])
    assert cfg["exposed_communication"]["exchanges_per_step"] == 7


def _bits_to_rows(bits, n):
    w = bits.cpu().numpy().view(np.uint32)
    return np.nonzero(np.unpackbits(w.view(np.uint8), bitorder="little")[:n])[0]


@pytest.mark.parametrize("D", [64, 128, 256])
def test_rowsparse_spmm_equals_the_dense_launch_bit_for_bit(dev, D):
    """chaorec_spmm_csr_rowsparse_f32: with the source's (and z's) non-zero rows flagged in bitmaps, the launch skips the
    gathers of unflagged rows and gives the dense launch's bits; out_bits is a superset of the output's non-zero rows; the
    chain source -> output -> next launch (the first two backward propagates of a training step) stays exact.  Heavy-tailed
    graph: inline entries, the remainder loop and the cooperative long-row walk are all on the path."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 3000, 900, 30000                              # (item degrees up to several hundred: long rows)
    edges = synthetic_interactions(U, I, E, seed=11)
    N = U + I
    csr = graph.lightgcn_csr(edges, N).to(dev)
    gen = torch.Generator(device=dev).manual_seed(D)
    rows = torch.randperm(N, device=dev, generator=gen)[:200]
    G = torch.zeros(N, D, device=dev)
    G[rows] = torch.randn(200, D, device=dev, generator=gen)
    bits0 = ops.row_bitmap(N, dev)
    w = bits0.cpu().numpy().view(np.uint32)
    for r in rows.cpu().tolist():
        w[r >> 5] |= np.uint32(1 << (r & 31))
    bits0.copy_(torch.from_numpy(w.view(np.int32)))
    dense1 = ops.spmm_raw(csr, G, alpha=0.25, z=G, beta=0.25)
    bits1 = ops.row_bitmap(N, dev)
    y1 = torch.full((N, D), 7.0, device=dev)                 # (stale contents: every row must be written)
    ops.spmm_rowsparse_raw(csr, G, y1, alpha=0.25, z=G, beta=0.25, src_bits=bits0, z_bits=bits0, out_bits=bits1)
    assert torch.equal(y1, dense1)
    nz = torch.nonzero(dense1.abs().sum(1) > 0).flatten().cpu().numpy()
    flagged = _bits_to_rows(bits1, N)
    assert np.isin(nz, flagged).all() and len(flagged) < N   # a superset, and a real restriction
    dense2 = ops.spmm_raw(csr, dense1, z=G, beta=0.25)
    y2 = torch.full((N, D), -3.0, device=dev)
    ops.spmm_rowsparse_raw(csr, y1, y2, z=G, beta=0.25, src_bits=bits1, z_bits=bits0)
    assert torch.equal(y2, dense2)
    # the frontier form the fused step uses: row masks from expand_row_bits (N1 = R0 + nbr(R0), N2 = nbr(N1)); rows outside a
    # mask walk no entries -- left unwritten (first launch: its only reader gathers flagged rows) or written as zeros (second)
    n1, n2 = ops.row_bitmap(N, dev), ops.row_bitmap(N, dev)
    ops.expand_row_bits(csr, bits0, n1)
    ops.expand_row_bits(csr, n1, n2)
    r0, r1, r2 = _bits_to_rows(bits0, N), _bits_to_rows(n1, N), _bits_to_rows(n2, N)
    rp, cl = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()
    want1 = np.unique(np.concatenate([r0] + [cl[rp[r]:rp[r + 1]] for r in r0]))
    want2 = np.unique(np.concatenate([r1] + [cl[rp[r]:rp[r + 1]] for r in r1]))
    assert np.array_equal(r1, want1) and np.array_equal(r2, want2) and np.isin(nz, r1).all()
    f1 = torch.full((N, D), 9.0, device=dev)
    ops.spmm_rowsparse_raw(csr, G, f1, alpha=0.25, z=G, beta=0.25, src_bits=bits0, z_bits=bits0, row_bits=n1, write_zeros=False)
    inside = torch.zeros(N, dtype=torch.bool, device=dev)
    inside[torch.from_numpy(r1).to(dev)] = True
    assert torch.equal(f1[inside], dense1[inside]) and bool((f1[~inside] == 9.0).all()) and bool((dense1[~inside] == 0).all())
    f2 = torch.full((N, D), 5.0, device=dev)
    ops.spmm_rowsparse_raw(csr, f1, f2, z=G, beta=0.25, src_bits=n1, z_bits=bits0, row_bits=n2, write_zeros=True)
    assert torch.equal(f2, dense2)
    if True:
        # ... and the first launch as the fused step issues it at BASELINE configs[4]: the frontier as a LIST (emitted by the
        # expansion), one lane group per listed row -- the listed rows get the dense launch's bits, no other row is touched
        lst, ln, nb = torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), ops.row_bitmap(N, dev)
        ops.expand_row_bits(csr, bits0, nb, lst, ln)
        assert torch.equal(nb, n1) and np.array_equal(np.sort(lst[:int(ln)].cpu().numpy()), r1)
        fl = torch.full((N, D), 9.0, device=dev)
        ops.spmm_rowlist_raw(csr, G, fl, lst, ln, alpha=0.25, z=G, beta=0.25, src_bits=bits0, z_bits=bits0)
        assert torch.equal(fl[inside], dense1[inside]) and bool((fl[~inside] == 9.0).all())
        ops.spmm_rowsparse_raw(csr, fl, f2, z=G, beta=0.25, src_bits=n1, z_bits=bits0, row_bits=n2, write_zeros=True)
        assert torch.equal(f2, dense2)
    # no bitmaps at all = the dense launch; src bitmap only / z bitmap only
    y3 = torch.empty((N, D), device=dev)
    ops.spmm_rowsparse_raw(csr, G, y3, alpha=0.25, z=G, beta=0.25, src_bits=bits0)
    assert torch.equal(y3, dense1)
    ops.spmm_rowsparse_raw(csr, G, y3, alpha=0.25, z=G, beta=0.25, z_bits=bits0)
    assert torch.equal(y3, dense1)


def test_fused_step_with_the_rowsparse_backward_trains_like_the_dense_one(dev, monkeypatch):
    """optim.FusedLightGCNStep with the first two backward propagates in their row-sparse form (the BPR launch flags the
    rows of the batch gradient, the Adam launch clears the bitmaps): the same batches give the same tables as the dense
    backward up to the order of the BPR launch's atomic adds, eager and captured, and the bitmaps are clean after a step."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B = 4000, 1500, 30000, 256
    edges = synthetic_interactions(U, I, E, seed=5)
    rng = np.random.default_rng(1)
    batches = []
    for _ in range(5):
        sel = rng.choice(E, B, replace=False)
        batches.append((torch.from_numpy(edges[sel, 0].astype(np.int64)).to(dev), torch.from_numpy(edges[sel, 1].astype(np.int64)).to(dev),
                        torch.from_numpy(rng.integers(U, U + I, B)).to(dev)))

    def run(sparse, capture):
        monkeypatch.setenv("CHAOREC_SPARSE_BACKWARD", "1" if sparse else "0")
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
        step = FusedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=capture,
                                 light_forward=False)
        assert step.sparse_bwd == sparse and not step.light
        losses = [float(step(*b)) for b in batches]
        torch.cuda.synchronize()
        if sparse:
            assert int(step._bits_all.abs().sum()) == 0
            assert float(step.G.abs().max()) == 0.0
        return m._flat.detach().clone(), losses

    ref, ref_losses = run(False, False)
    for capture in (False, True):
        got, losses = run(True, capture)
        d = (got - ref).abs()
        assert float((d > 2e-6).float().mean()) <= 1e-4 and float(d.median()) <= 1e-7, (capture, float(d.max()))
        assert np.allclose(losses, ref_losses, rtol=1e-5)


def test_bpr_launch_flags_exactly_the_rows_it_touched(dev):
    from chaorec_amd import ops
    gen = torch.Generator(device=dev).manual_seed(2)
    U, I, D, B = 700, 500, 64, 128
    tab = torch.randn(U + I, D, device=dev, generator=gen) * 0.1
    G = torch.zeros_like(tab)
    ids = (torch.randint(0, U, (B,), device=dev, generator=gen), torch.randint(0, I, (B,), device=dev, generator=gen),
           torch.randint(0, I, (B,), device=dev, generator=gen))
    bits = ops.row_bitmap(U + I, dev)
    ops.bpr_fwd_bwd(tab, U, G, B, ops.VARIANT_LOG_SIGMOID_EPS, 1e-3, torch.empty(B, device=dev), torch.empty(4 * B, device=dev), ids,
                    row_bits=bits)
    torch.cuda.synchronize()
    want = np.unique(np.concatenate([ids[0].cpu().numpy(), U + ids[1].cpu().numpy(), U + ids[2].cpu().numpy()]))
    assert np.array_equal(_bits_to_rows(bits, U + I), want)
    assert np.array_equal(torch.nonzero(G.abs().sum(1) > 0).flatten().cpu().numpy(), want)


@pytest.mark.parametrize("D,L,long_t", [(64, 3, None), (128, 2, 6), (256, 4, 3), (128, 3, 20), (64, 2, 2), (96, 2, 5), (192, 3, 4)])
def test_rowlist_layer_mean_equals_the_dense_forward_in_the_listed_rows(dev, D, L, long_t):
    """The light forward's last two launches (ops.spmm_rowlist_raw over N1's list, then over R0's list with the layer mean in
    the epilogue) against ops.forward_layers on every row: the listed rows of the mean carry the same bits -- whichever of the
    dense forward's two accumulation forms (all terms in the last epilogue / one read-modify-write per layer) D and L select.
    long_t: listed rows with more entries go through the workgroup-per-row launch (thresholds far below the product's 1024,
    so that most rows of this small graph take it)."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 3000, 1100, 26000
    edges = synthetic_interactions(U, I, E, seed=11)
    csr = graph.lightgcn_csr(edges, U + I).to(dev)
    N = U + I
    gen = torch.Generator(device=dev).manual_seed(D + L)
    x0 = torch.randn(N, D, device=dev, generator=gen) * 0.1
    want = torch.empty_like(x0)
    ops.forward_layers(csr, x0, L, want, [torch.empty_like(x0) for _ in range(L - 1)])
    # R0: 200 random rows; N1 by the expansion kernel
    bits0, bits1 = ops.row_bitmap(N, dev), ops.row_bitmap(N, dev)
    r0 = torch.randperm(N, device=dev, generator=gen)[:200]
    ids = (r0[:80].clamp(max=U - 1), (r0[80:140] % I), (r0[140:] % I)[:60])
    ids = tuple(torch.cat([t, t.new_zeros(80 - t.numel())]) if t.numel() < 80 else t for t in ids)
    list0, n0 = torch.empty(3 * 80, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.batch_rows(ids, bits0, U, list0, n0)
    list1, n1 = torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.expand_row_bits(csr, bits0, bits1, list1, n1)
    rows0 = np.unique(np.concatenate([ids[0].cpu().numpy(), U + ids[1].cpu().numpy(), U + ids[2].cpu().numpy()]))
    assert np.array_equal(np.sort(list0[:int(n0)].cpu().numpy()), rows0) and np.array_equal(_bits_to_rows(bits0, N), rows0)
    xs = [x0]
    for l in range(L - 2):
        xs.append(ops.spmm_raw(csr, xs[-1], y=torch.empty_like(x0)))
    long_rows = ops.long_row_buffers(csr, long_t) if long_t else None
    stale = torch.full_like(x0, float("nan"))                  # rows outside N1 are never written -- and never read
    ops.spmm_rowlist_raw(csr, xs[-1], stale, list1, n1, long_rows=long_rows)
    xs.append(stale)
    got = torch.full_like(x0, float("nan"))
    ops.spmm_rowlist_raw(csr, xs[-1], None, list0, n0, mean_out=got, mean_terms=xs, mean_w=1.0 / (L + 1), long_rows=long_rows)
    torch.cuda.synchronize()
    if long_rows is not None:
        deg = (csr.rowptr[1:] - csr.rowptr[:-1]).cpu().numpy()
        assert (deg[rows0] > long_t).any() and int(long_rows[1].abs().sum()) == 0       # used, and left clean
    rows = torch.from_numpy(rows0).to(dev)
    assert torch.equal(got[rows], want[rows])
    rest = torch.ones(N, dtype=torch.bool, device=dev)
    rest[rows] = False
    assert bool(torch.isnan(got[rest]).all())


def test_batch_rows_draws_the_bpr_launch_s_batch(dev):
    """ops.batch_rows with edges = the triples ops.bpr_fwd_bwd draws in its own launch for the same seed / step / counter /
    permutation cursor (a light step draws first and hands the ids over)."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B, D = 900, 400, 7000, 128, 64
    edges = synthetic_interactions(U, I, E, seed=3)
    ed = torch.from_numpy(edges.astype(np.int64)).to(dev).contiguous()
    hist = tuple(t.to(dev) for t in graph.user_hist_csr_from_edges(edges, U))
    tab = torch.randn(U + I, D, device=dev) * 0.1
    step_dev = torch.tensor([5], dtype=torch.int64, device=dev)
    perm = torch.randperm(E, device=dev)
    perm_pos = torch.tensor([256], dtype=torch.int64, device=dev)
    for use_perm in (False, True):
        kw = dict(perm=perm, perm_pos=perm_pos, pos_offset=B) if use_perm else {}
        a = tuple(torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(3))
        b = tuple(torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(3))
        bits = ops.row_bitmap(U + I, dev)
        ops.batch_rows(a, bits, U, edges=ed, hist=hist, num_user=U, num_item=I, seed=77, step=2, step_dev=step_dev, **kw)
        ops.bpr_fwd_bwd(tab, U, torch.zeros_like(tab), B, ops.VARIANT_LOG_SIGMOID_EPS, 1e-3, torch.empty(B, device=dev),
                        torch.empty(4 * B, device=dev), b, edges=ed, hist=hist, num_user=U, num_item=I, seed=77, step=2,
                        step_dev=step_dev, **kw)
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        want = np.unique(np.concatenate([a[0].cpu().numpy(), U + a[1].cpu().numpy(), U + a[2].cpu().numpy()]))
        assert np.array_equal(_bits_to_rows(bits, U + I), want)


@pytest.mark.parametrize("L,capture", [(3, False), (3, True), (2, True), (4, True)])
def test_light_steps_train_like_full_steps_bit_for_bit(dev, L, capture):
    """optim.FusedLightGCNStep with the light forward (the batch drawn first; layers L-1 and L over N1's / R0's row lists
    only) against the same step with every row of every layer: batches without a repeated row make the BPR launch's atomic
    adds order-free, so loss and tables must agree BIT FOR BIT after every step; model.result is withheld after a light step
    (gene_ranklist says why), and the full_result step that precedes an evaluation leaves the full step's table."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B, D, T = 5000, 1800, 40000, 200, 128, 4
    edges = synthetic_interactions(U, I, E, seed=5)
    rng = np.random.default_rng(L)
    batches = []
    for _ in range(T):
        u = rng.choice(U, B, replace=False)
        it = rng.choice(I, 2 * B, replace=False)
        batches.append(tuple(torch.from_numpy(a.astype(np.int64)).to(dev) for a in (u, U + it[:B], U + it[B:])))

    def run(light):
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, None, D, 1e-3, L, "add", dev).to(dev)
        step = FusedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=capture,
                                 light_forward=light)
        assert step.light == light and step.sparse_bwd == light
        out = []
        for t, b in enumerate(batches):
            last = t == T - 1
            loss = float(step(*b, full_result=last))
            out.append((loss, m._flat.detach().clone()))
            if light and not last:
                assert m.result is None and not step.result_complete
                with pytest.raises(RuntimeError, match="light"):
                    m.gene_ranklist(topk=5)
        torch.cuda.synchronize()
        if light:
            assert int(step._bits_all.abs().sum()) == 0 and float(step.G.abs().max()) == 0.0
        return out, m.result.detach().clone(), m.gene_ranklist(topk=10)

    ref, ref_result, ref_rank = run(False)
    got, result, rank = run(True)
    for (la, xa), (lb, xb) in zip(ref, got):
        assert la == lb and torch.equal(xa, xb)
    assert torch.equal(result, ref_result) and torch.equal(rank, ref_rank)


def test_light_run_ends_with_a_full_step(dev):
    """FusedLightGCNStep.run(n): n - 1 light steps (k-step replays + singles) and one full step -- an epoch of the product
    loop (chaorec_amd/train_and_evaluate.py), after which gene_ranklist ranks the table of the last training forward as the
    reference does.  Drawn batches: same counters, same draws, the same trajectory as full steps up to atomic-add order."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, B, D = 5000, 1800, 40000, 256, 64
    edges = synthetic_interactions(U, I, E, seed=6)
    ed = torch.from_numpy(edges.astype(np.int64)).to(dev).contiguous()

    def run(light):
        torch.manual_seed(0)
        m = LightGCN(U, I, edges, None, D, 1e-3, 3, "add", dev).to(dev)
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        acc = torch.zeros(1, dtype=torch.float32, device=dev)
        step = FusedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, edges=ed, seed=9, step_dev=counter,
                                 loss_accum=acc, steps_per_replay=3, light_forward=light)
        step.run(11)
        torch.cuda.synchronize()
        assert int(counter) == 11 and step.result_complete and m.result is not None
        return m._flat.detach().clone(), float(acc), m.result.detach().clone()

    (xa, la, ra), (xb, lb, rb) = run(False), run(True)
    d = (xa - xb).abs()
    assert float((d > 2e-6).float().mean()) <= 1e-4 and float(d.median()) <= 1e-7
    assert la == pytest.approx(lb, rel=1e-5)
    assert float((ra - rb).abs().max()) <= 1e-5


@pytest.mark.parametrize("D,frac", [(64, 0.02), (128, 0.3), (256, 1.0)])
def test_gated_rowlist_with_long_rows_equals_the_dense_launch(dev, D, frac):
    """ops.spmm_rowlist_raw with src_bits AND long_rows (the backward's first propagate on a graph with popular items: rows
    above the threshold go to the workgroup-per-row scan-and-queue launch) against the dense launch on the same row-sparse
    operand: the listed rows bit for bit -- few flagged sources per row, many, and ALL of them (rows of several thousand flagged
    entries overflow the launch's 2048-entry queue: flushed in order)."""
    from chaorec_amd import graph, ops
    rng = np.random.default_rng(int(D + 100 * frac))
    U, I = 6000, 900
    deg_i = np.minimum((rng.pareto(1.1, I) * 6 + 2).astype(np.int64), U - 1)
    deg_i[:3] = [5200, 3100, 2300]                     # (popular items)
    rows = np.repeat(np.arange(I), deg_i)
    cols = np.concatenate([rng.choice(U, d, replace=False) for d in deg_i])
    edges = np.stack([cols, U + rows], 1)
    csr = graph.lightgcn_csr(edges, U + I).to(dev)
    N = U + I
    gen = torch.Generator(device=dev).manual_seed(3)
    flagged = torch.rand(N, device=dev, generator=gen) < frac
    x = torch.randn(N, D, device=dev, generator=gen) * flagged[:, None]
    z = torch.randn(N, D, device=dev, generator=gen) * flagged[:, None]
    bits = ops.row_bitmap(N, dev)
    idx = torch.nonzero(flagged).flatten()
    bits_np = np.zeros(bits.numel(), dtype=np.uint32)
    np.bitwise_or.at(bits_np, idx.cpu().numpy() >> 5, np.uint32(1) << (idx.cpu().numpy() & 31).astype(np.uint32))
    bits.copy_(torch.from_numpy(bits_np.view(np.int32)))
    want = ops.spmm_raw(csr, x, y=torch.empty_like(x), alpha=0.25, z=z, beta=0.5)
    listed = torch.randperm(N, device=dev, generator=gen)[:N // 2]
    listed = torch.unique(torch.cat([listed, torch.arange(U, U + 3, device=dev)]))        # (the popular items are in)
    row_list = listed.to(torch.int32)
    list_n = torch.tensor([listed.numel()], dtype=torch.int32, device=dev)
    for long_t in (None, 40):
        long_rows = ops.long_row_buffers(csr, long_t) if long_t else None
        got = torch.full_like(x, float("nan"))
        ops.spmm_rowlist_raw(csr, x, got, row_list, list_n, alpha=0.25, z=z, beta=0.5, src_bits=bits, z_bits=bits, long_rows=long_rows)
        torch.cuda.synchronize()
        assert torch.equal(got[listed], want[listed]), long_t
        rest = torch.ones(N, dtype=torch.bool, device=dev)
        rest[listed] = False
        assert bool(torch.isnan(got[rest]).all())
        if long_rows is not None:
            assert int(long_rows[1].abs().sum()) == 0


@pytest.mark.parametrize("E", [10300, 10240])
def test_product_epoch_with_light_steps_equals_the_dense_epoch(dev, monkeypatch, E):
    """chaorec_amd/train_and_evaluate.py's epoch (in-launch batches from the epoch permutation, k-step replays, the short last
    batch through the ordinary path) with the light step forced on (CHAOREC_SPARSE_BACKWARD=1: row-sparse backward + light
    forward) against the same epoch on dense launches: the same epoch loss and parameters up to atomic-add order, and
    gene_ranklist afterwards ranks a COMPLETE table in both -- with a short last batch (E = 10300: its ordinary forward leaves
    the table) and without one (E = 10240: the epoch's last fused step is a full one)."""
    from chaorec_amd import graph, dataload
    from chaorec_amd import train_and_evaluate as tae
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, B = 1500, 900, 1024
    edges = synthetic_interactions(U, I, E, seed=5)
    uid = graph.user_item_dict_from_edges(edges)

    def epoch(light):
        monkeypatch.setenv("CHAOREC_SPARSE_BACKWARD", "1" if light else "0")
        torch.manual_seed(1)
        m = LightGCN(U, I, edges, uid, 64, 1e-3, 3, "add", dev).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-3)
        ld = dataload.DeviceBatchSampler(U, I, uid, edges, B, dev, "LightGCN", 7)
        graphed = tae._capture_step(m, ld, opt, "LightGCN")
        assert getattr(graphed, "fused", False) and bool(getattr(graphed, "light", False)) == light
        ld.gen.manual_seed(7)
        ld.global_step = 0
        ld.step_dev.zero_()
        losses = [tae.train(m, ld, opt, "LightGCN", graphed) for _ in range(2)]
        assert m.result is not None
        return losses, m._flat.detach().clone(), m.result.detach().clone(), m.gene_ranklist(topk=20)

    (la, xa, ra, ka), (lb, xb, rb, kb) = epoch(False), epoch(True)
    assert lb == pytest.approx(la, rel=1e-5)
    d = (xa - xb).abs()
    assert float((d > 2e-6).float().mean()) <= 1e-4 and float(d.median()) <= 1e-7
    assert float((ra - rb).abs().max()) <= 1e-5
    assert float((ka != kb).float().mean()) <= 2e-3        # (near-ties may swap under 1e-7 differences)


def test_frontier_helper_launches(dev):
    """The small launches of the frontier-restricted sharded step, one by one against numpy: the expansion over a RECTANGULAR
    block (bits over the rows in, bits over the columns out, the other side's batch rows as bits_self, the list = exactly the
    newly flagged columns), the list of a bitmap's rows (bits past the end ignored), the layer mean / zero fill / copy of flagged
    rows only, the union of gathered bitmaps."""
    from chaorec_amd import graph, ops
    rng = np.random.default_rng(12)
    U, I, D = 1000, 333, 64
    ul, il = rng.integers(0, U, 6000), rng.integers(0, I, 6000)
    pairs = np.unique(np.stack([ul, il], 1), axis=0)
    ui = graph.coo_to_csr(torch.from_numpy(pairs[:, 0]), torch.from_numpy(pairs[:, 1]), torch.ones(len(pairs)), U, I).to(dev)

    def bitmap(rows, n):
        b = np.zeros((n + 31) // 32 + 1, dtype=np.uint32)
        np.bitwise_or.at(b, rows >> 5, np.uint32(1) << (rows & 31).astype(np.uint32))
        return torch.from_numpy(b.view(np.int32)).to(dev)

    users = rng.choice(U, 40, replace=False)
    items_self = rng.choice(I, 25, replace=False)
    bu, bself, bout = bitmap(users, U), bitmap(items_self, I), ops.row_bitmap(I, dev)
    lst, n = torch.zeros(I, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.expand_row_bits(ui, bu, bout, lst, n, bits_self=bself)
    want = np.unique(np.concatenate([items_self, pairs[np.isin(pairs[:, 0], users), 1]]))
    assert np.array_equal(_bits_to_rows(bout, I), want) and np.array_equal(np.sort(lst[:int(n)].cpu().numpy()), want)
    with pytest.raises(ValueError, match="bits_self"):
        ops.expand_row_bits(ui, bu, bout, lst, n)
    # rows_list_from_bits (a stray bit past n_rows is not a row)
    stray = bout.clone()
    stray[I // 32] |= 1 << ((I % 32) + 1 if I % 32 < 30 else 31)
    lst2, n2 = torch.zeros(I, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.rows_list_from_bits(stray, I, lst2, n2)
    assert int(n2) == len(want) and np.array_equal(np.sort(lst2[:int(n2)].cpu().numpy()), want)
    # rows_mean_by_bits / zero_rows_by_bits / rows_copy_by_bits
    terms = [torch.randn(I, D, device=dev) for _ in range(4)]
    out = torch.full((I, D), float("nan"), device=dev)
    ops.rows_mean_by_bits(terms, 0.25, out, bout)
    ref = torch.empty(I, D, device=dev)
    ops.rows_mean(terms, 0.25, ref)
    rows = torch.from_numpy(want).to(dev)
    rest = torch.ones(I, dtype=torch.bool, device=dev)
    rest[rows] = False
    assert torch.equal(out[rows], ref[rows]) and bool(torch.isnan(out[rest]).all())
    src, dst = torch.randn(I, D, device=dev), torch.zeros(I, D, device=dev)
    ops.rows_copy_by_bits(dst, src, bout)
    assert torch.equal(dst[rows], src[rows]) and float(dst[rest].abs().max()) == 0.0
    ops.zero_rows_by_bits(src, bout)
    assert float(src[rows].abs().max()) == 0.0 and float(src[rest].abs().min()) > 0.0
    # frontier_pack / frontier_unpack: bitmap order, zeroed tail, the inverse copy
    tab = torch.randn(I, D, device=dev)
    cap = len(want) + 17
    compact = torch.full((cap, D), float("nan"), device=dev)
    prefix = torch.zeros((I + 31) // 32 + 1, dtype=torch.int32, device=dev)
    ops.frontier_pack(tab, stray, prefix, compact)                      # (the stray bit past the end is not a row)
    assert int(prefix[-1]) == len(want) and torch.equal(compact[:len(want)], tab[rows])
    assert float(compact[len(want):].abs().max()) == 0.0
    back = torch.zeros(I, D, device=dev)
    ops.frontier_unpack(back, stray, prefix, compact * 2)
    assert torch.equal(back[rows], 2 * tab[rows]) and float(back[rest].abs().max()) == 0.0
    # more flagged rows than the compact buffer holds: loud when eager, recorded in a device flag when asked to
    small = torch.empty((len(want) - 3, D), device=dev)
    with pytest.raises(RuntimeError, match="do not fit"):
        ops.frontier_pack(tab, stray, prefix, small)
    over = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.frontier_pack(tab, stray, prefix, small, overflow=over)
    ops.frontier_pack(tab, stray, prefix, compact, overflow=over)        # (sticky: a later call that fits does not clear it)
    assert int(over) == 3 and torch.equal(small, tab[rows][:len(want) - 3])
    # or_words
    parts = torch.stack([bitmap(rng.choice(I, 30, replace=False), I) for _ in range(4)])
    acc = torch.zeros_like(parts[0])
    ops.or_words(acc, parts)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), np.bitwise_or.reduce(parts.cpu().numpy().view(np.uint32), axis=0))


def test_frontier_launches_on_empty_frontiers(dev):
    """Edge cases of the frontier-restricted step's launches: an EMPTY row list / bitmap is a no-op everywhere (a list launch
    -- with and without long rows, gated and ungated -- leaves y alone and its long-row counters clean; the expansion flags
    nothing; the compact frontier buffer comes back all-zero), and a batch whose size is no multiple of the workgroup is
    flagged completely."""
    from chaorec_amd import graph, ops
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, D = 900, 400, 128
    edges = synthetic_interactions(U, I, 7000, seed=3)
    csr = graph.lightgcn_csr(edges, U + I).to(dev)
    N = U + I
    x = torch.randn(N, D, device=dev)
    empty_bits, out_bits = ops.row_bitmap(N, dev), ops.row_bitmap(N, dev)
    lst, n = torch.zeros(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.expand_row_bits(csr, empty_bits, out_bits, lst, n)
    assert int(n) == 0 and int(out_bits.abs().sum()) == 0
    long_rows = ops.long_row_buffers(csr, 4)
    for kw in (dict(), dict(long_rows=long_rows), dict(src_bits=empty_bits, z_bits=empty_bits, z=x, beta=0.5, long_rows=long_rows)):
        y = torch.full((N, D), 7.0, device=dev)
        ops.spmm_rowlist_raw(csr, x, y, lst, n, **kw)
        torch.cuda.synchronize()
        assert float((y - 7.0).abs().max()) == 0.0 and int(long_rows[1].abs().sum()) == 0
    compact = torch.full((16, D), float("nan"), device=dev)
    prefix = torch.zeros((N + 31) // 32 + 1, dtype=torch.int32, device=dev)
    ops.frontier_pack(x, empty_bits, prefix, compact)
    assert int(prefix[-1]) == 0 and float(compact.abs().max()) == 0.0
    back = x.clone()
    ops.frontier_unpack(back, empty_bits, prefix, compact)
    assert torch.equal(back, x)
    for B in (1, 37, 300):
        ids = (torch.randint(0, U, (B,), device=dev), torch.randint(0, I, (B,), device=dev), torch.randint(0, I, (B,), device=dev))
        bits = ops.row_bitmap(N, dev)
        l0, n0 = torch.zeros(3 * B, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        ops.batch_rows(ids, bits, U, l0, n0)
        want = np.unique(np.concatenate([ids[0].cpu().numpy(), U + ids[1].cpu().numpy(), U + ids[2].cpu().numpy()]))
        assert np.array_equal(_bits_to_rows(bits, N), want) and np.array_equal(np.sort(l0[:int(n0)].cpu().numpy()), want)
