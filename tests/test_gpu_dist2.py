"""Two ranks on ONE GPU (gloo for the collectives, the HIP kernels for the compute): dist.FusedShardedLightGCNStep at
world size 2 on real kernels -- joined shard graphs over different user ranges, padded exchange buffers, the 1 / world
factors in the SpMM epilogues, Adam in the user-row epilogue and the fused launch on the replicated item rows -- against
optim.FusedLightGCNStep on the whole graph in the parent process.  (RCCL itself needs one GPU per rank: the driver's
multi-GPU run is the first place it meets this code; what is checked here is everything around the collective.)"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

U, I, E, D, B, T = 2400, 901, 16000, 64, 192, 3


def _problem():
    from chaorec_amd.synthetic import synthetic_interactions
    edges = synthetic_interactions(U, I, E, seed=6)
    g = torch.Generator().manual_seed(8)
    x0 = (torch.rand(U + I, D, generator=g) * 2 - 1) * 0.05
    rng = np.random.default_rng(2)
    batches = []
    for t in range(T):
        per_rank = []
        for r in range(2):
            sel = rng.choice(E, B, replace=False)
            per_rank.append((edges[sel, 0].astype(np.int64), edges[sel, 1].astype(np.int64), rng.integers(U, U + I, B)))
        batches.append(per_rank)
    return edges, x0, batches


def _worker(rank, world, port, tmp, L, exchange="allreduce", split=False, sparse=False, light=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = exchange
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    edges, x0, batches = _problem()
    deg = np.bincount(edges[:, 0], minlength=U)
    bounds = cdist.partition_users_by_nnz(deg, world)
    mine = edges[(edges[:, 0] >= bounds[rank]) & (edges[:, 0] < bounds[rank + 1])]
    shard = cdist.UserShard.from_local(mine, bounds, I, world, rank, dev)
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, L, dev, seed=1).to(dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(x0[shard.u0:shard.u1])
        m.item_embedding.weight.copy_(x0[U:])
    step = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=False,
                                          split=split, sparse_bwd=sparse, light_forward=light)
    assert step.split == split and step.sparse_bwd == sparse and step.light == light
    losses = []
    for t in range(T):
        # this rank's batch: the triples of BOTH ranks' draws whose user it owns would change the batch size; instead
        # every rank gets its own B triples of users it owns (re-drawn from its local edges with the problem's seed)
        rng = np.random.default_rng(100 * t + rank)
        sel = rng.choice(len(shard.local_edges), B, replace=False)
        users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64)).to(dev)
        pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64)).to(dev)
        neg = torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + I, B)).to(dev)
        losses.append(float(step(users, pos, neg, full_result=(t == T - 1))))
        assert (m.result_u is None) == (light and t < T - 1)
    torch.cuda.synchronize()
    assert float(step.G.abs().max()) == 0.0
    if light:
        assert float(step.Z0.abs().max()) == 0.0
    if sparse:          # what the step leaves behind for the next one: no flag, no listed row, an all-zero frontier buffer
        assert float(step.Z.abs().max()) == 0.0 and int(step._bits_all.abs().max()) == 0
        assert float(step.S[step.U:].abs().max()) == 0.0
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), u0=shard.u0, u1=shard.u1, xu=m.user_embedding.weight.detach().cpu().numpy(),
             xi=m.item_embedding.weight.detach().cpu().numpy(), fu=m.result_u.cpu().numpy(), fi=m.result_i.cpu().numpy(),
             losses=np.array(losses), n_local_edges=len(shard.local_edges))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _p2p_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = "p2p"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    ok = True
    for rows, D, reps in ((world * 300, 64, 5), (world * 7, 128, 3), (world * 300, 64, 2)):
        for rep in range(reps):
            g = torch.Generator().manual_seed(1000 * rows + 10 * rep + rank)
            mine = torch.randn(rows, D, generator=g)
            want = mine.clone()
            dist.all_reduce(want)                                   # gloo, on the host: the reference sum
            # the p2p sum adds the ranks in rank order: bit-identical to a rank-ordered host sum
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            ordered = parts[0].clone()
            for q in parts[1:]:
                ordered += q
            buf = mine.to(dev)
            cdist._sum_exchange_async(buf, None).wait()
            torch.cuda.synchronize()
            got = buf.cpu()
            ok = ok and torch.equal(got, ordered) and torch.allclose(got, want, rtol=0, atol=1e-5)
    px = cdist.P2PExchange.of(None)
    np.savez(os.path.join(tmp, f"p2p{rank}.npz"), ok=ok, boxes=len(px.boxes), got=got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _rccl_worker(rank, world, port, tmp, exchange, direct_capture, split=False, sparse=False, light=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = exchange
    os.environ["CHAOREC_DIST_DIRECT_CAPTURE"] = direct_capture
    os.environ["CHAOREC_FORCE_COLLECTIVES"] = "1"          # a 1-rank group still issues every exchange
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    edges, x0, _ = _problem()
    bounds = [0, U]
    shard = cdist.UserShard.from_local(edges, bounds, I, world, rank, dev)
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, 3, dev, seed=1).to(dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(x0[:U])
        m.item_embedding.weight.copy_(x0[U:])
    step = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=True,
                                          split=split, sparse_bwd=sparse, light_forward=light)
    losses = []
    for t in range(T):
        rng = np.random.default_rng(100 * t)
        sel = rng.choice(len(shard.local_edges), B, replace=False)
        users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64)).to(dev)
        pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64)).to(dev)
        neg = torch.from_numpy(rng.integers(U, U + I, B)).to(dev)
        losses.append(float(step(users, pos, neg, full_result=(t == T - 1))))
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, f"rccl_{exchange}_{direct_capture}_{int(split)}_{int(sparse) + int(light)}.npz"), xu=m.user_embedding.weight.detach().cpu().numpy(),
             xi=m.item_embedding.weight.detach().cpu().numpy(), losses=np.array(losses), used=cdist.exchange_mode_used())
    dist.destroy_process_group()


def test_captured_exchanges_on_a_one_rank_rccl_group():
    """The fused sharded step CAPTURED in a hipGraph with its exchanges really issued through RCCL (1-rank group,
    CHAOREC_FORCE_COLLECTIVES=1): all-reduce, the hand-written peer-to-peer exchange (barriers = one-element RCCL
    all-reduces inside the graph), and `direct` run as p2p under capture -- three Adam steps, the same tables."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for exchange, dc, split, sparse in (("allreduce", "rs_ag", False, False), ("p2p", "rs_ag", False, False),
                                            ("direct", "p2p", False, False), ("direct", "rs_ag", False, False),
                                            ("allreduce", "rs_ag", True, False), ("p2p", "rs_ag", True, False),
                                            ("rs_ag", "rs_ag", True, False), ("allreduce", "rs_ag", True, True),
                                            ("p2p", "rs_ag", True, True), ("allreduce", "rs_ag", True, 2), ("p2p", "rs_ag", True, 2)):
            # split=True: every exchange in flight under the next launches (RCCL's own stream / the p2p side stream, forked
            # and joined inside the captured graph); sparse=True: + the row-sparse backward (its bitmap all-gathers are
            # RCCL calls inside the graph too); sparse=2: + the light forward (frontier exchanges; a second captured graph
            # holds the full step that the last call replays)
            mp.spawn(_rccl_worker, args=(1, _free_port(), tmp, exchange, dc, split, bool(sparse), sparse == 2), nprocs=1, join=True)
            out[(exchange, dc, split, sparse)] = dict(np.load(os.path.join(tmp, f"rccl_{exchange}_{dc}_{int(split)}_{int(sparse)}.npz")))
    ref = out[("allreduce", "rs_ag", False, False)]
    for key, r in out.items():
        # two runs differ by the order of the BPR backward's atomic row adds (~1e-7 relative on a gradient); Adam turns a
        # gradient that is ALL rounding noise into a full step, so a handful of elements may differ by lr: count them
        for name in ("xu", "xi"):
            d = np.abs(r[name] - ref[name])
            assert (d > 2e-6).mean() <= 1e-4 and np.median(d) <= 1e-7, (key, name, float(d.max()))
        assert np.allclose(r["losses"], ref["losses"], rtol=1e-5), key
    assert "p2p" in str(out[("direct", "p2p", False, False)]["used"]) and "rs_ag" in str(out[("direct", "rs_ag", False, False)]["used"])


def _p2p_rows_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = "p2p"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist, ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    ok = True
    for n_rows, D, n_flag in ((1001, 64, 37), (4099, 128, 900), (333, 256, 333), (1001, 64, 0)):
        rows_pad = cdist.padded_rows(n_rows, None)
        for rep in range(3):
            g = torch.Generator().manual_seed(7 * n_rows + rep)               # the same on every rank: the GLOBAL frontier
            flagged = torch.randperm(n_rows, generator=g)[:n_flag]
            bits_np = np.zeros((n_rows + 31) // 32 + 1, dtype=np.uint32)
            np.bitwise_or.at(bits_np, flagged.numpy() >> 5, np.uint32(1) << (flagged.numpy() & 31).astype(np.uint32))
            bits = torch.from_numpy(bits_np.view(np.int32)).to(dev)
            g2 = torch.Generator().manual_seed(1000 * rep + rank)             # this rank's rows: a subset of the frontier
            mine = torch.zeros(rows_pad, D)
            own = flagged[torch.rand(n_flag, generator=g2) < 0.6]
            mine[own] = torch.randn(len(own), D, generator=g2)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            ordered = parts[0].clone()
            for q in parts[1:]:
                ordered += q
            buf = mine.to(dev)
            before = cdist.STATS.get("frontier_exchanges", 0)
            cdist._sum_exchange_async(buf, None, bits=bits, n_rows=n_rows).wait()
            torch.cuda.synchronize()
            ok = ok and torch.equal(buf.cpu(), ordered) and cdist.STATS.get("frontier_exchanges", 0) == before + 1
    np.savez(os.path.join(tmp, f"p2prows{rank}.npz"), ok=ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_frontier_exchange_moves_flagged_rows_only_and_sums_like_the_dense_one(world):
    """dist.P2PExchange with a row bitmap (the frontier buffers of the row-sparse sharded backward: the gradient seed, the
    first backward item partial): every rank's buffer is zero outside a GLOBAL set of flagged rows and holds values in its own
    part of it; the exchange copies / pulls / gathers flagged rows only and must leave every rank with the rank-ordered sum
    bit for bit -- block boundaries inside bitmap words, pad rows, an empty frontier."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_p2p_rows_worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
        assert all(bool(np.load(os.path.join(tmp, f"p2prows{k}.npz"))["ok"]) for k in range(world))


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_sums_like_an_all_reduce(world):
    """dist.P2PExchange (CHAOREC_DIST_EXCHANGE=p2p): IPC-mapped mailboxes + chaorec_exchange_pull_{sum,gather}_f32 between
    `world` processes sharing one GPU -- repeated exchanges on the same mailboxes, several sizes: the rank-ordered sum bit for
    bit, identical on every rank."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_p2p_worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
        r = [np.load(os.path.join(tmp, f"p2p{k}.npz")) for k in range(world)]
    assert all(bool(x["ok"]) for x in r)
    assert all(int(x["boxes"]) == 2 for x in r)            # two buffer sizes -> two pairs of mailboxes, set up once each
    for k in range(1, world):
        assert np.array_equal(r[0]["got"], r[k]["got"])


@pytest.mark.parametrize("L,world,exchange,split,sparse", [
    (1, 2, "allreduce", False, False), (3, 2, "allreduce", False, False), (2, 4, "allreduce", False, False),
    (3, 2, "p2p", False, False), (2, 4, "p2p", False, False), (3, 2, "allreduce", True, False), (2, 4, "p2p", True, False),
    (1, 2, "p2p", True, False), (3, 2, "allreduce", True, True), (4, 4, "p2p", True, True), (2, 2, "allreduce", True, True),
    (3, 2, "p2p", True, 2), (4, 4, "allreduce", True, 2), (2, 2, "p2p", True, 2)])
def test_fused_sharded_step_world2_on_the_kernels(oracle, L, world, exchange, split, sparse):
    """split=True: the launch sequence of large item tables (dist.FusedShardedLightGCNStep._launch_split) -- every joined
    launch as its two row blocks, every exchange travelling under the launches that follow it.  sparse=True: + the
    row-sparse backward (row lists at L >= 3, gated gathers, item bitmaps united over the ranks); sparse=2: + the light
    forward (the last two layers over the frontier's row lists, frontier exchanges, the last step a full one)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), tmp, L, exchange, split, bool(sparse), sparse == 2), nprocs=world, join=True)
        r = [np.load(os.path.join(tmp, f"rank{k}.npz")) for k in range(world)]
    # the whole-graph reference: the oracle's loss gradient of the mean over the ranks' batches + Adam, in fp64
    from chaorec_amd import dist as cdist
    edges, x0, _ = _problem()
    csr = oracle.lightgcn_csr(edges, U + I)
    x = x0.numpy().astype(np.float64)
    m, v = np.zeros_like(x), np.zeros_like(x)
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    bounds = cdist.partition_users_by_nnz(np.bincount(edges[:, 0], minlength=U), world)
    for t in range(T):
        g_tot = np.zeros_like(x)
        for k in range(world):
            mine = edges[(edges[:, 0] >= bounds[k]) & (edges[:, 0] < bounds[k + 1])]
            rng = np.random.default_rng(100 * t + k)
            sel = rng.choice(len(mine), B, replace=False)
            n_loc = bounds[k + 1] - bounds[k]
            bu = mine[sel, 0].astype(np.int64)
            bp = mine[sel, 1].astype(np.int64) - U
            bn = rng.integers(n_loc, n_loc + I, B) - n_loc
            out, g = oracle.lightgcn_loss(x.astype(np.float32), csr, L, U, bu, bp, bn, 1e-3)
            g_tot += g / world
            assert r[k]["losses"][t] == pytest.approx(out[0], rel=2e-5), (t, k)
        m = b1 * m + (1 - b1) * g_tot
        v = b2 * v + (1 - b2) * g_tot * g_tot
        x = x - lr * (m / (1 - b1 ** (t + 1))) / (np.sqrt(v) / np.sqrt(1 - b2 ** (t + 1)) + eps)
    xu = np.concatenate([x_["xu"] for x_ in r], 0)
    assert np.allclose(xu, x[:U], rtol=0, atol=3e-5)
    for k in range(world):
        assert np.allclose(r[k]["xi"], x[U:], rtol=0, atol=3e-5)
        assert np.array_equal(r[0]["xi"], r[k]["xi"])                  # identical item update on every rank


def test_split_launches_equal_the_joined_launches_bit_for_bit():
    """One process, no group: the split step's forward (row blocks of the joined graph as separate launches, the user
    rows' mean in the B_g launch's epilogue) writes the same bits as the joined step's; after three Adam steps the tables
    agree up to the BPR backward's atomic-add order."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    edges, x0, _ = _problem()
    out = {}
    for split in (False, True):
        shard = cdist.UserShard.from_local(edges, [0, U], I, 1, 0, dev)
        m = cdist.ShardedLightGCN(shard, None, D, 1e-3, 3, dev, seed=1).to(dev)
        with torch.no_grad():
            m.user_embedding.weight.copy_(x0[:U])
            m.item_embedding.weight.copy_(x0[U:])
        step = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True,
                                              capture=False, split=split)
        finals = []
        for t in range(T):
            rng = np.random.default_rng(100 * t)
            sel = rng.choice(len(shard.local_edges), B, replace=False)
            users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64)).to(dev)
            pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64)).to(dev)
            neg = torch.from_numpy(rng.integers(U, U + I, B)).to(dev)
            step(users, pos, neg)
            if t == 0:
                finals.append(step.final[:U + I].clone())
        out[split] = (finals[0].cpu(), m.user_embedding.weight.detach().cpu(), m.item_embedding.weight.detach().cpu())
    assert torch.equal(out[False][0], out[True][0])                       # the first forward: same weights in, same bits out
    for a, b in zip(out[False][1:], out[True][1:]):
        d = (a - b).abs()
        assert float((d > 2e-6).float().mean()) <= 1e-4 and float(d.median()) <= 1e-7


def test_sharded_rowsparse_backward_equals_the_dense_one():
    """One process, no group, captured: the split step with the row-sparse backward against the same step with dense
    launches -- the same tables after three Adam steps up to the BPR backward's atomic-add order (the propagates themselves
    give the same bits: tests/test_gpu_round4.py), the step's own frontier buffers back to all-zero."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    edges, x0, _ = _problem()
    for L in (2, 3, 4):
        out = {}
        for sparse in (False, True, 2):                   # (2: + the light forward)
            shard = cdist.UserShard.from_local(edges, [0, U], I, 1, 0, dev)
            m = cdist.ShardedLightGCN(shard, None, D, 1e-3, L, dev, seed=1).to(dev)
            with torch.no_grad():
                m.user_embedding.weight.copy_(x0[:U])
                m.item_embedding.weight.copy_(x0[U:])
            step = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True,
                                                  capture=True, split=True, sparse_bwd=bool(sparse), light_forward=sparse == 2)
            for t in range(T):
                rng = np.random.default_rng(100 * t)
                sel = rng.choice(len(shard.local_edges), B, replace=False)
                users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64)).to(dev)
                pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64)).to(dev)
                neg = torch.from_numpy(rng.integers(U, U + I, B)).to(dev)
                step(users, pos, neg, full_result=(t == T - 1))
            torch.cuda.synchronize()
            assert float(step.G.abs().max()) == 0.0
            if sparse:
                assert float(step.Z.abs().max()) == 0.0 and int(step._bits_all.abs().max()) == 0
            if sparse == 2:
                assert float(step.Z0.abs().max()) == 0.0
            out[sparse] = (m.user_embedding.weight.detach().cpu(), m.item_embedding.weight.detach().cpu(),
                           m.result_u.detach().cpu(), m.result_i.detach().cpu())
        for key in (True, 2):
            for a, b in zip(out[False], out[key]):
                d = (a - b).abs()
                assert float((d > 2e-6).float().mean()) <= 1e-4 and float(d.median()) <= 1e-7, (L, key)


def _calibrate_worker(rank, world, port, tmp, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("CHAOREC_DIST_EXCHANGE", None)
    os.environ["CHAOREC_FORCE_COLLECTIVES"] = "1"
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    cdist.AUTO_BIG_BYTES = 1 << 16
    table = cdist.calibrate_exchange(1001, 64, dev, captured=backend == "nccl", candidates=("allreduce", "rs_ag", "p2p"))
    buf = torch.zeros((cdist.padded_rows(1001), 64), device=dev)
    chosen = cdist.resolve_mode(buf)
    # the product path with `auto`: a buffer of the calibrated size goes through the chosen mode and sums correctly
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    src = torch.rand(buf.shape, generator=g, device=dev)
    want = src.clone()
    dist.all_reduce(want)
    got = src.clone()
    cdist._sum_exchange_async(got, None).wait()
    torch.cuda.synchronize()
    import json
    with open(os.path.join(tmp, f"cal{rank}.json"), "w") as f:
        json.dump({"table": table, "chosen": chosen, "sum_ok": bool(torch.allclose(got, want, rtol=0, atol=1e-5)),
                   "used": cdist.exchange_mode_used()}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_calibrate_exchange_on_the_gpu(world, backend):
    """dist.calibrate_exchange on the real kernels: two ranks sharing one GPU (gloo collectives + the IPC pull kernels),
    and a 1-rank RCCL group with the exchanges also replayed from a hipGraph (the p2p exchange on its side stream, its
    barriers RCCL launches inside the graph).  Every mode passes; `auto` then resolves large buffers to the chosen one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_calibrate_worker, args=(world, _free_port(), tmp, backend), nprocs=world, join=True)
        r = [json.load(open(os.path.join(tmp, f"cal{k}.json"))) for k in range(world)]
    for x in r:
        assert all(x["table"][m]["ok"] for m in ("allreduce", "rs_ag", "p2p")), x["table"]
        assert x["table"]["p2p"]["frontier_form_ok"] is True         # (the flagged-rows form had its first contact too)
        assert x["chosen"] == x["table"]["chosen"] == r[0]["chosen"] and x["sum_ok"], x


# ---------------------------------------------------------------------------------------------------- MMGCN (configs[3])
def _mmgcn_problem():
    from chaorec_amd.synthetic import synthetic_interactions
    Um, Im, Em = 2000, 700, 12000
    edges = synthetic_interactions(Um, Im, Em, seed=8)
    g = torch.Generator().manual_seed(1)
    return Um, Im, edges, torch.randn(Im, 128, generator=g), torch.randn(Im, 256, generator=g)


def _mmgcn_full(dev):
    from chaorec_amd import graph
    from chaorec_amd.Model import MMGCN
    Um, Im, edges, v_feat, t_feat = _mmgcn_problem()
    torch.manual_seed(77)
    return MMGCN(Um, Im, edges, graph.user_item_dict_from_edges(edges), v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)


def _mmgcn_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    Um, Im, edges, _, _ = _mmgcn_problem()
    full = _mmgcn_full(dev)
    shard = cdist.UserShard(edges, Um, Im, world, rank, dev, self_loops=True)
    m = cdist.ShardedMMGCN(full, shard, dev)
    rng = np.random.default_rng(200 + rank)
    sel = rng.choice(len(shard.local_edges), 256, replace=False)
    u = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64))
    pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64))
    neg = torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + Im, 256))
    ut, it = torch.stack((u, u), 1), torch.stack((pos, neg), 1)
    m.zero_grad()
    loss = m.loss(ut, it)
    loss.backward()
    m.sync_grads()
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, f"mm{rank}.npz"), u0=shard.u0, u1=shard.u1, loss=float(loss), ut=ut.numpy(), it=it.numpy(),
             res=m.result.detach().cpu().numpy(), **{"g_" + n: p.grad.cpu().numpy() for n, p in m.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mmgcn_world2_on_the_kernels():
    """BASELINE configs[3]'s sharding on the real kernels: two ranks (one GPU, gloo) against the single-process MMGCN on
    the whole graph -- representation, loss, the summed gradient of every Linear, identical on both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_mmgcn_worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(tmp, f"mm{k}.npz"))) for k in range(world)]
    dev = torch.device("cuda:0")
    Um = _mmgcn_problem()[0]
    full = _mmgcn_full(dev)
    uts, its = [], []
    for k in range(world):
        n_local = int(r[k]["u1"] - r[k]["u0"])
        uts.append(torch.from_numpy(r[k]["ut"] + int(r[k]["u0"])))
        its.append(torch.from_numpy(r[k]["it"] - n_local + Um))
    loss = full.loss(torch.cat(uts), torch.cat(its))
    loss.backward()
    assert sum(float(x["loss"]) for x in r) == pytest.approx(float(loss.detach()), rel=2e-5)
    ref = full.result.detach().cpu().numpy()
    scale = np.abs(ref).max()
    for k in range(world):
        u0, u1 = int(r[k]["u0"]), int(r[k]["u1"])
        assert np.abs(r[k]["res"][:u1 - u0] - ref[u0:u1]).max() <= 3e-4 * scale
        assert np.abs(r[k]["res"][u1 - u0:] - ref[Um:]).max() <= 3e-4 * scale
    for n, p in full.named_parameters():
        g = p.grad.cpu().numpy()
        assert np.abs(r[0]["g_" + n] - g).max() <= 2e-3 * (np.abs(g).max() + 1e-12), n
        assert np.array_equal(r[0]["g_" + n], r[1]["g_" + n]), n       # identical update on every rank


# ---------------------------------------------------------------------------------------------------- FREEDOM (configs[2])
def _freedom_full(dev):
    """The small reference golden's FREEDOM (dropout 0.2: the per-epoch pruning is on) with its stored weights."""
    from conftest import load_golden
    from chaorec_amd.Model import FREEDOM
    from chaorec_amd import graph
    g = load_golden("freedom_small_drop.npz")
    Uf, If = int(g["U"]), int(g["I"])
    torch.manual_seed(0)
    m = FREEDOM(Uf, If, g["edges"], graph.user_item_dict_from_edges(g["edges"]), torch.from_numpy(g["v_feat"]),
                torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["D"]), float(g["reg"]), float(g["dropout"]),
                int(g["L"]), int(g["mm_layers"]), int(g["knn"]), float(g["w"]), dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(g["x0"][:Uf]))
        m.item_embedding.weight.copy_(torch.from_numpy(g["x0"][Uf:]))
    return m.to(dev), Uf, If, g["edges"]


def _freedom_batch(n_edges_local, rank, B=64):
    return np.random.default_rng(300 + rank).choice(n_edges_local, B, replace=False), np.random.default_rng(400 + rank)


def _freedom_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    full, Uf, If, edges = _freedom_full(dev)
    bounds = cdist.partition_users_by_nnz(np.bincount(edges[:, 0], minlength=Uf), world)
    sh = cdist.ShardedFREEDOM(full, bounds, world, rank, dev)
    sh.pre_epoch_processing()
    loc = sh.local_edges                                   # [global user, item + U_global] of this rank's users
    sel, rng = _freedom_batch(len(loc), rank)
    users = torch.from_numpy(loc[sel, 0] - sh.u0).to(dev)
    pos = torch.from_numpy(loc[sel, 1] - Uf).to(dev)
    neg = torch.from_numpy(rng.integers(0, If, len(sel))).to(dev)
    sh.zero_grad()
    loss = sh.loss(users, pos, neg)
    loss.backward()
    sh.sync_grads()
    torch.cuda.synchronize()
    np.savez(os.path.join(tmp, f"fr{rank}.npz"), u0=sh.u0, u1=sh.u1, loss=float(loss), users=(users + sh.u0).cpu().numpy(),
             pos=pos.cpu().numpy(), neg=neg.cpu().numpy(), res=sh.result.detach().cpu().numpy(), kept=sh.shard.nnz,
             **{"g_" + n: p.grad.cpu().numpy() for n, p in sh.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_freedom_world2_on_the_kernels():
    """BASELINE configs[2]'s sharding on the real kernels: two ranks (one GPU, gloo) with the DISTRIBUTED per-epoch pruning
    (keys of the own edges, k-th smallest over all ranks by all-reduced histograms) against the single-process FREEDOM: the
    same number of kept edges, representation, loss, gradients of the id embeddings and of the modality transforms."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_freedom_worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
        r = [dict(np.load(os.path.join(tmp, f"fr{k}.npz"))) for k in range(world)]
    dev = torch.device("cuda:0")
    full, Uf, If, _ = _freedom_full(dev)
    full.pre_epoch_processing()
    assert sum(int(x["kept"]) for x in r) * 2 == full.masked_adj.nnz           # the same pruned edge count
    users = torch.from_numpy(np.concatenate([x["users"] for x in r]))
    pos = torch.from_numpy(np.concatenate([x["pos"] for x in r]) + Uf)
    neg = torch.from_numpy(np.concatenate([x["neg"] for x in r]) + Uf)
    loss = full.loss(users, pos, neg)
    loss.backward()
    assert sum(float(x["loss"]) for x in r) == pytest.approx(float(loss.detach()), rel=1e-5)
    ref = full.result.detach().cpu().numpy()
    for k in range(world):
        u0, u1 = int(r[k]["u0"]), int(r[k]["u1"])
        assert np.allclose(r[k]["res"][:u1 - u0], ref[u0:u1], rtol=1e-5, atol=1e-7)
        assert np.allclose(r[k]["res"][u1 - u0:], ref[Uf:], rtol=1e-5, atol=1e-7)
    named = dict(full.named_parameters())
    gu = np.concatenate([x["g_user_embedding.weight"] for x in r], 0)
    ref_gu = named["user_embedding.weight"].grad.cpu().numpy()
    assert np.abs(gu - ref_gu).max() <= 2e-4 * np.abs(ref_gu).max() + 1e-9
    for n in ("item_embedding.weight", "image_trs.weight", "text_trs.weight", "image_trs.bias", "text_trs.bias"):
        g = named[n].grad.cpu().numpy()
        assert np.abs(r[0]["g_" + n] - g).max() <= 2e-4 * np.abs(g).max() + 1e-9, n
        assert np.allclose(r[0]["g_" + n], r[1]["g_" + n], rtol=0, atol=1e-12 + 1e-6 * np.abs(g).max()), n
