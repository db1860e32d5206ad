"""N>1 path on CPU: world_size-2 gloo processes run the user-sharded LightGCN propagate / loss / backward
with an oracle-backed stand-in for the HIP kernels (the product has no CPU kernels), and must reproduce the
single-process oracle on the full graph.  Covers the partition, the per-layer all-reduce pattern and the
gradient all-reduce."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _oracle_spmm(csr, x, y=None, alpha=1.0, z=None, beta=0.0, acc=None, acc_init=None, acc_w=0.0):
    """ops.spmm_raw's keyword contract (include/chaorec_hip.h, chaorec_spmm_csr_f32) on the CPU oracle."""
    from oracle import oracle
    s = torch.from_numpy(oracle.spmm((csr.rowptr.numpy(), csr.col.numpy(), csr.val.numpy()), x.detach().numpy()))
    if acc is not None:
        a0 = acc_w * acc_init if acc_init is not None else acc
        acc.copy_(a0 + acc_w * s)
    out = alpha * s
    if z is not None:
        out = out + beta * z
    if y is not None:
        y.copy_(out)
        return y
    return out


class _OracleBPR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tu, ti, users, pos, neg, variant, reg):
        from oracle import oracle
        out, coef = oracle.bpr_fwd(tu.detach().numpy(), ti.detach().numpy(), users.numpy(), pos.numpy(), neg.numpy(),
                                   variant, reg)
        ctx.save_for_backward(tu, ti, users, pos, neg)
        ctx.coef, ctx.reg = coef, reg
        return torch.tensor(out, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        from oracle import oracle
        tu, ti, users, pos, neg = ctx.saved_tensors
        gu, gi = oracle.bpr_bwd(tu.detach().numpy(), ti.detach().numpy(), users.numpy(), pos.numpy(), neg.numpy(),
                                ctx.coef, ctx.reg, float(g[0]))
        return torch.from_numpy(gu).float(), torch.from_numpy(gi).float(), None, None, None, None, None


def _bpr(tu, ti, users, pos, neg, variant, reg):
    return _OracleBPR.apply(tu, ti, users, pos, neg, variant, reg)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from chaorec_amd import dist as cdist, graph
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, L, B = 600, 250, 3000, 16, 2, 64
    edges = synthetic_interactions(U, I, E, seed=5)
    shard = cdist.UserShard(edges, U, I, world, rank, torch.device("cpu"))
    uid = graph.user_item_dict_from_edges(shard.local_edges)
    for u in range(shard.num_user_local):
        uid.setdefault(u, [])
    m = cdist.ShardedLightGCN(shard, uid, D, 1e-3, L, torch.device("cpu"), seed=9, spmm_fn=_oracle_spmm, bpr_fn=_bpr)
    # per-rank batch drawn from the rank's own edges
    rng = np.random.default_rng(100 + rank)
    sel = rng.choice(len(shard.local_edges), B, replace=False)
    users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64))
    pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64))
    neg = torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + I, B))
    loss = m.loss(users, pos, neg)
    loss.backward()
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), u0=shard.u0, u1=shard.u1, bounds=np.array(shard.bounds),
             fu=m.result_u.detach().numpy(), fi=m.result_i.detach().numpy(), loss=float(loss),
             gu=m.user_embedding.weight.grad.numpy(), gi=m.item_embedding.weight.grad.numpy(),
             xu=m.user_embedding.weight.detach().numpy(), xi=m.item_embedding.weight.detach().numpy(),
             users=users.numpy() + shard.u0, pos=pos.numpy() - shard.num_user_local,
             neg=neg.numpy() - shard.num_user_local, nnz=shard.ui.nnz)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, args_after_port, nprocs):
    """mp.spawn(fn, (world, port, *args_after_port)) with ONE retry on another port when the rendezvous itself failed (the
    port found by _free_port() can be taken between its close() and the store's bind -- the one thing in this file that is
    not deterministic; one run of it failed once and passed the next ten times).  Anything else -- an assertion, a wrong
    number -- is raised as it is."""
    for attempt in (0, 1):
        try:
            return mp.spawn(fn, args=(world, _free_port()) + tuple(args_after_port), nprocs=nprocs, join=True)
        except Exception as exc:     # noqa: BLE001
            text = str(exc)
            rendezvous = any(k in text for k in ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset",
                                                 "connect() timed out", "client socket has timed out", "DistNetworkError"))
            if attempt or not rendezvous:
                raise


def test_sharded_lightgcn_matches_single_process_oracle(oracle):
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker, world, (tmp,), world)
        r = [np.load(os.path.join(tmp, f"rank{k}.npz")) for k in range(world)]
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D, L = 600, 250, 3000, 16, 2
    edges = synthetic_interactions(U, I, E, seed=5)
    # partition: contiguous, covering, balanced by nnz
    assert r[0]["u0"] == 0 and r[0]["u1"] == r[1]["u0"] and r[1]["u1"] == U
    assert abs(int(r[0]["nnz"]) - int(r[1]["nnz"])) < 0.1 * E
    # same replicated item parameters on both ranks, user parameters are slices of one global init
    assert np.array_equal(r[0]["xi"], r[1]["xi"])
    x0 = np.concatenate([r[0]["xu"], r[1]["xu"], r[0]["xi"]], 0)
    csr = oracle.lightgcn_csr(edges, U + I)
    final, _ = oracle.lightgcn_forward(x0, csr, L)
    fu = np.concatenate([r[0]["fu"], r[1]["fu"]], 0)
    assert np.allclose(fu, final[:U], rtol=1e-5, atol=1e-7)
    for k in range(world):
        assert np.allclose(r[k]["fi"], final[U:], rtol=1e-5, atol=1e-7)      # replicated after the all-reduce
    # global loss = mean over ranks of the per-rank batch losses; gradients = d(global loss)
    outs, g_tot = [], np.zeros((U + I, D))
    for k in range(world):
        out, g = oracle.lightgcn_loss(x0, csr, L, U, r[k]["users"], r[k]["pos"], r[k]["neg"], 1e-3)
        outs.append(out[0])
        g_tot += g / world
        assert r[k]["loss"] == pytest.approx(out[0] / world, rel=1e-5)
    gu = np.concatenate([r[0]["gu"], r[1]["gu"]], 0)
    assert np.allclose(gu, g_tot[:U], rtol=2e-4, atol=1e-9)
    assert np.allclose(r[0]["gi"], g_tot[U:], rtol=2e-4, atol=1e-9)
    assert np.array_equal(r[0]["gi"], r[1]["gi"])                              # identical item update on every rank


def _worker4(rank, world, port, tmp, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = mode
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from chaorec_amd import dist as cdist
    U, I, D, L, B = 900, 301, 16, 3, 64          # (301 items: not a multiple of the world size -> padded exchange rows)
    edges = _heavy_tailed_graph(U, I)
    # per-rank construction: this rank sees ONLY its own users' edges
    deg = np.bincount(edges[:, 0], minlength=U)
    bounds = cdist.partition_users_by_nnz(deg, world)
    mine = edges[(edges[:, 0] >= bounds[rank]) & (edges[:, 0] < bounds[rank + 1])]
    shard = cdist.UserShard.from_local(mine, bounds, I, world, rank, torch.device("cpu"))
    whole = cdist.UserShard(edges, U, I, world, rank, torch.device("cpu"))            # the all-edges constructor
    assert shard.bounds == whole.bounds and torch.equal(shard.ui.val, whole.ui.val) and torch.equal(shard.iu.col, whole.iu.col)
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, L, torch.device("cpu"), seed=9, spmm_fn=_oracle_spmm, bpr_fn=_bpr)
    rng = np.random.default_rng(100 + rank)
    sel = rng.choice(len(shard.local_edges), B, replace=False)
    users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64))
    pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64))
    neg = torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + I, B))
    loss = m.loss(users, pos, neg)
    loss.backward()
    assert torch.equal(m.result[:shard.num_user_local], m.result_u) and torch.equal(m.result[shard.num_user_local:], m.result_i)
    # gather_ranklists: every rank contributes [U_g, K] rows, rank order = user order
    K = 7
    mine_rank = (torch.arange(shard.num_user_local).view(-1, 1) + shard.u0) * 100 + torch.arange(K).view(1, -1)
    gathered = cdist.gather_ranklists(mine_rank, shard)
    want = (torch.arange(U).view(-1, 1)) * 100 + torch.arange(K).view(1, -1)
    assert torch.equal(gathered, want)
    # GradBucket: gradients live in one flat buffer, one all-reduce sums them on every rank
    lin = torch.nn.Linear(5, 3)
    bucket = cdist.GradBucket(list(lin.parameters()))
    bucket.zero()
    lin(torch.full((2, 5), float(rank + 1))).sum().backward()
    assert bucket.attached()
    local = bucket.flat.clone()
    bucket.all_reduce()
    tot = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(tot, local)
    assert torch.allclose(bucket.flat, sum(tot)) and lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), u0=shard.u0, u1=shard.u1, bounds=np.array(shard.bounds),
             fu=m.result_u.detach().numpy(), fi=m.result_i.detach().numpy(), loss=float(loss),
             gu=m.user_embedding.weight.grad.numpy(), gi=m.item_embedding.weight.grad.numpy(),
             xu=m.user_embedding.weight.detach().numpy(), xi=m.item_embedding.weight.detach().numpy(),
             users=users.numpy() + shard.u0, pos=pos.numpy() - shard.num_user_local,
             neg=neg.numpy() - shard.num_user_local, nnz=shard.nnz)
    dist.barrier()
    dist.destroy_process_group()


def _heavy_tailed_graph(U, I, seed=3):
    """Users with Zipf-like degrees (a few users hold a large share of the edges: the case shard-by-count gets wrong)."""
    rng = np.random.default_rng(seed)
    deg = np.minimum(3 + (rng.pareto(1.1, U) * 2).astype(np.int64), I // 2)
    deg[:5] = I // 2                                   # five users at the front as heavy as 150 median users
    rows = []
    for u in range(U):
        for i in rng.choice(I, int(deg[u]), replace=False):
            rows.append((u, int(i) + U))
    return np.array(rows, dtype=np.int32)


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag", "direct"])
def test_world4_exchange_modes_per_rank_shards(oracle, mode):
    """SURVEY 8(e) at world size 4: shards built per rank from local edges only, balanced by nnz on a heavy-tailed
    graph; the per-layer item exchange as all-reduce, reduce-scatter + all-gather, and direct all-to-all + all-gather;
    rank-list gather; flat gradient bucket.  Every mode reproduces the single-process oracle."""
    world = 4
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker4, world, (tmp, mode), world)
        r = [np.load(os.path.join(tmp, f"rank{k}.npz")) for k in range(world)]
    U, I, D, L = 900, 301, 16, 3
    edges = _heavy_tailed_graph(U, I)
    nnz = np.array([int(x["nnz"]) for x in r])
    assert nnz.sum() == len(edges) and nnz.max() <= 1.25 * nnz.mean(), nnz            # balanced by edges ...
    sizes = np.array([int(x["u1"] - x["u0"]) for x in r])
    assert sizes.min() < 0.5 * sizes.max(), sizes                                     # ... not by user count
    x0 = np.concatenate([x["xu"] for x in r] + [r[0]["xi"]], 0)
    csr = oracle.lightgcn_csr(edges, U + I)
    final, _ = oracle.lightgcn_forward(x0, csr, L)
    assert np.allclose(np.concatenate([x["fu"] for x in r], 0), final[:U], rtol=1e-5, atol=1e-7)
    g_tot = np.zeros((U + I, D))
    for k in range(world):
        assert np.allclose(r[k]["fi"], final[U:], rtol=1e-5, atol=1e-7)
        out, g = oracle.lightgcn_loss(x0, csr, L, U, r[k]["users"], r[k]["pos"], r[k]["neg"], 1e-3)
        g_tot += g / world
        assert r[k]["loss"] == pytest.approx(out[0] / world, rel=1e-5)
    assert np.allclose(np.concatenate([x["gu"] for x in r], 0), g_tot[:U], rtol=2e-4, atol=1e-9)
    for k in range(world):
        assert np.allclose(r[k]["gi"], g_tot[U:], rtol=2e-4, atol=1e-9)
        assert np.array_equal(r[k]["gi"], r[0]["gi"])                                 # identical item update on every rank


class _OracleStepKernels:
    """dist._HipStepKernels' contract on the CPU (oracle SpMM / BPR, torch Adam arithmetic): the stand-in the fused
    sharded step runs on in the gloo tests."""
    spmm = staticmethod(_oracle_spmm)

    @staticmethod
    def mean_terms_limit(D):
        return 3

    @staticmethod
    def spmm_mean(csr, x, terms, w, mean_out, y=None):
        s = _oracle_spmm(csr, x)
        if y is not None:
            y.copy_(s)
        a = w * terms[0]
        for t in terms[1:]:
            a = a + w * t
        mean_out.copy_(a + w * s)

    @staticmethod
    def rows_mean(terms, w, out):
        a = w * terms[0]
        for t in terms[1:]:
            a = a + w * t
        out.copy_(a)

    @staticmethod
    def _adam(p, g, m, v, bc1, bc2s, lr, betas, eps, wd):
        if wd:
            g = g + wd * p
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        p.addcdiv_(m, v.sqrt() / bc2s + eps, value=-lr / bc1)

    @classmethod
    def spmm_adam(cls, csr, x, p, m, v, bc, lr, betas, eps, wd, alpha=1.0, z=None, beta=0.0, clear_z=False, clear_bits=()):
        g = _oracle_spmm(csr, x, alpha=alpha, z=z, beta=beta)
        cls._adam(p, g, m, v, float(bc[0]), float(bc[1]), lr, betas, eps, wd)
        if clear_z:
            z.zero_()
        for b in clear_bits:
            b.zero_()

    @classmethod
    def adam_step(cls, p, g, m, v, step, lr, betas, eps, wd, step_dev=None):
        t = int(step_dev) if step_dev is not None else step
        cls._adam(p, g, m, v, 1 - betas[0] ** t, (1 - betas[1] ** t) ** 0.5, lr, betas, eps, wd)

    # ---- the row-sparse backward's launches: the same results, computed the dense way, with the launches' contract about
    # WHAT IS LEFT ALONE made hostile -- a row the launch does not write turns into NaN when it holds stale values (a later
    # reader of it would poison the step), and a source row that is not flagged is not read at all
    sparse_widths = (4, 4096)

    @staticmethod
    def _rows(bits, n):
        w = bits.numpy().view(np.uint32)
        r = np.arange(n)
        return ((w[r >> 5] >> (r & 31).astype(np.uint32)) & 1).astype(bool)

    @staticmethod
    def _set(bits, rows):
        w = bits.numpy().view(np.uint32)                 # (shares the tensor's memory)
        np.bitwise_or.at(w, rows >> 5, (np.uint32(1) << (rows & 31).astype(np.uint32)))

    @classmethod
    def expand_row_bits(cls, csr, bits_in, bits_out, row_list=None, list_n=None, bits_self=None):
        assert bits_self is not None or csr.symmetric
        rp, col = csr.rowptr.numpy(), csr.col.numpy()
        hit = cls._rows(bits_self if bits_self is not None else bits_in, csr.n_cols).copy()
        for r in np.nonzero(cls._rows(bits_in, csr.n_rows))[0]:
            hit[col[rp[r]:rp[r + 1]]] = True
        new = np.nonzero(hit & ~cls._rows(bits_out, csr.n_cols))[0]
        cls._set(bits_out, new)
        if row_list is not None:
            n0 = int(list_n[0])
            row_list[n0:n0 + len(new)] = torch.from_numpy(new[::-1].astype(np.int32).copy())       # (any order)
            list_n[0] = n0 + len(new)

    @classmethod
    def _gated(cls, csr, x, alpha, z, beta, src_bits, z_bits):
        xm = torch.where(torch.from_numpy(cls._rows(src_bits, csr.n_cols))[:, None], x, torch.zeros(()))
        zm = torch.where(torch.from_numpy(cls._rows(z_bits, csr.n_rows))[:, None], z, torch.zeros(()))
        assert not torch.isnan(xm).any() and not torch.isnan(zm).any()
        return _oracle_spmm(csr, xm, alpha=alpha, z=zm, beta=beta)

    @classmethod
    def spmm_rowlist(cls, csr, x, y, row_list, list_n, alpha=1.0, z=None, beta=0.0, src_bits=None, z_bits=None, mean_out=None,
                     mean_terms=(), mean_w=0.0, long_rows=None):
        rows = row_list[:int(list_n[0])].long()
        assert len(torch.unique(rows)) == len(rows)
        keep = torch.zeros(csr.n_rows, dtype=torch.bool)
        keep[rows] = True
        if src_bits is None:
            # a forward list launch: every entry of a listed row is gathered -- a NaN (stale) source row it reads shows up
            assert z is None
            xs = x.clone()
            full = torch.zeros(csr.n_rows, x.shape[1])
            rp, col, val = csr.rowptr.numpy(), csr.col.numpy(), csr.val
            sub = _oracle_spmm(csr, torch.nan_to_num(xs, nan=0.0), alpha=alpha)
            for r in rows.tolist():
                assert not torch.isnan(xs[col[rp[r]:rp[r + 1]]]).any(), "a listed row gathers a row nobody computed"
            full[rows] = sub[rows]
        else:
            full = cls._gated(csr, x, alpha, z, beta, src_bits, z_bits)
            assert float(full[~keep].abs().max()) == 0.0 if (~keep).any() else True  # (the list covers every non-zero row)
        if y is not None:
            if float(y[~keep].abs().nan_to_num(1.0).max() if (~keep).any() else 0.0) != 0.0:
                y[~keep] = float("nan")                  # stale rows: poisoned; an all-zero buffer stays all-zero
            y[rows] = full[rows]
        if mean_out is not None:
            a = mean_w * mean_terms[0][rows]
            for t in mean_terms[1:]:
                a = a + mean_w * t[rows]
            assert not torch.isnan(a).any()
            mean_out[~keep] = float("nan")               # (only the listed rows of the mean exist)
            mean_out[rows] = a + mean_w * full[rows]

    @classmethod
    def batch_rows(cls, ids, row_bits, bits_item_offset, row_list=None, list_n=None, edges=None, hist=None, num_user=0,
                   num_item=0, seed=0, step=0, step_dev=None, perm=None, perm_pos=None, pos_offset=0):
        assert edges is None and row_list is None
        u, p, n = (t.numpy() for t in ids)
        cls._set(row_bits, u)
        cls._set(row_bits, bits_item_offset + p)
        cls._set(row_bits, bits_item_offset + n)

    @classmethod
    def rows_list_from_bits(cls, bits, n_rows, row_list, list_n):
        rows = np.nonzero(cls._rows(bits, n_rows))[0]
        n0 = int(list_n[0])
        row_list[n0:n0 + len(rows)] = torch.from_numpy(rows[::-1].astype(np.int32).copy())
        list_n[0] = n0 + len(rows)

    @classmethod
    def rows_mean_by_bits(cls, terms, w, out, bits):
        m = torch.from_numpy(cls._rows(bits, out.shape[0]))
        a = w * terms[0][m]
        for t in terms[1:]:
            a = a + w * t[m]
        assert not torch.isnan(a).any()
        out[~m] = float("nan")
        out[m] = a

    @staticmethod
    def long_row_buffers(csr, threshold=None):
        return None

    @classmethod
    def frontier_pack(cls, src, bits, prefix, compact, overflow=None):
        rows = np.nonzero(cls._rows(bits, src.shape[0]))[0]              # (bitmap order: the same on every rank)
        if overflow is not None:                                         # (the device kernel drops the rows past the capacity)
            overflow[0] = max(int(overflow[0]), len(rows) - compact.shape[0])
        else:
            assert len(rows) <= compact.shape[0]
        keep = rows[:compact.shape[0]]
        compact.zero_()
        compact[:len(keep)] = src[keep]
        prefix[-1] = len(rows)

    @classmethod
    def frontier_unpack(cls, dst, bits, prefix, compact):
        rows = np.nonzero(cls._rows(bits, dst.shape[0]))[0]
        assert len(rows) == int(prefix[-1])
        keep = rows[:compact.shape[0]]                                    # (like the kernel: rows past the capacity are left alone)
        dst[keep] = compact[:len(keep)]

    @classmethod
    def spmm_rowsparse(cls, csr, x, y, alpha=1.0, z=None, beta=0.0, src_bits=None, z_bits=None):
        y.copy_(cls._gated(csr, x, alpha, z, beta, src_bits, z_bits))

    @classmethod
    def zero_rows_by_bits(cls, y, bits):
        y[torch.from_numpy(cls._rows(bits, y.shape[0]))] = 0.0

    @classmethod
    def rows_copy_by_bits(cls, dst, src, bits):
        m = torch.from_numpy(cls._rows(bits, dst.shape[0]))
        dst[m] = src[m]

    @staticmethod
    def or_words(dst, src):
        acc = src[0].clone()
        for k in range(1, src.shape[0]):
            acc |= src[k]
        dst.copy_(acc)

    @classmethod
    def bpr_fwd_bwd(cls, tab, item_offset, grad, B, variant, reg, coef, ws, ids, edges=None, hist=None, num_user=0, num_item=0,
                    seed=0, step=0, step_dev=None, adam_step=None, betas=(0.9, 0.999), adam_bc=None, row_bits=None,
                    bits_item_offset=None):
        from oracle import oracle
        assert edges is None
        U, I = item_offset, num_item
        tu, ti = tab[:U].numpy(), tab[U:U + I].numpy()
        u, p, n = (t.numpy() for t in ids)
        if row_bits is not None:
            cls._set(row_bits, u)
            cls._set(row_bits, bits_item_offset + p)
            cls._set(row_bits, bits_item_offset + n)
        out, cf = oracle.bpr_fwd(tu, ti, u, p, n, variant, reg)
        gu, gi = oracle.bpr_bwd(tu, ti, u, p, n, cf, reg, 1.0)
        grad[:U] += torch.from_numpy(gu).float()
        grad[U:U + I] += torch.from_numpy(gi).float()
        ws[:3] = torch.tensor(out, dtype=torch.float32)
        adam_step += 1
        t = int(adam_step)
        adam_bc[0], adam_bc[1] = 1 - betas[0] ** t, (1 - betas[1] ** t) ** 0.5

    @staticmethod
    def bpr_finalize(ws, B, D, reg, out, out_total=None, loss_accum=None, advance=None):
        out.copy_(ws[:3])
        if out_total is not None:
            out_total.copy_(ws[0])
        if loss_accum is not None:
            loss_accum += ws[0]


def _worker_fused(rank, world, port, tmp, mode, L, split=False, sparse=False, light=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = mode
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    U, I, D, B, T = 500, 203, 16, 48, 3           # (203 items: padded exchange rows at every world size used here)
    edges = _heavy_tailed_graph(U, I)
    deg = np.bincount(edges[:, 0], minlength=U)
    bounds = cdist.partition_users_by_nnz(deg, world)
    mine = edges[(edges[:, 0] >= bounds[rank]) & (edges[:, 0] < bounds[rank + 1])]
    shard = cdist.UserShard.from_local(mine, bounds, I, world, rank, torch.device("cpu"))
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, L, torch.device("cpu"), seed=9)
    x0u, x0i = m.user_embedding.weight.detach().clone().numpy(), m.item_embedding.weight.detach().clone().numpy()
    opt = FusedAdam(m.parameters(), lr=1e-2)
    step = cdist.FusedShardedLightGCNStep(m, opt, batch_size=B, given_batch=True, capture=False, kernels=_OracleStepKernels,
                                          split=split, sparse_bwd=sparse, light_forward=light)
    assert step.split == split and step.sparse_bwd == sparse and step.light == light
    assert step.N_pad % world == (shard.num_user_local % world)          # item rows padded to a multiple of the world size
    rng = np.random.default_rng(100 + rank)
    batches, losses = [], []
    for t in range(T):
        sel = rng.choice(len(shard.local_edges), B, replace=False)
        users = torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64))
        pos = torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64))
        neg = torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + I, B))
        losses.append(float(step(users, pos, neg, full_result=(t == T - 1))))
        assert (m.result_u is None) == (light and t < T - 1)
        batches.append(np.stack([users.numpy() + shard.u0, pos.numpy() - shard.num_user_local,
                                 neg.numpy() - shard.num_user_local]))
    assert float(step.G.abs().max()) == 0.0                               # the gradient buffer is all-zero between steps
    if light:
        assert float(step.Z0.abs().max()) == 0.0
    if sparse:
        assert float(step.Z.abs().max()) == 0.0 and int(step._bits_all.abs().max()) == 0       # ... and so are these
        assert float(step.S[step.U:].abs().max()) == 0.0
        # N1's item frontier went through the COMPACT exchange too (capacity fixed at the first step: VERDICT r4 #3), within
        # its capacity
        if L >= 3 or light:             # (at L = 2 without the light forward no launch has N1's item frontier as its output)
            assert step._cap1 is not None and 0 < step._cap1 <= I and (step.Z.data_ptr(), step._cap1) in step._compact
        assert "compact-allreduce" in cdist.MODES_USED and cdist.STATS.get("frontier_exchanges", 0) >= T
        step.check_frontier()
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), x0u=x0u, x0i=x0i, xu=m.user_embedding.weight.detach().numpy(),
             xi=m.item_embedding.weight.detach().numpy(), fu=m.result_u.numpy(), fi=m.result_i.numpy(),
             batches=np.stack(batches), losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


def _worker_frontier_cap(rank, world, port, tmp):
    """A frontier capacity that is too small (here: forced to 8 rows) must surface as an error, not as a wrong step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CHAOREC_DIST_EXCHANGE"] = "allreduce"
    os.environ["CHAOREC_DIST_FRONTIER_CAP"] = "8"
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam
    U, I, D, B = 500, 203, 16, 48
    edges = _heavy_tailed_graph(U, I)
    bounds = cdist.partition_users_by_nnz(np.bincount(edges[:, 0], minlength=U), world)
    mine = edges[(edges[:, 0] >= bounds[rank]) & (edges[:, 0] < bounds[rank + 1])]
    shard = cdist.UserShard.from_local(mine, bounds, I, world, rank, torch.device("cpu"))
    m = cdist.ShardedLightGCN(shard, None, D, 1e-3, 3, torch.device("cpu"), seed=9)
    step = cdist.FusedShardedLightGCNStep(m, FusedAdam(m.parameters(), lr=1e-2), batch_size=B, given_batch=True, capture=False,
                                          kernels=_OracleStepKernels, split=True, sparse_bwd=True, light_forward=True)
    rng = np.random.default_rng(100 + rank)
    sel = rng.choice(len(shard.local_edges), B, replace=False)
    step(torch.from_numpy(shard.local_edges[sel, 0].astype(np.int64)), torch.from_numpy(shard.local_edges[sel, 1].astype(np.int64)),
         torch.from_numpy(rng.integers(shard.num_user_local, shard.num_user_local + I, B)), full_result=True)
    raised = ""
    try:
        step.check_frontier()
    except RuntimeError as exc:
        raised = str(exc)
    open(os.path.join(tmp, f"rank{rank}.txt"), "w").write(raised)
    dist.barrier()
    dist.destroy_process_group()


def test_a_frontier_that_outgrows_its_compact_exchange_is_an_error():
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_frontier_cap, 2, (tmp,), 2)
        for k in range(2):
            msg = open(os.path.join(tmp, f"rank{k}.txt")).read()
            assert "overflowed its capacity" in msg and "CHAOREC_DIST_FRONTIER_CAP" in msg, msg


@pytest.mark.parametrize("world,mode,L,split,sparse,light", [
    (2, "allreduce", 3, False, False, False), (2, "direct", 1, False, False, False), (4, "rs_ag", 2, False, False, False),
    (2, "auto", 3, True, False, False), (4, "allreduce", 2, True, False, False), (2, "rs_ag", 1, True, False, False),
    (2, "allreduce", 3, True, True, False), (4, "rs_ag", 4, True, True, False), (2, "auto", 2, True, True, False),
    (2, "allreduce", 3, True, True, True), (4, "rs_ag", 4, True, True, True), (2, "auto", 2, True, True, True)])
def test_fused_sharded_step_trains_like_the_single_process_oracle(oracle, world, mode, L, split, sparse, light):
    """dist.FusedShardedLightGCNStep (joined-graph propagates, in-place item exchanges, Adam in the last propagate /
    one fused launch for the replicated item rows) over T optimizer steps against the oracle on the WHOLE graph: the global
    loss is the mean of the ranks' batch losses, torch.optim.Adam's arithmetic on its gradient.  split=True: the launch
    sequence of large item tables (every joined launch as its two row blocks, every exchange in flight under the next
    launches, dist.FusedShardedLightGCNStep._launch_split).  sparse=True: the first two backward propagates over the
    batch's frontier only (row lists / gated gathers; item bitmaps united over the ranks) -- the stand-in kernels turn every
    stale row such a launch leaves behind into NaN, so a reader of one cannot go unnoticed.  light=True: + the light forward
    (batch rows first; the last two layers over the frontier's row lists, the item rows' partials through frontier buffers;
    model.result withheld until the last step, which is a full one)."""
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_fused, world, (tmp, mode, L, split, sparse, light), world)
        r = [np.load(os.path.join(tmp, f"rank{k}.npz")) for k in range(world)]
    U, I, D, T = 500, 203, 16, 3
    edges = _heavy_tailed_graph(U, I)
    csr = oracle.lightgcn_csr(edges, U + I)
    x = np.concatenate([x_["x0u"] for x_ in r] + [r[0]["x0i"]], 0).astype(np.float64)
    m, v = np.zeros_like(x), np.zeros_like(x)
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    for t in range(T):
        g_tot = np.zeros_like(x)
        for k in range(world):
            bu, bp, bn = r[k]["batches"][t]
            out, g = oracle.lightgcn_loss(x.astype(np.float32), csr, L, U, bu, bp, bn, 1e-3)
            g_tot += g / world
            assert r[k]["losses"][t] == pytest.approx(out[0], rel=2e-5), (t, k)
        m = b1 * m + (1 - b1) * g_tot
        v = b2 * v + (1 - b2) * g_tot * g_tot
        x = x - lr * (m / (1 - b1 ** (t + 1))) / (np.sqrt(v) / np.sqrt(1 - b2 ** (t + 1)) + eps)
    xu = np.concatenate([x_["xu"] for x_ in r], 0)
    assert np.allclose(xu, x[:U], rtol=0, atol=2e-5)
    for k in range(world):
        assert np.allclose(r[k]["xi"], x[U:], rtol=0, atol=2e-5)
        assert np.array_equal(r[k]["xi"], r[0]["xi"])                      # identical item update on every rank


def test_partition_users_by_nnz():
    from chaorec_amd.dist import partition_users_by_nnz
    deg = np.array([1000] + [1] * 999)
    b = partition_users_by_nnz(deg, 4)
    assert b[0] == 0 and b[-1] == 1000 and all(b[i] <= b[i + 1] for i in range(4))
    deg = np.full(800, 5)
    assert partition_users_by_nnz(deg, 8) == [100 * k for k in range(9)]


# ---------------------------------------------------------------------------------------------------- MMGCN
def _cpu_standins():
    """CPU stand-ins with the ops.* contracts (the product has no CPU kernels): plain torch / the oracle."""
    import torch.nn.functional as F
    from chaorec_amd import ops

    def linear(x, w, b=None, act=0):
        y = F.linear(x, w, b)
        return F.leaky_relu(y) if act == 1 else y

    class _Spmm(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, csr):
            ctx.csr = csr
            return _oracle_spmm(csr, x)

        @staticmethod
        def backward(ctx, g):
            return _oracle_spmm(ctx.csr.t(), g.contiguous()), None

    def bpr_loss(tab_u, tab_i, users, pos, neg, variant, reg_weight=0.0, item_offset=0):
        assert tab_i is None and variant == ops.VARIANT_LOG_SIGMOID and reg_weight == 0.0
        s = (tab_u[users] * tab_u[pos]).sum(1) - (tab_u[users] * tab_u[neg]).sum(1)
        return (-torch.mean(torch.log(torch.sigmoid(s))),)

    ops.linear = linear
    ops.spmm = lambda csr, x: _Spmm.apply(x, csr)
    ops.spmm_raw = _oracle_spmm
    ops.bpr_loss = bpr_loss
    ops.mean_all = lambda x: x.mean()


def _mmgcn_problem():
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E = 300, 120, 1500
    edges = synthetic_interactions(U, I, E, seed=8)
    g = torch.Generator().manual_seed(1)
    return U, I, edges, torch.randn(I, 32, generator=g), torch.randn(I, 24, generator=g)


def _mmgcn_full(U, I, edges, v_feat, t_feat):
    from chaorec_amd import graph
    from chaorec_amd.Model import MMGCN
    torch.manual_seed(77)
    return MMGCN(U, I, edges, graph.user_item_dict_from_edges(edges), v_feat, t_feat, 16, 1e-4, "add", "False", True,
                 torch.device("cpu"))


def _mmgcn_batch(shard_edges, n_local, I, rank, B=48):
    rng = np.random.default_rng(200 + rank)
    sel = rng.choice(len(shard_edges), B, replace=False)
    u = torch.from_numpy(shard_edges[sel, 0].astype(np.int64))
    pos = torch.from_numpy(shard_edges[sel, 1].astype(np.int64))
    neg = torch.from_numpy(rng.integers(n_local, n_local + I, B))
    return torch.stack((u, u), 1), torch.stack((pos, neg), 1)


def _mmgcn_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    _cpu_standins()
    from chaorec_amd import dist as cdist
    U, I, edges, v_feat, t_feat = _mmgcn_problem()
    full = _mmgcn_full(U, I, edges, v_feat, t_feat)
    shard = cdist.UserShard(edges, U, I, world, rank, torch.device("cpu"), self_loops=True)
    m = cdist.ShardedMMGCN(full, shard, torch.device("cpu"))
    ut, it = _mmgcn_batch(shard.local_edges, shard.num_user_local, I, rank)
    loss = m.loss(ut, it)
    loss.backward()
    first = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.sync_grads()
    # the same step again through the persistent gradient bucket (zero_grad keeps the views, sync is ONE all-reduce)
    m.zero_grad()
    assert m._bucket is not None and m._bucket.attached()
    m.loss(ut, it).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, first[n], rtol=1e-6, atol=1e-9), n
    m.sync_grads()
    assert m._bucket.attached()
    np.savez(os.path.join(tmp, f"mm{rank}.npz"), u0=shard.u0, u1=shard.u1, loss=float(loss), ut=ut.numpy(), it=it.numpy(),
             res=m.result.detach().numpy(), **{"g_" + n: p.grad.numpy() for n, p in m.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mmgcn_matches_single_process():
    """BASELINE configs[3]: MMGCN sharded by user rows (2 gloo ranks, CPU stand-ins for the kernels) reproduces the
    single-process model: representations, loss, and the summed gradients of every Linear."""
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_mmgcn_worker, world, (tmp,), world)
        r = [dict(np.load(os.path.join(tmp, f"mm{k}.npz"))) for k in range(world)]
    from chaorec_amd import ops
    saved = (ops.linear, ops.spmm, ops.spmm_raw, ops.bpr_loss, ops.mean_all)
    try:
        _cpu_standins()
        U, I, edges, v_feat, t_feat = _mmgcn_problem()
        full = _mmgcn_full(U, I, edges, v_feat, t_feat)
        # the ranks' batches in global ids: users + u0, items - U_g + U
        uts, its = [], []
        for k in range(world):
            n_local = int(r[k]["u1"] - r[k]["u0"])
            uts.append(torch.from_numpy(r[k]["ut"] + int(r[k]["u0"])))
            its.append(torch.from_numpy(r[k]["it"] - n_local + U))
        loss = full.loss(torch.cat(uts), torch.cat(its))
        loss.backward()
        assert sum(float(x["loss"]) for x in r) == pytest.approx(float(loss.detach()), rel=2e-5)
        ref = full.result.detach().numpy()
        for k in range(world):
            u0, u1 = int(r[k]["u0"]), int(r[k]["u1"])
            assert np.allclose(r[k]["res"][:u1 - u0], ref[u0:u1], rtol=2e-4, atol=1e-6)
            assert np.allclose(r[k]["res"][u1 - u0:], ref[U:], rtol=2e-4, atol=1e-6)
        for n, p in full.named_parameters():
            g = p.grad.numpy()
            assert np.allclose(r[0]["g_" + n], g, rtol=2e-3, atol=1e-7 + 1e-4 * np.abs(g).max()), n
            assert np.array_equal(r[0]["g_" + n], r[1]["g_" + n]), n       # identical update on every rank
    finally:
        ops.linear, ops.spmm, ops.spmm_raw, ops.bpr_loss, ops.mean_all = saved


# ---------------------------------------------------------------------------------------------------- FREEDOM
def _freedom_full(dropout=0.2):
    """Everything dist.ShardedFREEDOM reads from a single-process FREEDOM, without its constructor (the kNN build is a
    GPU kernel): same seed -> same object on every rank and in the checking process."""
    import types
    from chaorec_amd import graph
    from chaorec_amd.Model.FREEDOM import FREEDOM
    from chaorec_amd.synthetic import synthetic_interactions
    U, I, E, D = 260, 90, 1300, 16
    edges = synthetic_interactions(U, I, E, seed=12)
    torch.manual_seed(31)
    ns = types.SimpleNamespace(num_user=U, num_item=I, n_layers=2, mm_layers=1, reg_weight=1e-3, dropout=dropout,
                               _prune_seed=77)
    ns.user_embedding, ns.item_embedding = torch.nn.Embedding(U, D), torch.nn.Embedding(I, D)
    torch.nn.init.xavier_uniform_(ns.user_embedding.weight)
    torch.nn.init.xavier_uniform_(ns.item_embedding.weight)
    ns.text_embedding = torch.nn.Embedding.from_pretrained(torch.randn(I, 12), freeze=False)
    ns.image_embedding = torch.nn.Embedding.from_pretrained(torch.randn(I, 20), freeze=False)
    ns.text_trs, ns.image_trs = torch.nn.Linear(12, D), torch.nn.Linear(20, D)
    g = torch.Generator().manual_seed(5)
    rows = torch.arange(I).repeat_interleave(4)
    cols = torch.randint(0, I, (4 * I,), generator=g)
    ns.mm_adj = graph.coo_to_csr_coalesced(rows, cols, torch.full((4 * I,), 0.25), I, I)
    e = torch.from_numpy(np.asarray(edges, dtype=np.int64))
    ns.edge_indices = torch.stack([e[:, 0], e[:, 1] - U])
    ns.edge_values = FREEDOM._normalize_adj_m(None, ns.edge_indices, torch.Size((U, I)))
    return ns, edges


def _freedom_standins(oracle_mod):
    import torch.nn.functional as F
    from chaorec_amd import ops

    class _Spmm(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, csr):
            ctx.csr = csr
            return _oracle_spmm(csr, x)

        @staticmethod
        def backward(ctx, g):
            return _oracle_spmm(ctx.csr.t(), g.contiguous()), None

    def bpr(tab_u, tab_i, users, pos, neg, variant, reg):
        assert variant == ops.VARIANT_LOGSIGMOID and reg == 0.0
        s = (tab_u[users] * tab_i[pos]).sum(1) - (tab_u[users] * tab_i[neg]).sum(1)
        return (-torch.mean(F.logsigmoid(s)),)

    def keys(w, ids, seed, step):
        return torch.from_numpy(oracle_mod.race_keys(w.numpy(), seed, step, ids.numpy()).view(np.int64))

    class _LinearRows(torch.autograd.Function):
        """ops.linear_rows' contract on the CPU: (x W^T + b)[rows]; a claimed table's gradient leaves as gy [I, R] + W
        through the sink (optim.FusedAdam's protocol), an unclaimed one's as the dense tensor."""
        @staticmethod
        def forward(ctx, x, rows, weight, bias):
            xg = x.index_select(0, rows)
            ctx.save_for_backward(xg, rows, weight)
            ctx.x_param = x
            return F.linear(xg, weight, bias)

        @staticmethod
        def backward(ctx, gy):
            xg, rows, weight = ctx.saved_tensors
            x = ctx.x_param
            gy_full = torch.zeros((x.shape[0], gy.shape[1]), dtype=gy.dtype).index_add_(0, rows, gy)
            sink = getattr(x, "_chaorec_lowrank_sink", None)
            gx = None
            if sink is not None and sink.accepts(x):
                sink.submit(x, gy_full, weight, None)
            else:
                gx = gy_full @ weight
            return gx, None, gy.t() @ xg, gy.sum(0)

    return dict(spmm_fn=_oracle_spmm, mm_spmm_fn=lambda csr, x: _Spmm.apply(x, csr), bpr_fn=bpr,
                linear_rows_fn=_LinearRows.apply, keys_fn=keys)


class _CpuRowSink:
    """The optimizer side of the claimed-table protocol (optim.FusedAdam: accepts / submit / reduce_pending), in torch:
    after the ranks' sum, the dense gradient a claimed table WOULD have had is gy_full @ W."""
    lazy_rows = False

    def __init__(self, params):
        self._pending = {}
        self.ids = {id(p) for p in params}
        for p in params:
            p._chaorec_lowrank_sink = self

    def accepts(self, p):
        return id(p) in self.ids

    def submit(self, p, gy_full, weight, token=None):
        assert p not in self._pending
        self._pending[p] = [gy_full, weight, token]

    def reduce_pending(self, p, reduce_fn):
        cur = self._pending.get(p)
        if cur is not None:
            reduce_fn(cur[0])
            cur[2] = None

    def dense_gradient(self, p):
        gy_full, weight, _ = self._pending[p]
        return (gy_full @ weight).detach()


def _freedom_batch(m, rank, B=40):
    rng = np.random.default_rng(300 + rank)
    sel = rng.choice(len(m.local_edges), B, replace=False)
    users = torch.from_numpy(m.local_edges[sel, 0] - m.u0)
    pos = torch.from_numpy(m.local_edges[sel, 1] - m.num_user_global)
    neg = torch.from_numpy(rng.integers(0, m.num_item, B))
    return users, pos, neg


def _freedom_worker(rank, world, port, tmp, dropout, claimed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from chaorec_amd import dist as cdist
    from oracle import oracle as oracle_mod
    full, edges = _freedom_full(dropout)
    bounds = cdist.partition_users_by_nnz(np.bincount(edges[:, 0], minlength=full.num_user), world)
    m = cdist.ShardedFREEDOM(full, bounds, world, rank, torch.device("cpu"), **_freedom_standins(oracle_mod))
    tables = [m.text_embedding.weight, m.image_embedding.weight]
    sink = _CpuRowSink(tables) if claimed else None
    m.pre_epoch_processing()
    m.pre_epoch_processing()                               # the second epoch's draw (step 1)
    users, pos, neg = _freedom_batch(m, rank)
    m.zero_grad()
    loss = m.loss(users, pos, neg)
    loss.backward()
    m.sync_grads()
    if claimed:
        # no dense gradient was formed or exchanged for the tables: what travelled is gy [I, R]
        assert all(p.grad is None for p in tables) and not any(p is t for p in m.replicated_parameters() for t in tables)
        grads = {n: (sink.dense_gradient(p) if any(p is t for t in tables) else p.grad).numpy()
                 for n, p in m.named_parameters()}
    else:
        grads = {n: p.grad.numpy() for n, p in m.named_parameters()}
    kept = np.stack([m.shard.local_edges[:, 0] + m.u0, m.shard.local_edges[:, 1] - m.num_user], 1)
    np.savez(os.path.join(tmp, f"fr{rank}.npz"), u0=m.u0, u1=m.u1, loss=float(loss), users=users.numpy() + m.u0,
             pos=pos.numpy(), neg=neg.numpy(), res=m.result.detach().numpy(), kept=kept,
             **{"g_" + n: v for n, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dropout,claimed", [(0.2, False), (0.0, False), (0.2, True)])
def test_sharded_freedom_matches_single_process(oracle, dropout, claimed):
    """BASELINE north_star "FREEDOM-style": FREEDOM sharded by user rows (2 gloo ranks, CPU stand-ins for the kernels):
    the per-epoch pruning keeps exactly the single-process edge set (race keys numbered over the whole edge list, k-th
    smallest key by a distributed radix select), representations, loss, and every gradient -- user rows, the item rows
    summed inside the backward, the modality tables and transforms after sync_grads() -- equal the plain-torch
    restatement of Model/FREEDOM.py:164-217 on the whole graph.  `claimed`: the modality tables' gradients travel as
    gy [I, R] (an optimizer claimed them, chaorec_adam_lowrank_f32's protocol): their sum over the ranks, times W, is the
    same dense gradient -- at I x 64 instead of I x K floats per table on the wire."""
    from chaorec_amd import graph
    from chaorec_amd.Model.FREEDOM import FREEDOM
    from oracle.torch_ref import freedom_reference_loss
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_freedom_worker, world, (tmp, dropout, claimed), world)
        r = [dict(np.load(os.path.join(tmp, f"fr{k}.npz"))) for k in range(world)]
    full, edges = _freedom_full(dropout)
    U, I = full.num_user, full.num_item
    # single process: the second epoch's kept set (step 1) and its normalised, symmetrised graph (Model/FREEDOM.py:143-162)
    if dropout > 0:
        k = int(full.edge_values.numel() * (1 - dropout))
        keys = oracle.race_keys(full.edge_values.numpy(), 77, 1)
        keep = keys <= np.partition(keys, k - 1)[k - 1]
        assert keep.sum() == k
        ki = full.edge_indices[:, torch.from_numpy(keep)]
        kept_ranks = np.concatenate([x["kept"] for x in r], 0)
        assert {(int(a), int(b)) for a, b in kept_ranks} == {(int(a), int(b)) for a, b in ki.t().numpy()}
        vals = FREEDOM._normalize_adj_m(None, ki, torch.Size((U, I)))
    else:
        ki = full.edge_indices
        vals = 0.5 * FREEDOM._normalize_adj_m(None, ki, torch.Size((U, I)))     # (get_norm_adj_mat's doubled degrees, Q6)
    rows = torch.cat((ki[0], ki[1] + U))
    cols = torch.cat((ki[1] + U, ki[0]))
    full.masked_adj = graph.coo_to_csr_coalesced(rows, cols, torch.cat((vals, vals)), U + I, U + I, symmetric=True)
    users = torch.from_numpy(np.concatenate([x["users"] for x in r]))
    pos = torch.from_numpy(np.concatenate([x["pos"] for x in r])) + U
    neg = torch.from_numpy(np.concatenate([x["neg"] for x in r])) + U
    ref_loss, ref_res = freedom_reference_loss(full, users, pos, neg)
    ref_loss.backward()
    assert sum(x["loss"] for x in r) == pytest.approx(float(ref_loss.detach()), rel=1e-5)
    res_u = np.concatenate([x["res"][:x["u1"] - x["u0"]] for x in r], 0)
    assert np.allclose(res_u, ref_res[:U].detach().numpy(), rtol=1e-5, atol=1e-7)
    for x in r:
        assert np.allclose(x["res"][x["u1"] - x["u0"]:], ref_res[U:].detach().numpy(), rtol=1e-5, atol=1e-7)
    gu = np.concatenate([x["g_user_embedding.weight"] for x in r], 0)
    assert np.allclose(gu, full.user_embedding.weight.grad.numpy(), rtol=2e-4, atol=1e-9)
    for name, mod in (("item_embedding", full.item_embedding), ("text_embedding", full.text_embedding),
                      ("image_embedding", full.image_embedding), ("text_trs", full.text_trs), ("image_trs", full.image_trs)):
        for pn, p in mod.named_parameters():
            for x in r:
                got = x[f"g_{name}.{pn}"]
                assert np.allclose(got, p.grad.numpy(), rtol=2e-4, atol=1e-8), (name, pn)


def _worker_calibrate(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("CHAOREC_DIST_EXCHANGE", None)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chaorec_amd import dist as cdist
    cdist.AUTO_BIG_BYTES = 1024                        # (so that the test's small buffer counts as a large one)
    table = cdist.calibrate_exchange(203, 16, "cpu", candidates=("allreduce", "rs_ag", "direct"))
    assert all(table[m]["ok"] and table[m]["ms"] > 0 for m in ("allreduce", "rs_ag", "direct")), table
    buf = torch.zeros((cdist.padded_rows(203), 16))
    assert cdist.resolve_mode(buf) == table["chosen"]                    # `auto` now picks what was measured fastest
    assert cdist.resolve_mode(torch.zeros((4, 16))) == "allreduce"       # small buffers: one all-reduce
    # a mode whose sum is wrong is vetoed on EVERY rank, even when only one rank saw it differ
    real = cdist._sum_exchange_async

    def broken(b, group):
        pend = real(b, group)
        if cdist._FORCED[0] == "rs_ag" and rank == 1:
            pend.wait()
            b[0, 0] += 1.0
            return cdist._Pending(None)
        return pend

    cdist._sum_exchange_async = broken
    table = cdist.calibrate_exchange(203, 16, "cpu", candidates=("allreduce", "rs_ag"))
    cdist._sum_exchange_async = real
    assert table["allreduce"]["ok"] and not table["rs_ag"]["ok"] and table["chosen"] == "allreduce", table
    os.environ["CHAOREC_DIST_EXCHANGE"] = "rs_ag"
    assert cdist.resolve_mode(buf) == "allreduce" and "rs_ag" in cdist.exchange_mode_used()
    dist.barrier()
    dist.destroy_process_group()


def test_calibrate_exchange_checks_every_mode_against_all_reduce_and_vetoes_wrong_ones():
    """dist.calibrate_exchange: the first-contact check bench.py runs before it trusts an exchange mode on a node."""
    with tempfile.TemporaryDirectory() as tmp:
        _spawn(_worker_calibrate, 2, (tmp,), 2)
