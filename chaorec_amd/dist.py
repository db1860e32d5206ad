"""User-row sharding of the hot path over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-process; this is the MI355X-native scale-out of the same
arithmetic.  With A = [[0, B], [B^T, 0]] (B = normalised user x item block):
  * rank g owns a contiguous range of users (balanced by nnz), their embedding rows / Adam state, and the
    rows B_g of B; the item table [I, D] is replicated;
  * a layer is   y_u(g) = B_g x_i            -- local CSR SpMM, no communication
                 y_i    = sum_g B_g^T x_u(g)  -- local CSR SpMM into an [I, D] partial, then ONE all-reduce;
  * backward mirrors it (g_xu(g) = B_g g_yi local; g_xi = all-reduce of B_g^T g_yu(g)), which is also the
    "all-reduce on gradients" of the replicated item table;
  * BPR terms are evaluated by the rank that owns the user; the loss is the mean over ranks;
  * evaluation is embarrassingly parallel over users (each rank ranks its users against the replicated items).
Sums over ranks change the fp32 association, so N>1 equals N=1 to rounding (1e-6), not bit for bit.

`spmm_fn` is injectable so the communication pattern is covered by world_size-2 gloo tests on CPU
(tests/test_dist_gloo.py pass an oracle-backed stand-in; the product default is the HIP kernel).
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import graph, ops


def partition_users_by_nnz(user_deg, world):
    """Contiguous user ranges with ~equal edge counts (degrees are heavy-tailed: SURVEY 8(e))."""
    csum = np.concatenate([[0], np.cumsum(np.asarray(user_deg, dtype=np.int64))])
    total = csum[-1]
    bounds = [0]
    for g in range(1, world):
        bounds.append(int(np.searchsorted(csum, total * g / world, side="left")))
    bounds.append(len(user_deg))
    for g in range(1, len(bounds)):
        bounds[g] = max(bounds[g], bounds[g - 1])
    return bounds


class UserShard:
    """Rank-local graph blocks: `ui` rows = local users / cols = items, `iu` rows = items / cols = local users."""

    def __init__(self, edges, num_user, num_item, world, rank, device, self_loops=False):
        """self_loops: BasicGCN's D^-1/2 (A + I) D^-1/2 (BasicGCN.py:37-46): degrees count the loop, the loop's own
        weight 1/(d+1) is kept as the diagonals `diag_u` (local users) / `diag_i` (items)."""
        e = np.asarray(edges, dtype=np.int64)
        u, i = e[:, 0], e[:, 1] - num_user
        deg_u = np.bincount(u, minlength=num_user) + (1 if self_loops else 0)
        deg_i = np.bincount(i, minlength=num_item) + (1 if self_loops else 0)
        self.bounds = partition_users_by_nnz(deg_u - (1 if self_loops else 0), world)
        self.u0, self.u1 = self.bounds[rank], self.bounds[rank + 1]
        self.num_user_global, self.num_item, self.world, self.rank = num_user, num_item, world, rank
        sel = (u >= self.u0) & (u < self.u1)
        ul, il = u[sel] - self.u0, i[sel]
        # same fp32 normalisation as graph.lightgcn_csr, with GLOBAL degrees
        dis_u = torch.from_numpy(deg_u.astype(np.float32)).pow(-0.5)
        dis_i = torch.from_numpy(deg_i.astype(np.float32)).pow(-0.5)
        w = dis_u[torch.from_numpy(u[sel])] * dis_i[torch.from_numpy(il)]
        n_local = self.u1 - self.u0
        self.ui = graph.coo_to_csr(ul, il, w, n_local, num_item).to(device)        # y_u = B_g x_i
        self.iu = graph.coo_to_csr(il, ul, w, num_item, n_local).to(device)        # p_i = B_g^T x_u
        self.ui._t, self.iu._t = self.iu, self.ui
        self.local_edges = np.stack([ul, il + n_local], 1).astype(np.int32)        # shard-local id convention
        self.num_user_local = n_local
        self.diag_u = self.diag_i = None
        if self_loops:
            self.diag_u = (dis_u[self.u0:self.u1] * dis_u[self.u0:self.u1]).view(-1, 1).to(device)
            self.diag_i = (dis_i * dis_i).view(-1, 1).to(device)


def _all_reduce(t, group):
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _Pending:
    """Handle of an all-reduce issued with async_op=True (RCCL runs it on its own stream, so the SpMM launched
    next on the compute stream overlaps it; wait() makes the compute stream depend on the result)."""

    def __init__(self, work):
        self.work = work

    def wait(self):
        if self.work is not None:
            self.work.wait()


def _mean_all(x):
    """Scalar mean that survives hipGraph replay on the GPU (ops.mean_all); the gloo/CPU tests of the exchange logic
    run the same expression through torch."""
    return ops.mean_all(x) if x.is_cuda else x.mean()


def _all_reduce_async(t, group):
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES):
        return _Pending(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
    return _Pending(None)


# CHAOREC_FORCE_COLLECTIVES=1: issue the all-reduces on a 1-rank group too, so that a 1-GPU box exercises the
# RCCL launch (and hipGraph capture of it) that the N>1 job uses
import os as _os
_FORCE_COLLECTIVES = _os.environ.get("CHAOREC_FORCE_COLLECTIVES", "0") == "1"


class _ShardedLayerMean(torch.autograd.Function):
    """LightGCN.forward (Model/LightGCN.py:76-95) on a user shard: L x (2 local SpMM + 1 all-reduce).
    `spmm_fn(csr, x, **epilogue)` has ops.spmm_raw's keyword contract (alpha / z,beta / acc,acc_init,acc_w): the
    layer mean of the user rows and the `+ w G` terms of the backward ride in the SpMM epilogues."""

    @staticmethod
    def forward(ctx, xu, xi, shard, n_layers, spmm_fn, group):
        w = 1.0 / (n_layers + 1)
        xu, xi = xu.contiguous(), xi.contiguous()
        fu = torch.empty_like(xu) if n_layers else xu * w
        fi = xi * w
        cu, ci = xu, xi
        # Layer l+1's item partial B_g^T x_u only needs this rank's user rows of layer l, not the all-reduce of layer
        # l's partial: the wait for an all-reduce is therefore deferred until the user-row SpMM that consumes its result,
        # ONE LAYER LATER -- each exchange travels under two SpMMs (and next to the following exchange) instead of one
        pend = None                                     # all-reduce in flight for `ci`
        for l in range(n_layers):
            pi = spmm_fn(shard.iu, cu)
            pend_pi = _all_reduce_async(pi, group)
            if pend is not None:
                pend.wait()
                fi.add_(ci, alpha=w)                    # the previous layer's item rows join the layer mean
            yu = spmm_fn(shard.ui, ci, acc=fu, acc_init=xu if l == 0 else None, acc_w=w)
            cu, ci, pend = yu, pi, pend_pi
        if pend is not None:
            pend.wait()
            fi.add_(ci, alpha=w)
        ctx.shard, ctx.n_layers, ctx.w, ctx.spmm_fn, ctx.group = shard, n_layers, w, spmm_fn, group
        return fu, fi

    @staticmethod
    def backward(ctx, Gu, Gi):
        # Gi is this rank's PARTIAL gradient of the replicated item rows (its own batch terms); the rank sum is
        # folded into the per-layer all-reduce:  g_i <- allreduce(B_g^T g_u + w * Gi_partial)
        shard, L, w, spmm_fn, group = ctx.shard, ctx.n_layers, ctx.w, ctx.spmm_fn, ctx.group
        Gu, Gi = Gu.contiguous(), Gi.contiguous()
        Gi_full = Gi.clone()                            # layer-L seed needs the full item gradient
        pend = _all_reduce_async(Gi_full, group)
        gu, gi = Gu * w, Gi_full
        for it in range(L):                             # same deferral as in forward
            pi = spmm_fn(shard.iu, gu, z=Gi, beta=w)
            pend_pi = _all_reduce_async(pi, group)
            pend.wait()
            if it == 0:
                gi = Gi_full.mul_(w)
            nu = spmm_fn(shard.ui, gi, z=Gu, beta=w)
            gu, gi, pend = nu, pi, pend_pi
        pend.wait()
        if L == 0:
            gi = Gi_full.mul_(w)
        return gu, gi, None, None, None, None


def sharded_layer_mean_propagate(xu, xi, shard, n_layers, spmm_fn=None, group=None):
    return _ShardedLayerMean.apply(xu, xi, shard, n_layers, spmm_fn or ops.spmm_raw, group)


class ShardedLightGCN(nn.Module):
    """LightGCN on one user shard.  Ids are shard-local: users [0, U_g), items U_g + [0, I) (the reference's
    'global item id = item + num_user' convention, per shard).  Item parameters are replicated: their
    gradient leaves backward already summed over ranks, so every rank applies the same Adam update."""

    def __init__(self, shard, user_item_dict_local, dim_E, reg_weight, n_layers, device, seed=42, spmm_fn=None,
                 bpr_fn=None, group=None):
        super().__init__()
        self.shard, self.device, self.group = shard, device, group
        self.num_user, self.num_item = shard.num_user_local, shard.num_item
        self.reg_weight, self.n_layers, self.dim_embedding = reg_weight, n_layers, dim_E
        self.user_item_dict = user_item_dict_local
        self.spmm_fn, self.bpr_fn = spmm_fn, bpr_fn
        # one global initialisation, sliced: identical to the single-GPU model under the same seed
        g = torch.Generator().manual_seed(seed)
        bound_u = (6.0 / (shard.num_user_global + dim_E)) ** 0.5     # nn.init.xavier_uniform_ bounds
        bound_i = (6.0 / (shard.num_item + dim_E)) ** 0.5
        full_u = (torch.rand(shard.num_user_global, dim_E, generator=g) * 2 - 1) * bound_u
        full_i = (torch.rand(shard.num_item, dim_E, generator=g) * 2 - 1) * bound_i
        self.user_embedding = nn.Embedding.from_pretrained(full_u[shard.u0:shard.u1].clone(), freeze=False)
        self.item_embedding = nn.Embedding.from_pretrained(full_i, freeze=False)
        rowptr, col = graph.user_hist_csr(user_item_dict_local, self.num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self.graph = shard.ui
        self.result_u = self.result_i = None

    def forward(self):
        fu, fi = sharded_layer_mean_propagate(self.user_embedding.weight, self.item_embedding.weight, self.shard,
                                              self.n_layers, self.spmm_fn, self.group)
        self.result_u, self.result_i = fu, fi
        return fu, fi

    @property
    def result(self):
        """[U_g + I, D] in the reference's row convention (users then items), built on demand."""
        return None if self.result_u is None else torch.cat((self.result_u, self.result_i), 0)

    def loss(self, users, pos_items, neg_items):
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        fu, fi = self.forward()
        bpr = self.bpr_fn or ops.bpr_loss
        out = bpr(fu, fi, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, self.reg_weight)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        return out[0] / world          # global loss = mean over ranks; item grads are SUMMED by the all-reduce

    def loss_local(self, users, pos_items, neg_items):
        """loss() for device batches that already hold LOCAL item ids (ops.draw_batch)."""
        fu, fi = self.forward()
        bpr = self.bpr_fn or ops.bpr_loss
        out = bpr(fu, fi, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, self.reg_weight)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        return out[0] / world

    def gene_ranklist(self, topk=50, gather=False):
        """Rank this shard's users against the replicated item table; ids are GLOBAL (item + U_global)."""
        with torch.no_grad():
            idx, _ = ops.score_topk(self.result_u.detach(), self.result_i.detach(), self.hist, 1e-6, topk,
                                    id_offset=self.shard.num_user_global)
        if gather and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return gather_ranklists(idx, self.shard, self.group)
        return idx.cpu()

    def local_user_ids(self, users):
        return users


def gather_ranklists(idx_local, shard, group=None):
    """all_gather of the per-rank [U_g, K] lists into [U, K] in user order (no exchange inside the scoring)."""
    world = dist.get_world_size(group)
    K = idx_local.shape[1]
    sizes = [shard.bounds[g + 1] - shard.bounds[g] for g in range(world)]
    pad = max(sizes)
    buf = torch.zeros((pad, K), dtype=idx_local.dtype, device=idx_local.device)
    buf[:idx_local.shape[0]] = idx_local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0).cpu()


class WeakScalingJob:
    pass


def build_weak_scaling_job(U1, I, E1, world, rank, D, L, reg, device, seed=42):
    """bench.py at N GPUs: one synthetic graph with N x U1 users over the same I items (every rank generates it
    from the same seed), sharded by user rows; per-rank work stays that of the N=1 configuration."""
    from .synthetic import synthetic_interactions
    U = U1 * world
    edges = synthetic_interactions(U, I, E1 * world, seed=seed)
    shard = UserShard(edges, U, I, world, rank, device)
    uid_local = graph.user_item_dict_from_edges(shard.local_edges)
    for u in range(shard.num_user_local):
        uid_local.setdefault(u, [])
    job = WeakScalingJob()
    job.model = ShardedLightGCN(shard, uid_local, D, reg, L, device, seed=seed).to(device)
    job.local_edges = shard.local_edges
    job.num_user_local = shard.num_user_local
    job.shard = shard
    job.local_user_ids = lambda users: users
    return job


# ---------------------------------------------------------------------------------------------------------------------
# MMGCN (BASELINE configs[3]): the same row sharding for a model with dense layers between the propagations.
# Convention for everything REPLICATED (item rows, the Linear weights): a rank's autograd gradient is a PARTIAL -- the
# part of dL/d(.) that flows through this rank's users -- and the true gradient is the sum over ranks.  Row-wise ops
# (Linear, leaky_relu, normalize, concat) need nothing; the propagation is the one op that mixes rows:
#   forward   y_u(g) = B_g x_i + d_u x_u(g)                 y_i = sum_g B_g^T x_u(g) + d_i x_i     (one all-reduce)
#   backward  g_xu(g) = B_g (sum_g' G_yi(g')) + d_u G_yu(g)   (one all-reduce of the partial item gradient)
#             g_xi(g) = B_g^T G_yu(g) + d_i G_yi(g)           (stays partial)
# and after backward() the Linear weights' partial gradients are summed once (`allreduce_grads`).
# ---------------------------------------------------------------------------------------------------------------------
class _ShardedPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xu, xi, shard, spmm_fn, group):
        xu, xi = xu.contiguous(), xi.contiguous()
        pi = spmm_fn(shard.iu, xu)
        pending = _all_reduce_async(pi, group)          # item partials travel while the user rows are computed
        yu = spmm_fn(shard.ui, xi)
        if shard.diag_u is not None:
            yu.addcmul_(xu, shard.diag_u)
        pending.wait()
        if shard.diag_i is not None:
            pi.addcmul_(xi, shard.diag_i)
        ctx.shard, ctx.spmm_fn, ctx.group = shard, spmm_fn, group
        return yu, pi

    @staticmethod
    def backward(ctx, Gyu, Gyi):
        shard, spmm_fn, group = ctx.shard, ctx.spmm_fn, ctx.group
        Gyu, Gyi = Gyu.contiguous(), Gyi.contiguous()
        tot = Gyi.clone()
        pending = _all_reduce_async(tot, group)
        gxi = spmm_fn(shard.iu, Gyu)                    # partial: this rank's users only
        if shard.diag_i is not None:
            gxi.addcmul_(Gyi, shard.diag_i)
        pending.wait()
        gxu = spmm_fn(shard.ui, tot)
        if shard.diag_u is not None:
            gxu.addcmul_(Gyu, shard.diag_u)
        return gxu, gxi, None, None, None


class ShardedGraph:
    """The graph operator BasicGCN.forward accepts in place of an edge_index: x = [local users; all items] rows."""

    def __init__(self, shard, spmm_fn=None, group=None):
        self.shard, self.spmm_fn, self.group = shard, spmm_fn, group

    def propagate(self, x):
        n = self.shard.num_user_local
        yu, yi = _ShardedPropagate.apply(x[:n], x[n:], self.shard, self.spmm_fn or ops.spmm_raw, self.group)
        return torch.cat((yu, yi), 0)


def allreduce_grads(params, group=None):
    """Sum the ranks' partial gradients of the replicated parameters: one flat bucket, one all-reduce."""
    ps = [p for p in params if p.grad is not None]
    if not ps or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVES):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


class ShardedMMGCN(nn.Module):
    """MMGCN (Model/MMGCN.py) on one user shard, built from a single-process chaorec_amd MMGCN so that every rank
    starts from the same weights and the slices of the same preference / id_embedding tensors.  Ids are shard-local:
    users [0, U_g), items U_g + [0, I).  After loss.backward() call sync_grads() before optimizer.step()."""

    def __init__(self, full, shard, device, spmm_fn=None, group=None):
        super().__init__()
        import copy
        self.shard, self.device, self.group = shard, device, group
        self.num_user, self.num_item = shard.num_user_local, shard.num_item
        self.reg_weight = full.reg_weight
        U, u0, u1 = shard.num_user_global, shard.u0, shard.u1
        op = ShardedGraph(shard, spmm_fn, group)

        def take(t):       # [U + I, d] or [U, d] global rows -> this shard's layout
            t = t.detach().cpu()
            return (torch.cat((t[u0:u1], t[U:]), 0) if t.shape[0] > U else t[u0:u1]).clone().to(device)

        def shard_gcn(g):
            g = copy.deepcopy(g)
            g.edge_index, g.num_user, g.device = op, self.num_user, device
            g.preference = take(g.preference)
            return g.to(device)

        self.v_gcn, self.t_gcn = shard_gcn(full.v_gcn), shard_gcn(full.t_gcn)
        self.v_feat, self.t_feat = full.v_feat.detach().to(device), full.t_feat.detach().to(device)
        self.id_embedding = take(full.id_embedding)
        rowptr, col = graph.user_hist_csr(graph.user_item_dict_from_edges(shard.local_edges), self.num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self.result = None

    def forward(self):
        rep = (self.v_gcn(self.v_feat, self.id_embedding) + self.t_gcn(self.t_feat, self.id_embedding)) / 2
        self.result = rep
        return rep

    def loss(self, user_tensor, item_tensor, bpr_fn=None):
        """Model/MMGCN.py:188-202 on this rank's (u, pos, neg) triples; the global loss is the mean over ranks."""
        users = user_tensor[:, 0].contiguous().to(self.device)
        pos, neg = item_tensor[:, 0].contiguous().to(self.device), item_tensor[:, 1].contiguous().to(self.device)
        out = self.forward()
        bpr = bpr_fn or ops.bpr_loss
        loss = bpr(out, None, users, pos, neg, ops.VARIANT_LOG_SIGMOID, 0.0, item_offset=0)[0]
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        with torch.no_grad():  # the reported regulariser constant (Q2), this rank's share of it
            ut, it = user_tensor.reshape(-1).to(self.device), item_tensor.reshape(-1).to(self.device)
            mean = _mean_all                # (not .mean(): multi-block torch reductions break under hipGraph replay)
            pref = self.v_gcn.preference
            reg = mean(self.id_embedding[ut] ** 2 + self.id_embedding[it] ** 2) / world + \
                mean(pref ** 2) * (pref.shape[0] / self.shard.num_user_global)
        return loss / world + self.reg_weight * reg     # sum over ranks = the single-process loss

    def sync_grads(self):
        allreduce_grads(self.parameters(), self.group)

    def gene_ranklist(self, topk=50, gather=False):
        with torch.no_grad():
            r = self.result.detach()
            idx, _ = ops.score_topk(r[:self.num_user], r[self.num_user:], self.hist, 1e-5, topk,
                                    id_offset=self.shard.num_user_global)
        if gather and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return gather_ranklists(idx, self.shard, self.group)
        return idx.cpu()
