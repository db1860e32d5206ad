"""User-sharded MMGCN and FREEDOM (BASELINE configs[3] / [2]): the propagate over a user shard as an autograd
node, the gradient bucket of the replicated parameters, the sharded model classes.  Moved out of dist.py in round 5
 with no behaviour change; `chaorec_amd.dist` re-exports every name here."""
import os as _os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, graph, ops  # noqa: F401
from .dist import (  # noqa: F401  (dist.py imports this module at its END: every name below exists by then)
    UserShard, _active, _all_reduce, _count, _mean_all, _sum_exchange_async, exchange_buffer, joined_shard_csr,
    padded_rows, side_group)
from .dist_lightgcn import gather_ranklists, sharded_layer_mean_propagate  # noqa: F401


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
        pbuf, pi = exchange_buffer(xi.shape[0], xu.shape[1], xu, group)
        spmm_fn(shard.iu, xu, y=pi)
        pending = _sum_exchange_async(pbuf, group)      # item partials travel while the user rows are computed
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
        tbuf, tot = exchange_buffer(Gyi.shape[0], Gyi.shape[1], Gyi, group)
        tot.copy_(Gyi)
        pending = _sum_exchange_async(tbuf, group)
        gxi = spmm_fn(shard.iu, Gyu)                    # partial: this rank's users only
        if shard.diag_i is not None:
            gxi.addcmul_(Gyi, shard.diag_i)
        pending.wait()
        gxu = spmm_fn(shard.ui, tot)
        if shard.diag_u is not None:
            gxu.addcmul_(Gyu, shard.diag_u)
        return gxu, gxi, None, None, None


def joined_loop_csr(shard):
    """joined_shard_csr(shard) plus the self-loop weight of every LOCAL USER row on its diagonal (BasicGCN's
    D^-1/2 (A + I) D^-1/2, BasicGCN.py:37-46; the loop entry last in its row, as the reference appends it).  The item
    rows' loop term is NOT in the matrix: every rank would add it to its partial -- it is added once, after the exchange."""
    if getattr(shard, "_joined_loop", None) is None:
        base = joined_shard_csr(shard)
        U, N = shard.num_user_local, shard.num_user_local + shard.num_item
        dev = base.col.device
        counts = base.rowptr[1:] - base.rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(N, device=dev), counts)
        loop_rows = torch.arange(U, device=dev)
        # stable sort by row of (entries..., loops): a row's loop lands behind its entries
        all_rows = torch.cat((rows, loop_rows))
        order = torch.argsort(all_rows, stable=True)
        col = torch.cat((base.col, loop_rows.to(torch.int32)))[order].contiguous()
        val = torch.cat((base.val, shard.diag_u.view(-1).to(base.val.dtype)))[order].contiguous()
        rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(all_rows, minlength=N), 0, out=rowptr[1:])
        # symmetric up to the missing item-row loops: A^T = A on what is stored
        shard._joined_loop = graph.CSR(rowptr, col, val, N, N, symmetric=True)
    return shard._joined_loop


class _ShardedPropagateJoined(torch.autograd.Function):
    """_ShardedPropagate on the rank's JOINED table [local users; items] -- one SpMM launch per direction instead of two
    block launches, two diagonal updates and a concatenation:
        forward   y = A_g x  (user rows complete incl. their loop term; item rows = this rank's partial) -> exchange of
                  the item rows in place -> + d_i x_i once
        backward  S = [G_u; sum over ranks of G_i]  ->  g = A_g S  (user rows: B_g G_i_total + d_u G_u; item rows: B_g^T G_u,
                  PARTIAL as the convention demands) -> item rows += d_i G_i(partial)"""

    @staticmethod
    def forward(ctx, x, shard, spmm_fn, group, sync):
        ctx.shard, ctx.spmm_fn, ctx.group, ctx.sync = shard, spmm_fn, group, sync
        return propagate_joined_fwd(x, shard, spmm_fn, group, sync)

    @staticmethod
    def backward(ctx, G):
        return propagate_joined_bwd(G, ctx.shard, ctx.spmm_fn, ctx.group, ctx.sync), None, None, None, None


def propagate_joined_fwd(x, shard, spmm_fn, group, sync=False):
    """_ShardedPropagateJoined's forward as a plain function (also called by ops.mmgcn_layer's node).  sync: the
    synchronous collective form (_sum_exchange_async), for callers whose compute runs on several streams."""
    x = x.contiguous()
    U, N, D = shard.num_user_local, x.shape[0], x.shape[1]
    csr = joined_loop_csr(shard)
    buf = torch.empty((U + padded_rows(N - U, group), D), dtype=x.dtype, device=x.device)
    if buf.shape[0] > N:
        buf[N:].zero_()
    spmm_fn(csr, x, y=buf[:N])
    _sum_exchange_async(buf[U:], group, sync).wait()
    y = buf[:N]
    y[U:].addcmul_(x[U:], shard.diag_i)
    return y


def propagate_joined_bwd(G, shard, spmm_fn, group, sync=False):
    G = G.contiguous()
    U, N, D = shard.num_user_local, G.shape[0], G.shape[1]
    S = torch.empty((U + padded_rows(N - U, group), D), dtype=G.dtype, device=G.device)
    if S.shape[0] > N:
        S[N:].zero_()
    S[:N].copy_(G)
    _sum_exchange_async(S[U:], group, sync).wait()
    g = spmm_fn(joined_loop_csr(shard), S[:N])
    g[U:].addcmul_(G[U:], shard.diag_i)
    return g


class ShardedGraph:
    """The graph operator BasicGCN.forward accepts in place of an edge_index: x = [local users; all items] rows.
    CHAOREC_DIST_PROPAGATE=blocks restores the two-block form (_ShardedPropagate: the item exchange travels under the
    user-row SpMM there; one launch more per direction, a concatenation and two diagonal updates)."""

    def __init__(self, shard, spmm_fn=None, group=None):
        self.shard, self.spmm_fn, self.group = shard, spmm_fn, group
        self.joined = shard.diag_u is not None and _os.environ.get("CHAOREC_DIST_PROPAGATE", "joined") == "joined"
        self.sync = False      # synchronous collectives (a model whose compute runs on several streams sets it: _sum_exchange_async)

    def propagate_raw(self, x):
        """A x without an autograd node (joined form only): for nodes that own their backward (ops.mmgcn_layer)."""
        return propagate_joined_fwd(x, self.shard, self.spmm_fn or ops.spmm_raw, self.group, self.sync)

    def propagate_t_raw(self, g):
        return propagate_joined_bwd(g, self.shard, self.spmm_fn or ops.spmm_raw, self.group, self.sync)

    def propagate(self, x):
        if self.joined:
            return _ShardedPropagateJoined.apply(x, self.shard, self.spmm_fn or ops.spmm_raw, self.group, self.sync)
        n = self.shard.num_user_local
        yu, yi = _ShardedPropagate.apply(x[:n], x[n:], self.shard, self.spmm_fn or ops.spmm_raw, self.group)
        return torch.cat((yu, yi), 0)


class GradBucket:
    """The replicated parameters' gradients as views of ONE persistent flat buffer: autograd accumulates into the
    views in place, the rank sum is one all-reduce of the buffer -- no per-step concatenation or copy-back
    (optimizers must not drop the gradients: zero() instead of zero_grad(set_to_none=True))."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def attached(self):
        o = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + o * self.flat.element_size():
                return False
            o += p.numel()
        return True

    def zero(self):
        if not self.attached():            # someone ran zero_grad(set_to_none=True): hook the views up again
            o = 0
            for p in self.params:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
                o += p.numel()
        self.flat.zero_()

    def all_reduce(self, group=None):
        if not self.attached():
            raise RuntimeError("GradBucket: a gradient no longer lives in the bucket (zero_grad(set_to_none=True)?)")
        _all_reduce(self.flat, group)


def allreduce_grads(params, group=None):
    """Sum the ranks' partial gradients of the replicated parameters (one flat bucket, one all-reduce) for callers
    without a GradBucket: concatenates and copies back."""
    ps = [p for p in params if p.grad is not None]
    if not ps or not _active(group):
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


# ShardedMMGCN's two modality branches on two streams (CHAOREC_DIST_MMGCN_STREAMS=0: one stream).  Round 3 had this "dump
# core" under capture; tools/rccl_streams_repro.py (profiles/r04_f_rccl_streams_repro.txt) narrowed it down: `async_op=True`
# collectives from a second capturing stream segfault, and so does every collective hopped onto a third "communication"
# stream; the synchronous form issued by the branch's own stream captures and replays fine.
# Round 5 made it opt-in because its equality test against the one-stream step diverged once in ~10 runs inside the suite (the
# visual branch's weights 2e-4 off after six steps).  Round 6 found the cause, and it was not the streams: the BPR backward's
# fp32 atomic adds are applied in an order that moves with the load on the chip, the ONE-stream step was as irreproducible
# (tools/stream_stress.py, profiles/r06_stream_bisect.txt: 50 of 50 runs inexact with atomics, 0 of 400 with the ordered
# backward launch, one stream or two).  Default on again: 4.19 against 4.63 ms per step at microlens.
SHARDED_MMGCN_STREAMS_DEFAULT = "1"


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
        op = self._graph_op = ShardedGraph(shard, spmm_fn, group)
        # The visual branch gets a process group -- an RCCL communicator -- of ITS OWN: in two-stream mode its exchanges are
        # issued from the side stream while the textual branch's run from the main one, and two collectives of ONE
        # communicator must never be in flight at the same time (c10d runs a synchronous collective on the caller's
        # stream: two streams = two concurrent kernels on the communicator's buffers; seen once in ~10 runs as a step
        # with slightly wrong gradients).  Collective: every rank builds its ShardedMMGCN at the same point.
        self.group_v = side_group(group)
        op_v = self._graph_op_v = op if self.group_v is group else ShardedGraph(shard, spmm_fn, self.group_v)

        def take(t):       # [U + I, d] or [U, d] global rows -> this shard's layout
            t = t.detach().cpu()
            return (torch.cat((t[u0:u1], t[U:]), 0) if t.shape[0] > U else t[u0:u1]).clone().to(device)

        def shard_gcn(g, graph_op):
            g = copy.deepcopy(g)
            g.edge_index, g.num_user, g.device = graph_op, self.num_user, device
            g.preference = take(g.preference)
            return g.to(device)

        self.v_gcn, self.t_gcn = shard_gcn(full.v_gcn, op_v), shard_gcn(full.t_gcn, op)
        self.v_feat, self.t_feat = full.v_feat.detach().to(device), full.t_feat.detach().to(device)
        self.id_embedding = take(full.id_embedding)
        rowptr, col = graph.user_hist_csr(graph.user_item_dict_from_edges(shard.local_edges), self.num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self.result = None
        self._bucket = None

    def forward(self):
        import importlib
        _mm = importlib.import_module(__package__ + ".Model.MMGCN")        # (the package re-exports the CLASS under this name)
        streams = _os.environ.get("CHAOREC_DIST_MMGCN_STREAMS", SHARDED_MMGCN_STREAMS_DEFAULT) == "1" and _mm.BRANCH_STREAMS \
            and self._graph_op.joined
        if streams and self.id_embedding.is_cuda:
            # The two modality branches are independent until the mean: the visual one on a side stream, like the
            # single-process model (Model/MMGCN.py forward; autograd replays every node's backward on its forward's
            # stream).  Each branch issues its own exchanges from its own stream -- in c10d's SYNCHRONOUS form: that is what
            # survives a capture from two streams on this stack (_sum_exchange_async).  RCCL queues the collectives on its
            # own stream in host order, the same on every rank.
            cur = torch.cuda.current_stream()
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=self.id_embedding.device)
            self._graph_op.sync = self._graph_op_v.sync = True     # (the backward's exchanges, run by autograd later, too)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                v_rep = self.v_gcn(self.v_feat, self.id_embedding)
            t_rep = self.t_gcn(self.t_feat, self.id_embedding)
            cur.wait_stream(self._side_stream)
        else:
            self._graph_op.sync = self._graph_op_v.sync = False
            v_rep = self.v_gcn(self.v_feat, self.id_embedding)
            t_rep = self.t_gcn(self.t_feat, self.id_embedding)
        rep = (v_rep + t_rep) / 2
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

    def zero_grad(self, set_to_none=False):
        """Gradients live in one persistent flat bucket (GradBucket): they are zeroed in place, never dropped."""
        if self._bucket is None:
            self._bucket = GradBucket(list(self.parameters()))
        self._bucket.zero()

    def sync_grads(self):
        if self._bucket is not None and self._bucket.attached():
            self._bucket.all_reduce(self.group)
        else:
            allreduce_grads(self.parameters(), self.group)

    def gene_ranklist(self, topk=50, gather=False):
        from . import ranking
        want_gather = gather and dist.is_initialized() and dist.get_world_size(self.group) > 1
        idx = ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-5, topk, to_cpu=not want_gather,
                                    state=ranking.state_of(self), id_offset=self.shard.num_user_global)
        return gather_ranklists(idx, self.shard, self.group) if want_gather else idx


# ---------------------------------------------------------------------------------------------------- FREEDOM
class _SumGradAcrossRanks(torch.autograd.Function):
    """Identity whose gradient is summed over the ranks: a replicated tensor feeding a rank-local branch of the loss."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        if _active(ctx.group):
            _count(g)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def global_kth_smallest(keys, k, group=None):
    """The k-th smallest (1-based) of the ranks' int64 keys taken together, keys >= 0 (entries < 0 never count): a radix
    select, 4 digits of 16 bits, one all-reduce of a 65 536-bin histogram per digit -- no rank sees another's keys."""
    keys = keys[keys >= 0]
    prefix, want = 0, int(k)
    for shift in (48, 32, 16, 0):
        digit = (keys >> shift) & 0xFFFF
        hist = torch.bincount(digit, minlength=65536)
        if _active(group):
            dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        csum = torch.cumsum(hist, 0)
        d = int(torch.searchsorted(csum, torch.tensor([want], device=csum.device, dtype=csum.dtype))[0])
        if d >= 65536:
            raise ValueError("global_kth_smallest: k exceeds the number of keys")
        want -= int(csum[d - 1]) if d > 0 else 0
        prefix |= d << shift
        keys = keys[digit == d]
    return prefix


class ShardedFREEDOM(nn.Module):
    """FREEDOM (Model/FREEDOM.py) on one user shard, built from a single-process chaorec_amd FREEDOM (or any object
    with its attributes) so that every rank starts from the same weights.  Users are sharded by rows like LightGCN's;
    everything on the item side -- item embeddings, the modality tables and their transforms, the item-item kNN graph
    mm_adj and its SpMM -- is replicated: the item-item propagation runs on every rank (no exchange), and the partial
    gradients of the replicated parameters are summed (item rows inside the backward, the rest in one flat bucket:
    sync_grads()).  Ids are shard-local: users [0, U_g), items [0, I) as in FREEDOM.loss after its own offset.

    The per-epoch degree-sensitive pruning (Model/FREEDOM.py:143-162) keeps the k edges with the smallest race keys of
    the WHOLE edge list: every rank computes the keys of its own edges numbered as in the whole list
    (chaorec_weighted_sample_keys), the k-th smallest key over all ranks comes from global_kth_smallest, and the kept
    set is exactly the single-process one for the same seed.  The pruned shard is rebuilt per rank from its kept edges
    (item degrees of the pruned graph by one all-reduce: UserShard.from_local)."""

    def __init__(self, full, bounds, world, rank, device, group=None, spmm_fn=None, mm_spmm_fn=None, bpr_fn=None,
                 linear_rows_fn=None, keys_fn=None, prune_seed=None):
        super().__init__()
        import copy
        self.device, self.group, self.world, self.rank = device, group, world, rank
        self.bounds = [int(b) for b in bounds]
        self.u0, self.u1 = self.bounds[rank], self.bounds[rank + 1]
        self.num_user, self.num_item = self.u1 - self.u0, full.num_item
        self.num_user_global = full.num_user
        self.n_layers, self.mm_layers = full.n_layers, full.mm_layers
        self.reg_weight, self.dropout = full.reg_weight, full.dropout
        self.user_embedding = nn.Embedding(self.num_user, full.user_embedding.weight.shape[1])
        with torch.no_grad():
            self.user_embedding.weight.copy_(full.user_embedding.weight[self.u0:self.u1])
        self.item_embedding = copy.deepcopy(full.item_embedding)
        self.text_embedding, self.image_embedding = copy.deepcopy(full.text_embedding), copy.deepcopy(full.image_embedding)
        self.text_trs, self.image_trs = copy.deepcopy(full.text_trs), copy.deepcopy(full.image_trs)
        self.mm_adj = full.mm_adj.to(device)
        self.to(device)
        # this rank's share of the edge list, with the edges' numbers in the whole list
        ei = full.edge_indices.cpu()
        mine = ((ei[0] >= self.u0) & (ei[0] < self.u1)).nonzero().flatten()
        self.edge_ids = mine.to(device)
        self.local_edges = np.stack([ei[0][mine].numpy(), ei[1][mine].numpy() + self.num_user_global], 1).astype(np.int64)
        self.edge_values = full.edge_values.detach().cpu()[mine].to(device)
        self.n_edges_global = int(ei.shape[1])
        self._spmm_fn = spmm_fn or ops.spmm_raw
        self._mm_spmm = mm_spmm_fn or ops.spmm
        self._bpr = bpr_fn or ops.bpr_loss
        self._linear_rows = linear_rows_fn or ops.linear_rows
        # the modality tables are read only through the batch rows of their projection (FREEDOM.loss): an optimizer that
        # claims them (optim.FusedAdam) gets gy [I, R] + W instead of the dense [I, K] gradient -- and the ranks then sum
        # THAT in sync_grads(): 2 x I x 64 floats per step over xGMI instead of I x (4096 + 384)
        self.image_embedding.weight._chaorec_projected_only = True
        self.text_embedding.weight._chaorec_projected_only = True
        self._batch_idx = self._loss_w = None
        self._keys_fn = keys_fn or ops.weighted_sample_keys
        self._prune_seed = int(prune_seed if prune_seed is not None else getattr(full, "_prune_seed", 0))
        self._prune_calls = 0
        rowptr, col = graph.user_hist_csr(graph.user_item_dict_from_edges(
            np.stack([self.local_edges[:, 0] - self.u0, self.local_edges[:, 1] - self.num_user_global + self.num_user], 1)),
            self.num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self.shard = None            # the (pruned) graph of this epoch
        self.result = None
        self._bucket = None
        if self.dropout <= .0:
            self._set_shard(self.local_edges, scale=0.5)

    def _set_shard(self, kept_local_edges, scale=1.0):
        sh = UserShard.from_local(kept_local_edges, self.bounds, self.num_item, self.world, self.rank, self.device,
                                  group=self.group)
        if scale != 1.0:
            # dropout == 0 trains on get_norm_adj_mat's graph: degrees counted over the bidirectional list (Q6), i.e.
            # (2 d_u)^-1/2 (2 d_i)^-1/2 = half of the values of the pruned graphs' normalisation
            sh.ui.val.mul_(scale)
            sh.iu.val.mul_(scale)
        self.shard = sh

    def pre_epoch_processing(self):
        """Model/FREEDOM.py:143-162 on the sharded edge list."""
        if self.dropout <= .0:
            return
        k = int(self.n_edges_global * (1. - self.dropout))
        keys = self._keys_fn(self.edge_values, self.edge_ids, self._prune_seed, self._prune_calls)
        self._prune_calls += 1
        kth = global_kth_smallest(keys, k, self.group)
        keep = ((keys >= 0) & (keys <= kth)).cpu().numpy()
        self._set_shard(self.local_edges[keep])

    def forward(self):
        xu, xi = self.user_embedding.weight, self.item_embedding.weight
        fu, fi = sharded_layer_mean_propagate(xu, xi, self.shard, self.n_layers, self._spmm_fn, self.group)
        h = _SumGradAcrossRanks.apply(xi, self.group)      # the item-item branch: replicated compute, summed gradient
        for _ in range(self.mm_layers):
            h = self._mm_spmm(self.mm_adj, h)
        ig = fi + h
        self._result_parts = (fu.detach(), ig.detach())
        return fu, ig

    @property
    def result(self):
        """[U_g + I, D]: concatenated when read (never cached: under a captured step the halves are static buffers)."""
        if self._result_parts is not None:
            return torch.cat(self._result_parts, 0)
        return None

    @result.setter
    def result(self, value):
        self._result_parts = None if value is None else (value[:self.num_user], value[self.num_user:])

    def loss(self, users, pos_items, neg_items):
        """Model/FREEDOM.py:194-217 on this rank's triples (local user ids, item ids in [0, I)); the global loss is the
        mean over ranks."""
        users, pos, neg = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        ua, ia = self.forward()
        V = ops.VARIANT_LOGSIGMOID
        B = users.shape[0]
        rows = torch.cat((pos, neg), 0)
        if self._batch_idx is None or self._batch_idx[0].shape[0] != B or self._batch_idx[0].device != users.device:
            idx = torch.arange(B, device=users.device)
            self._batch_idx = (idx, idx + B)
        idx, idx_neg = self._batch_idx
        tf = self._linear_rows(self.text_embedding.weight, rows, self.text_trs.weight, self.text_trs.bias)
        vf = self._linear_rows(self.image_embedding.weight, rows, self.image_trs.weight, self.image_trs.bias)
        if self._bpr is ops.bpr_loss and ua.is_cuda:
            # the three terms share the user table and the batch's users: ONE autograd node, as in Model/FREEDOM.py here
            if self._loss_w is None or self._loss_w.device != users.device:
                self._loss_w = torch.tensor([1.0, self.reg_weight, self.reg_weight], dtype=torch.float32, device=users.device)
            total = ops.bpr_loss_multi(ua, users, V, [(ia, pos, neg), (tf, idx, idx_neg), (vf, idx, idx_neg)], self._loss_w,
                                       gathered=[None, (rows, self.num_item), (rows, self.num_item)])
        else:
            total = self._bpr(ua, ia, users, pos, neg, V, 0.0)[0]
            total = total + self.reg_weight * (self._bpr(ua, tf, users, idx, idx_neg, V, 0.0)[0] +
                                               self._bpr(ua, vf, users, idx, idx_neg, V, 0.0)[0])
        return total / self.world

    def _claimed_tables(self):
        """The modality tables an optimizer has claimed (their gradient travels as gy [I, R], see __init__)."""
        out = []
        for p in (self.text_embedding.weight, self.image_embedding.weight):
            sink = getattr(p, "_chaorec_lowrank_sink", None)
            if sink is not None and sink.accepts(p):
                out.append((p, sink))
        return out

    def replicated_parameters(self):
        """Parameters every rank holds whose DENSE gradients are partial after backward (item_embedding's is already
        summed; a claimed modality table has no dense gradient)."""
        claimed = {id(p) for p, _ in self._claimed_tables()}
        return [p for m in (self.text_embedding, self.image_embedding, self.text_trs, self.image_trs)
                for p in m.parameters() if id(p) not in claimed]

    def zero_grad(self, set_to_none=False):
        if self._bucket is None:
            self._bucket = GradBucket(self.replicated_parameters())
        self._bucket.zero()
        for p in (self.user_embedding.weight, self.item_embedding.weight):
            if p.grad is not None:
                p.grad.zero_()

    def sync_grads(self):
        if self._bucket is not None and self._bucket.attached():
            self._bucket.all_reduce(self.group)
        else:
            allreduce_grads(self.replicated_parameters(), self.group)
        # claimed modality tables: the ranks' batches touch different rows -- sum the [I, R] row gradients (the update
        # g = gy W is linear in gy), and let the optimizer find the touched rows in the sum, not in this rank's batch
        for p, sink in self._claimed_tables():
            sink.reduce_pending(p, lambda t: _all_reduce(t, self.group))

    def gene_ranklist(self, topk=50, gather=False):
        from . import ranking
        want_gather = gather and dist.is_initialized() and dist.get_world_size(self.group) > 1
        idx = ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=not want_gather,
                                    state=ranking.state_of(self), id_offset=self.num_user_global)
        return gather_ranklists(idx, self, self.group) if want_gather else idx
