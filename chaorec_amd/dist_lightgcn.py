"""User-sharded LightGCN: the autograd form (ShardedLightGCN) and the fused, captured training step over a
user shard (FusedShardedLightGCNStep: joined / split launches, row-sparse backward, light forward, frontier exchanges).
Moved out of dist.py in round 5 with no behaviour change; `chaorec_amd.dist` re-exports every name here.
The shard, the exchange modes and their calibration, the capture helpers stay in dist.py."""
import os as _os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, graph, ops  # noqa: F401
from .dist import (  # noqa: F401  (dist.py imports this module at its END: every name below exists by then)
    MODES_USED, STATS, UserShard, _Pending, _active, _count, _p2p_usable, _sum_exchange_async, capture_mode,
    capture_with_retry, exchange_buffer, joined_shard_csr, padded_rows, resolve_mode, settle_before_capture)


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
        pend = None                                     # exchange in flight for `ci`
        I, D = xi.shape
        for l in range(n_layers):
            pbuf, pi = exchange_buffer(I, D, xi, group)
            spmm_fn(shard.iu, cu, y=pi)
            pend_pi = _sum_exchange_async(pbuf, group)
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
        I, D = Gi.shape
        gbuf, Gi_full = exchange_buffer(I, D, Gi, group)  # layer-L seed needs the full item gradient
        Gi_full.copy_(Gi)
        pend = _sum_exchange_async(gbuf, group)
        gu, gi = Gu * w, Gi_full
        for it in range(L):                             # same deferral as in forward
            pbuf, pi = exchange_buffer(I, D, Gi, group)
            spmm_fn(shard.iu, gu, y=pi, z=Gi, beta=w)
            pend_pi = _sum_exchange_async(pbuf, group)
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
                 bpr_fn=None, group=None, global_init=True):
        super().__init__()
        self.shard, self.device, self.group = shard, device, group
        self.num_user, self.num_item = shard.num_user_local, shard.num_item
        self.reg_weight, self.n_layers, self.dim_embedding = reg_weight, n_layers, dim_E
        self.user_item_dict = user_item_dict_local
        self.spmm_fn, self.bpr_fn = spmm_fn, bpr_fn
        bound_u = (6.0 / (shard.num_user_global + dim_E)) ** 0.5     # nn.init.xavier_uniform_ bounds
        bound_i = (6.0 / (shard.num_item + dim_E)) ** 0.5
        if global_init:
            # one global initialisation, sliced: identical to the single-GPU model under the same seed
            g = torch.Generator().manual_seed(seed)
            full_u = (torch.rand(shard.num_user_global, dim_E, generator=g) * 2 - 1) * bound_u
            mine_u = full_u[shard.u0:shard.u1].clone()
        else:
            # per-rank: the item table from the shared seed (replicated), this rank's user rows from its own stream --
            # no rank materialises the [U_global, D] table (5 GB at config 5)
            g = torch.Generator().manual_seed(seed)
            gu = torch.Generator().manual_seed(seed * 1_000_003 + 1 + shard.rank)
            mine_u = (torch.rand(self.num_user, dim_E, generator=gu) * 2 - 1) * bound_u
        full_i = (torch.rand(shard.num_item, dim_E, generator=g) * 2 - 1) * bound_i
        self.user_embedding = nn.Embedding.from_pretrained(mine_u, freeze=False)
        self.item_embedding = nn.Embedding.from_pretrained(full_i, freeze=False)
        rowptr, col = (graph.user_hist_csr(user_item_dict_local, self.num_user) if user_item_dict_local is not None
                       else graph.user_hist_csr_from_edges(shard.local_edges, self.num_user))
        self.hist = (rowptr.to(device), col.to(device))
        self.graph = shard.ui
        self.result_u = self.result_i = self._result_cat = None

    def forward(self):
        fu, fi = sharded_layer_mean_propagate(self.user_embedding.weight, self.item_embedding.weight, self.shard,
                                              self.n_layers, self.spmm_fn, self.group)
        self.result_u, self.result_i, self._result_cat = fu, fi, None
        return fu, fi

    @property
    def result(self):
        """[U_g + I, D] in the reference's row convention (users then items): concatenated when it is read.  Not cached:
        under a captured step the two halves are static buffers that every replay rewrites without any Python running."""
        if self.result_u is None:
            return None
        return torch.cat((self.result_u, self.result_i), 0)

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
        from . import ranking
        if self.result_u is None:
            raise RuntimeError("ShardedLightGCN.gene_ranklist: no propagated table -- the last training step was a light one "
                               "(FusedShardedLightGCNStep with light_forward: only its batch's rows were computed).  Run the "
                               "step before an evaluation with full_result=True (FusedShardedLightGCNStep.run does), or call "
                               "forward() first")
        want_gather = gather and dist.is_initialized() and dist.get_world_size(self.group) > 1
        # (the shared evaluation path: carried thresholds from call to call, the list written straight to pinned memory)
        idx = ranking.gene_ranklist(self.result_u, self.num_user, self.num_item, self.hist, 1e-6, topk,
                                    to_cpu=not want_gather, state=ranking.state_of(self),
                                    id_offset=self.shard.num_user_global, items=self.result_i)
        return gather_ranklists(idx, self.shard, self.group) if want_gather else idx

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


class _HipStepKernels:
    """The launches FusedShardedLightGCNStep is made of, on the MI355X.  tests/test_dist_gloo.py injects an oracle-backed
    stand-in with the same methods (the product has no CPU kernels)."""
    spmm = staticmethod(lambda *a, **k: ops.spmm_raw(*a, **k))       # (looked up per call: bench.py records the step's SpMMs)
    spmm_mean = staticmethod(ops.spmm_mean_raw)
    spmm_adam = staticmethod(ops.spmm_adam_raw)
    rows_mean = staticmethod(ops.rows_mean)
    adam_step = staticmethod(ops.adam_step)
    bpr_fwd_bwd = staticmethod(ops.bpr_fwd_bwd)
    bpr_finalize = staticmethod(ops.bpr_finalize)
    mean_terms_limit = staticmethod(ops.mean_terms_limit)
    # the row-sparse backward (sparse_bwd)
    expand_row_bits = staticmethod(ops.expand_row_bits)
    spmm_rowlist = staticmethod(lambda *a, **k: ops.spmm_rowlist_raw(*a, **k))
    spmm_rowsparse = staticmethod(lambda *a, **k: ops.spmm_rowsparse_raw(*a, **k))
    zero_rows_by_bits = staticmethod(ops.zero_rows_by_bits)
    rows_copy_by_bits = staticmethod(ops.rows_copy_by_bits)
    or_words = staticmethod(ops.or_words)
    # the light forward
    batch_rows = staticmethod(ops.batch_rows)
    rows_list_from_bits = staticmethod(ops.rows_list_from_bits)
    frontier_pack = staticmethod(ops.frontier_pack)
    frontier_unpack = staticmethod(ops.frontier_unpack)
    rows_mean_by_bits = staticmethod(ops.rows_mean_by_bits)
    long_row_buffers = staticmethod(ops.long_row_buffers)
    sparse_widths = (64, 256)           # chaorec_spmm_csr_rowsparse_f32 / _rowlist_f32 are built for these D


SPLIT_BYTES = int(_os.environ.get("CHAOREC_DIST_SPLIT_BYTES", str(64 << 20)))     # item partial size from which a step splits


class FusedShardedLightGCNStep:
    """optim.FusedLightGCNStep for a user-row shard: one training iteration of Model/LightGCN.py:76-135 +
    train_and_evaluate.py:43-48 on rank g's users as a fixed launch sequence -- no autograd tape, no optimizer launch for
    the user rows, Adam for them in the last backward propagate's epilogue:

        L x   SpMM over the rank's JOINED graph [[0, B_g], [B_g^T, 0]] (user rows complete, item rows = this rank's
              partial) + the exchange that sums the item rows over the ranks, in place
        1 x   layer mean of the item rows (chaorec_rows_mean_f32; the user rows' mean rides in the last SpMM's epilogue)
        1 x   BPR forward + backward on the rank's batch (its users only), gradient rows into G; 1 x the loss scalar
        1 x   copy of G + exchange of its item rows: the seed of the backward needs the item gradient of ALL ranks
        L-1 x SpMM  g_l = A_g g_{l+1} + (w / world) G   (the epilogue adds this rank's PARTIAL item gradient: the
              exchange sums it with the others') + exchange
        last: SpMM over B_g^T (item rows) -> exchange, travelling under the SpMM over B_g (user rows) with the Adam
              epilogue; then one fused Adam launch on the replicated item rows (the same update on every rank)

    2 L + 5 launches + 2 L + 1 exchanges (the autograd path: 4 L SpMMs, ~3 L elementwise launches, the four-kernel BPR,
    two Adam launches).  The global loss is the mean over the ranks' batch losses, so every gradient carries 1 / world:
    folded into the epilogue factors.  Item rows end identical on every rank (same sums, same Adam arithmetic)."""

    def __init__(self, model, optimizer, batch_size=1024, edges=None, seed=42, step_dev=None, given_batch=False,
                 loss_accum=None, capture=True, kernels=None, group=None, steps_per_replay=1, split=None, sparse_bwd=None,
                 light_forward=None):
        """split: None = by size (item partial I_pad * D * 4 >= SPLIT_BYTES, or CHAOREC_DIST_SPLIT=0/1), True / False =
        the split / joined launch sequence (see _launch_split).  sparse_bwd: None = by size (optim.FusedLightGCNStep's
        rule: CHAOREC_SPARSE_BACKWARD=auto/0/1, CHAOREC_SPARSE_BACKWARD_MIN_ROWS), True / False = the first two backward
        propagates over the batch's frontier only / dense (split launch sequence only).  light_forward: None = with the
        row-sparse backward (CHAOREC_LIGHT_FORWARD=0 switches it off), True / False: optim.FusedLightGCNStep's light step for
        a shard -- the last two forward layers over the frontier's rows only (see _launch_split), model.result_u / result_i
        withheld until a step with full_result=True."""
        from .optim import FusedAdam
        if not isinstance(optimizer, FusedAdam) or len(optimizer.param_groups) != 1:
            raise TypeError("FusedShardedLightGCNStep needs a FusedAdam with one parameter group")
        if model.n_layers < 1:
            raise ValueError("FusedShardedLightGCNStep: n_layers >= 1")
        if (edges is None) == (not given_batch):
            raise ValueError("FusedShardedLightGCNStep: either edges (in-launch draw) or given_batch=True")
        self.model, self.optimizer, self.B, self.L = model, optimizer, int(batch_size), model.n_layers
        self.K = kernels or _HipStepKernels
        self.group = group if group is not None else model.group
        self.edges, self.seed, self.step_dev, self.loss_accum = edges, seed, step_dev, loss_accum
        shard = model.shard
        U, I = shard.num_user_local, shard.num_item
        uw, iw = model.user_embedding.weight, model.item_embedding.weight
        if [id(p) for p in optimizer.param_groups[0]["params"]] != [id(uw), id(iw)]:
            raise ValueError("FusedShardedLightGCNStep: the optimizer must hold exactly the two embedding tables")
        D = uw.shape[1]
        dev = uw.device
        self.U, self.I, self.N, self.D = U, I, U + I, D
        self.N_pad = U + padded_rows(I, self.group)
        # the two tables as views of ONE [U + I_pad, D] buffer (same Parameters): the joined graph's operand
        flat = torch.zeros((self.N_pad, D), dtype=torch.float32, device=dev)
        flat[:U].copy_(uw.data)
        flat[U:U + I].copy_(iw.data)
        uw.data, iw.data = flat[:U], flat[U:U + I]
        self.flat = flat
        optimizer.make_moments_adjacent([uw, iw])          # (existing moments are migrated into one buffer, not refused)
        st_u, st_i = optimizer.state[uw], optimizer.state[iw]
        self.m = torch.as_strided(st_u["exp_avg"], (self.N, D), (D, 1))
        self.v = torch.as_strided(st_u["exp_avg_sq"], (self.N, D), (D, 1))
        new = lambda: torch.zeros((self.N_pad, D), dtype=torch.float32, device=dev)       # (pad rows stay zero)
        self.ybuf = [new() for _ in range(max(self.L, 1))]          # x_1 .. x_L, then the backward's g buffers
        self.final, self.G, self.S = new(), new(), new()
        self.csr = joined_shard_csr(shard)
        self.ids = tuple(torch.zeros(self.B, dtype=torch.int64, device=dev) for _ in range(3))
        self.coef = torch.empty(self.B, dtype=torch.float32, device=dev)
        self.ws = torch.empty(4 * self.B, dtype=torch.float32, device=dev)
        self.out = torch.zeros(3, dtype=torch.float32, device=dev)
        self.static_loss = torch.zeros((), dtype=torch.float32, device=dev)    # this rank's batch loss (global = mean over ranks)
        self.bc = torch.ones(2, dtype=torch.float32, device=dev)
        self.use_mean = self.L <= self.K.mean_terms_limit(D)
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if split is None:
            env = _os.environ.get("CHAOREC_DIST_SPLIT", "")
            split = (env == "1") if env in ("0", "1") else (self.N_pad - U) * D * 4 >= SPLIT_BYTES
        self.split = bool(split)
        # Row-sparse backward (optim.FusedLightGCNStep's, for a shard): the batch gradient has B user rows and <= 2 B item
        # rows per rank, the first backward propagate's result lives in their neighbours.  One bitmap per side and level:
        # bits = [users R0, items R0, users N1, items N1]; the ITEM bitmaps are made the union over the ranks (the seed and
        # every item partial are sums over the ranks), the user ones are local.
        if sparse_bwd is None:
            mode = _os.environ.get("CHAOREC_SPARSE_BACKWARD", "auto")
            lo, hi = getattr(self.K, "sparse_widths", (1, 0))
            sparse_bwd = self.split and lo <= D <= hi and D % 4 == 0 and self.L >= 2 and mode != "0" and \
                (mode == "1" or self.N >= int(_os.environ.get("CHAOREC_SPARSE_BACKWARD_MIN_ROWS", "400000")))
        if sparse_bwd and not self.split:
            raise ValueError("FusedShardedLightGCNStep: the row-sparse backward exists for the split launch sequence only")
        self.sparse_bwd = bool(sparse_bwd)
        if light_forward is None:
            light_forward = self.sparse_bwd and 2 <= self.L <= 4 and _os.environ.get("CHAOREC_LIGHT_FORWARD", "auto") != "0"
        if light_forward and not (self.sparse_bwd and 2 <= self.L <= 4):
            raise ValueError("FusedShardedLightGCNStep: the light forward needs the row-sparse backward and 2 <= n_layers <= 4")
        self.light = bool(light_forward)
        self.result_complete = True
        self.graph_full = None
        self._compact = {}                   # (buffer, cap) -> ([cap, D] packed rows, bitmap prefix): _exchange_frontier
        self._cap0 = min(I, 2 * self.B * self.world)      # the batch items of all ranks: a static bound
        self._frontier_overflow = torch.zeros(1, dtype=torch.int32, device=dev)    # (sticky: check_frontier())
        self._cap1 = None                    # capacity of N1's compact frontier exchange: _auto_frontier_cap
        if self.sparse_bwd:
            wu, wi = (U + 31) // 32, (I + 31) // 32
            self._wu = wu
            self._bits_all = torch.zeros(2 * (wu + wi) + 5, dtype=torch.int32, device=dev)     # (+ the five lists' lengths)
            cut = [0, wu, wu + wi, 2 * wu + wi, 2 * (wu + wi)]
            self.bits = [self._bits_all[cut[k]:cut[k + 1]] for k in range(4)]
            self._list_n = self._bits_all[cut[4]:]
            self._list_u = torch.zeros(U, dtype=torch.int32, device=dev)
            self._list_i = torch.zeros(I, dtype=torch.int32, device=dev)
            self._long_ui = self.K.long_row_buffers(shard.ui)
            self._long_iu = self.K.long_row_buffers(shard.iu)
            if self.light:
                # R0's local users / R0's items of ALL ranks / N1's items of ALL ranks: every rank computes its partial of
                # every frontier item row (the other ranks' users may neighbour it)
                self._list0_u = torch.zeros(self.B, dtype=torch.int32, device=dev)
                self._list0_i = torch.zeros(min(I, 2 * self.B * self.world), dtype=torch.int32, device=dev)
                self._list1_ig = torch.zeros(I, dtype=torch.int32, device=dev)
                self.Z0 = torch.zeros((self.N_pad - U, D), dtype=torch.float32, device=dev)   # layer L's frontier partial
            self._bits_gather = torch.zeros((self.world, wi), dtype=torch.int32, device=dev)
            # the first backward item partial: non-zero in the frontier's rows only, ALL-ZERO between steps (its exchange
            # sums whole buffers; the rows a step wrote are zeroed again by that step)
            self.Z = torch.zeros((self.N_pad - U, D), dtype=torch.float32, device=dev)
        self.replays = 0
        self.graph = self.graph1 = None
        # k steps per hipGraph (in-launch batches only): a replay boundary costs ~5.5 us on this stack, the launches inside
        # a graph follow each other without a gap
        self.steps_per_replay = int(steps_per_replay) if (capture and edges is not None) else 1
        if capture:
            for c in (self.csr, shard.ui, shard.iu):
                c.schedule(D)                   # lazily built by the first SpMM: must exist before capture
            # whatever happens below (a capture that raises included), the model, the Adam moments and the step / loss
            # counters leave this constructor as they entered it: a caller that falls back to capture=False then starts
            # from the same state as the captured run would have
            saved = self._save_state()
            try:
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._launch()                  # eager first: communicators are set up outside capture
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self._restore_state(saved)
                if self.light:
                    with torch.cuda.stream(s):
                        self._launch(light=False)       # (eager first, like the light one above)
                    torch.cuda.current_stream().wait_stream(s)
                    torch.cuda.synchronize()
                    self._restore_state(saved)
                def capture_all():                      # (every eager launch is behind us: captures only from here on)
                    settle_before_capture()
                    self.graph1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph1, capture_error_mode=capture_mode()):
                        self._launch()
                    self.graph = self.graph1
                    if self.light:
                        self.graph_full = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.graph_full, capture_error_mode=capture_mode()):
                            self._launch(light=False)
                    if self.steps_per_replay > 1:
                        self.graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.graph, capture_error_mode=capture_mode()):
                            for _ in range(self.steps_per_replay):
                                self._launch()

                def reset():
                    self.graph = self.graph1 = self.graph_full = None
                    torch.cuda.synchronize()
                    self._restore_state(saved)

                self.capture_attempts = capture_with_retry(capture_all, reset, what="FusedShardedLightGCNStep")
                for gph in {id(g_): g_ for g_ in (self.graph1, self.graph, self.graph_full) if g_ is not None}.values():
                    gph.replay()
                torch.cuda.synchronize()
            finally:
                torch.cuda.synchronize()
                self._restore_state(saved)

    def check_frontier(self):
        """Raise if a compact frontier exchange ever saw more flagged rows than its static capacity (the pack kernel drops
        the rows past it: the step would have trained on an incomplete sum).  Costs a sync: run() calls it once at its
        end, tests and the bench after their steps."""
        over = int(self._frontier_overflow.item())
        if over > 0:
            raise RuntimeError(f"FusedShardedLightGCNStep: a compact frontier exchange overflowed its capacity by {over} rows "
                               f"(batch items of all ranks: 2 * batch * world = {self._cap0} rows; N1's item frontier: "
                               f"{self._cap1} rows, twice the first step's -- CHAOREC_DIST_FRONTIER_CAP sets it, 0 = dense "
                               f"exchange): the batch size changed, a later frontier outgrew the first step's by more than "
                               f"2 x, or a row bitmap was not cleared after an aborted replay")

    def _counters(self):
        return [t for t in (self.step_dev, self.loss_accum, self.optimizer._step_dev) if t is not None]

    def _save_state(self):
        return ([self.flat.clone(), self.m.clone(), self.v.clone()], [t.clone() for t in self._counters()])

    def _restore_state(self, saved):
        with torch.no_grad():
            for dst, src in zip((self.flat, self.m, self.v), saved[0]):
                dst.copy_(src)
            for dst, src in zip(self._counters(), saved[1]):
                dst.copy_(src)
            self.G.zero_()
            if self.sparse_bwd:
                self._bits_all.zero_()
                self.Z.zero_()
                self.S.zero_()
                if self.light:
                    self.Z0.zero_()

    def _union_item_bits(self, bits):
        """An item-row bitmap becomes the union over the ranks (one small all-gather + one launch; issued BEFORE the
        step's large exchanges: a process group's collectives run in issue order)."""
        if _active(self.group):
            dist.all_gather_into_tensor(self._bits_gather.view(-1), bits, group=self.group)
            self.K.or_words(bits, self._bits_gather)

    def _exchange(self, buf):
        """Sum the item rows of a joined buffer over the ranks, in place; -> a handle to wait on."""
        return _sum_exchange_async(buf[self.U:], self.group)

    def _exchange_frontier(self, buf, bits, cap=None):
        """The same for a FRONTIER buffer of item rows ([I_pad, D], all-zero on every rank outside the rows flagged in
        `bits`, a bitmap united over the ranks).  The p2p exchange moves the flagged rows only.  RCCL's collectives cannot
        skip rows -- but where the frontier has a STATIC bound `cap` on its size (the batch items of all ranks: the seed
        of the backward, the last forward layer's item partial) the flagged rows are packed in bitmap order (the same order
        on every rank) into a [cap, D] buffer, THAT is all-reduced, and the sums are written back: 2 B world rows instead of
        the item table.  N1's item frontier (cap="auto") has no such bound, but a step cannot size a collective on the device
        either: its capacity is fixed at FIRST CONTACT (_auto_frontier_cap: twice what the first, eager step's frontier
        needed) and a later frontier that outgrows it is recorded by the pack launch and raised by check_frontier() at the
        end of run() -- never exchanged incompletely in silence.  cap=None: the dense exchange (the buffer is zero outside
        the frontier)."""
        if not _active(self.group):
            return _Pending(None)
        p2p = buf.is_cuda and resolve_mode(buf) == "p2p" and _p2p_usable(buf, self.group)
        if cap == "auto" and not p2p:
            cap = self._auto_frontier_cap(bits)
        if cap is None or cap == "auto" or p2p or _os.environ.get("CHAOREC_DIST_COMPACT_FRONTIER", "1") != "1":
            return _sum_exchange_async(buf, self.group, bits=bits, n_rows=self.I)
        K, I = self.K, self.I
        key = (buf.data_ptr(), int(cap))
        if key not in self._compact:
            self._compact[key] = (torch.zeros((int(cap), self.D), dtype=torch.float32, device=buf.device),
                                  torch.zeros((I + 31) // 32 + 1, dtype=torch.int32, device=buf.device))
        compact, prefix = self._compact[key]
        K.frontier_pack(buf[:I], bits, prefix, compact, overflow=self._frontier_overflow)
        _count(compact)
        MODES_USED.add("compact-allreduce")
        STATS["frontier_exchanges"] = STATS.get("frontier_exchanges", 0) + 1
        work = dist.all_reduce(compact, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def unpack():
            K.frontier_unpack(buf[:I], bits, prefix, compact)
            return _Pending(None)

        return _Pending(work, unpack)

    def _auto_frontier_cap(self, bits):
        """Capacity (rows) of the compact exchange of N1's item frontier, fixed the first time one is exchanged: the rows
        flagged in that frontier's bitmap -- the same bitmap on every rank (it is the union over the ranks), hence the same
        number everywhere -- doubled, rounded up to 1024, at most the item table.  CHAOREC_DIST_FRONTIER_CAP = rows fixes it
        by hand, = 0 keeps N1's frontiers on the dense exchange.  None while unknown and not knowable (inside a capture
        before any eager step: the constructor's eager warm-up steps come first)."""
        if self._cap1 is None:
            env = _os.environ.get("CHAOREC_DIST_FRONTIER_CAP")
            if env is not None:
                self._cap1 = min(self.I, int(env))
            elif torch.cuda.is_available() and bits.is_cuda and torch.cuda.is_current_stream_capturing():
                return None
            else:
                words = bits.detach().cpu().numpy().view(np.uint8)
                flagged = int(np.unpackbits(words).sum())
                self._cap1 = min(self.I, max(4096, (2 * flagged + 1023) // 1024 * 1024))
        return self._cap1 if self._cap1 > 0 else None

    @torch.no_grad()
    def _launch(self, light=None):
        K, model, opt, L, B, D = self.K, self.model, self.optimizer, self.L, self.B, self.D
        U, I, N = self.U, self.I, self.N
        if model.user_embedding.weight.data_ptr() != self.flat.data_ptr() or \
                model.item_embedding.weight.data_ptr() != self.flat[U:].data_ptr():
            # (model.to(...) / weight.data = ... after this step was built: it would train a buffer nobody reads)
            raise RuntimeError("FusedShardedLightGCNStep: the model's embedding tables were re-allocated after the step was "
                               "built; build a new step")
        if self.split:
            return self._launch_split(self.light if light is None else bool(light))
        group = opt.param_groups[0]
        shard, csr, w = model.shard, self.csr, 1.0 / (L + 1)
        xs = [self.flat]
        for l in range(L):
            y = self.ybuf[l]
            if l == L - 1 and self.use_mean:        # the user rows' layer mean in this launch's epilogue
                K.spmm_mean(csr, xs[-1][:N], [t[:N] for t in xs], w, self.final[:N], y=y[:N])
            else:
                K.spmm(csr, xs[-1][:N], y=y[:N])
            self._exchange(y).wait()
            xs.append(y)
        # the item rows' propagated values arrived with the exchanges, after the launches that could have averaged them
        lo = U if self.use_mean else 0
        K.rows_mean([t[lo:N] for t in xs], w, self.final[lo:N])
        draw = self.edges is not None
        K.bpr_fwd_bwd(self.final, U, self.G, B, ops.VARIANT_LOG_SIGMOID_EPS, model.reg_weight, self.coef, self.ws, self.ids,
                      edges=self.edges, hist=model.hist if draw else None, num_user=U, num_item=I, seed=self.seed, step=0,
                      step_dev=self.step_dev, adam_step=opt._step_dev, betas=group["betas"], adam_bc=self.bc)
        K.bpr_finalize(self.ws, B, D, model.reg_weight, self.out, out_total=self.static_loss, loss_accum=self.loss_accum,
                       advance=self.step_dev if draw else None)
        # backward.  S = [G_u; sum over ranks of G_i]: what the first propagate gathers from; its epilogue (and every
        # later one) adds this rank's PARTIAL item gradient G_i, which the exchange then sums with the others'.
        c = w / self.world
        self.S.copy_(self.G)
        self._exchange(self.S).wait()
        g, alpha = self.S, c
        for l in range(L - 1):
            y = self.ybuf[l & 1]
            K.spmm(csr, g[:N], y=y[:N], alpha=alpha, z=self.G[:N], beta=c)
            self._exchange(y).wait()
            g, alpha = y, 1.0
        Y = self.ybuf[min(L - 1, 2)]
        K.spmm(shard.iu, g[:U], y=Y[U:N], alpha=alpha, z=self.G[U:N], beta=c)
        pend = self._exchange(Y)                    # travels under the user rows' launch
        K.spmm_adam(shard.ui, g[U:N], self.flat[:U], self.m[:U], self.v[:U], self.bc, group["lr"], group["betas"],
                    group["eps"], group["weight_decay"], alpha=alpha, z=self.G[:U], beta=c, clear_z=True)
        pend.wait()
        K.adam_step(self.flat[U:N], Y[U:N], self.m[U:N], self.v[U:N], 0, group["lr"], group["betas"], group["eps"],
                    group["weight_decay"], step_dev=opt._step_dev)
        self.G[U:N].zero_()
        model.result_u, model.result_i, model._result_cat = self.final[:U], self.final[U:N], None

    @torch.no_grad()
    def _launch_split(self, light=False):
        """The same step with every joined launch cut into its two row blocks, so that EVERY exchange travels under
        compute (large item tables: config 5's 1 GB item partial takes longer over xGMI than the SpMM that produced it).
        Layer l + 1's item partial B_g^T x_u(l) needs only this rank's user rows of layer l -- not the exchanged item rows
        of layer l -- so per layer:

            SpMM over B_g^T (item partial of layer l+1)  ->  exchange l+1 starts
            wait for exchange l                           (it travelled under the two launches issued since it started)
            SpMM over B_g (user rows of layer l+1, gathers the now complete item rows of layer l)

        and the backward mirrors it (the gradient seed's exchange travels under the first B_g^T launch).  2 launches per
        layer and direction instead of 1 (4.4 us each: nothing against a millisecond exchange, too much at sports size --
        hence by size).  Row for row the same sums in the same order as the joined launches: bit-identical results.

        light: optim.FusedLightGCNStep's light step for a shard.  The batch is drawn first (its rows R0 flagged; the item
        bitmaps made the union over the ranks), R0 expanded to N1 on both sides, and the forward runs
            layers 1 .. L-2  dense, as above
            layer  L-1       B_g^T over the list of N1's items OF ALL RANKS (every rank owes its partial of every frontier
                             item row) into the frontier buffer Z -> exchanged as such; B_g over the list of N1's local users
            layer  L         B_g^T over the list of R0's items of all ranks into Z0 -> exchange; B_g over the list of R0's local
                             users with their layer mean in the epilogue; the item rows' mean by bitmap when Z0 has arrived
        -- the same arithmetic for every row it computes; model.result_u / result_i are withheld (only R0's rows exist)."""
        K, model, opt, L, B, D = self.K, self.model, self.optimizer, self.L, self.B, self.D
        U, I, N = self.U, self.I, self.N
        group = opt.param_groups[0]
        shard, w = model.shard, 1.0 / (L + 1)
        ui, iu = shard.ui, shard.iu
        draw = self.edges is not None
        sp = self.sparse_bwd
        if sp:
            bu0, bi0, bu1, bi1 = self.bits
        if light:
            n_u1, n_i1, n_u0, n_i0, n_i1g = (self._list_n[k:k + 1] for k in range(5))
            K.batch_rows(self.ids, self._bits_all, 32 * self._wu, edges=self.edges, hist=model.hist if draw else None,
                         num_user=U, num_item=I, seed=self.seed, step=0, step_dev=self.step_dev)
            self._frontier_bitmaps_and_lists()
            K.rows_list_from_bits(bu0, U, self._list0_u, n_u0)
            K.rows_list_from_bits(bi0, I, self._list0_i, n_i0)
            K.rows_list_from_bits(bi1, I, self._list1_ig, n_i1g)
        xs, pend = [self.flat], None
        for l in range(L - 2 if light else L):
            x, y = xs[-1], self.ybuf[l]
            K.spmm(iu, x[:U], y=y[U:N])
            nxt = self._exchange(y)
            if pend is not None:
                pend.wait()                          # x's item rows are the sum over the ranks from here on
            if l == L - 1 and self.use_mean:         # (x_L's user rows feed nothing but the mean: not stored)
                K.spmm_mean(ui, x[U:N], [t[:U] for t in xs], w, self.final[:U])
            else:
                K.spmm(ui, x[U:N], y=y[:U])
            pend = nxt
            xs.append(y)
        if light:
            x, y = xs[-1], self.ybuf[L - 2]
            # layer L-1 over N1
            K.spmm_rowlist(iu, x[:U], self.Z[:I], self._list1_ig, n_i1g, long_rows=self._long_iu)
            pz = self._exchange_frontier(self.Z, bi1, cap="auto")
            if pend is not None:
                pend.wait()
            K.spmm_rowlist(ui, x[U:N], y[:U], self._list_u, n_u1, long_rows=self._long_ui)
            # layer L over R0
            K.spmm_rowlist(iu, y[:U], self.Z0[:I], self._list0_i, n_i0, long_rows=self._long_iu)
            pz0 = self._exchange_frontier(self.Z0, bi0, cap=self._cap0)
            pz.wait()
            K.spmm_rowlist(ui, self.Z[:I], None, self._list0_u, n_u0, mean_out=self.final[:U],
                           mean_terms=[t[:U] for t in xs] + [y[:U]], mean_w=w, long_rows=self._long_ui)
            pz0.wait()
            K.rows_mean_by_bits([t[U:N] for t in xs] + [self.Z[:I], self.Z0[:I]], w, self.final[U:N], bi0)
            K.zero_rows_by_bits(self.Z[:I], bi1)             # (both frontier buffers had their readers: all-zero again)
            K.zero_rows_by_bits(self.Z0[:I], bi0)
            K.bpr_fwd_bwd(self.final, U, self.G, B, ops.VARIANT_LOG_SIGMOID_EPS, model.reg_weight, self.coef, self.ws, self.ids,
                          num_user=U, num_item=I, adam_step=opt._step_dev, betas=group["betas"], adam_bc=self.bc)
        else:
            pend.wait()
            lo = U if self.use_mean else 0
            K.rows_mean([t[lo:N] for t in xs], w, self.final[lo:N])
            flags = dict(row_bits=self._bits_all, bits_item_offset=32 * self._wu) if sp else {}
            K.bpr_fwd_bwd(self.final, U, self.G, B, ops.VARIANT_LOG_SIGMOID_EPS, model.reg_weight, self.coef, self.ws, self.ids,
                          edges=self.edges, hist=model.hist if draw else None, num_user=U, num_item=I, seed=self.seed, step=0,
                          step_dev=self.step_dev, adam_step=opt._step_dev, betas=group["betas"], adam_bc=self.bc, **flags)
        K.bpr_finalize(self.ws, B, D, model.reg_weight, self.out, out_total=self.static_loss, loss_accum=self.loss_accum,
                       advance=self.step_dev if draw else None)
        c = w / self.world
        if sp and not light:
            self._frontier_bitmaps_and_lists()
        # the seed: this rank's user rows as they are (G), the item rows summed over the ranks (S) while the first B_g^T
        # launch runs.  Row-sparse: S's item rows are a frontier buffer like Z (the batch items' rows copied in, exchanged as
        # such, zeroed again after their one reader)
        if sp:
            K.rows_copy_by_bits(self.S[U:N], self.G[U:N], bi0)
            pend = self._exchange_frontier(self.S[U:], bi0, cap=self._cap0)
        else:
            self.S[U:].copy_(self.G[U:])
            pend = self._exchange(self.S)
        gu, gi, alpha = self.G[:U], self.S[U:N], c
        for l in range(L):
            last = l == L - 1
            Y = self.ybuf[min(L - 1, 2)] if last else self.ybuf[l & 1]
            how = "dense" if (not sp or last or l >= 2) else "list" if (l == 0 and L >= 3) else "gated"
            if how == "list":
                K.spmm_rowlist(iu, gu, self.Z[:I], self._list_i, self._list_n[1:2], alpha=alpha, z=self.G[U:N], beta=c,
                               src_bits=bu0, z_bits=bi0, long_rows=self._long_iu)
                nxt = self._exchange_frontier(self.Z, bi1, cap="auto")
            elif how == "gated":
                K.spmm_rowsparse(iu, gu, Y[U:N], alpha=alpha, z=self.G[U:N], beta=c, src_bits=self.bits[2 * l], z_bits=bi0)
                nxt = self._exchange(Y)
            else:
                K.spmm(iu, gu, y=Y[U:N], alpha=alpha, z=self.G[U:N], beta=c)
                nxt = self._exchange(Y)
            if last and sp:
                K.zero_rows_by_bits(self.G[U:N], bi0)          # (G's item rows had their last reader)
            pend.wait()
            if last:
                extra = dict(clear_bits=(self._bits_all,)) if sp else {}
                K.spmm_adam(ui, gi, self.flat[:U], self.m[:U], self.v[:U], self.bc, group["lr"], group["betas"],
                            group["eps"], group["weight_decay"], alpha=alpha, z=self.G[:U], beta=c, clear_z=True, **extra)
            elif how == "list":
                K.spmm_rowlist(ui, gi, Y[:U], self._list_u, self._list_n[0:1], alpha=alpha, z=self.G[:U], beta=c,
                               src_bits=bi0, z_bits=bu0, long_rows=self._long_ui)
            elif how == "gated":
                K.spmm_rowsparse(ui, gi, Y[:U], alpha=alpha, z=self.G[:U], beta=c, src_bits=self.bits[2 * l + 1], z_bits=bu0)
                if l == 1 and L >= 3:
                    K.zero_rows_by_bits(self.Z[:I], bi1)       # (Z had its only reader: all-zero again)
            else:
                K.spmm(ui, gi, y=Y[:U], alpha=alpha, z=self.G[:U], beta=c)
            if sp and l == 0:
                K.zero_rows_by_bits(self.S[U:N], bi0)          # (the seed had its only reader: all-zero again)
            gu, gi = Y[:U], (self.Z[:I] if how == "list" else Y[U:N])
            alpha, pend = 1.0, nxt
        pend.wait()
        K.adam_step(self.flat[U:N], gi, self.m[U:N], self.v[U:N], 0, group["lr"], group["betas"], group["eps"],
                    group["weight_decay"], step_dev=opt._step_dev)
        if not sp:
            self.G[U:N].zero_()
        self._publish(not light)

    def _publish(self, complete):
        """model.result_u / result_i = this step's propagated tables -- or, after a light step, nothing (only the batch's
        rows of them exist; ShardedLightGCN.gene_ranklist fails on None)."""
        self.result_complete = bool(complete)
        U, N = self.U, self.N
        self.model.result_u, self.model.result_i = (self.final[:U], self.final[U:N]) if complete else (None, None)
        self.model._result_cat = None

    def _frontier_bitmaps_and_lists(self):
        """R0's bitmaps (set by the batch / BPR launch) -> item rows united over the ranks; N1 on both sides: bitmaps, this
        rank's work lists for the backward's first layer, N1's item bitmap united over the ranks.  The small collectives go
        FIRST: a process group's collectives run in issue order, behind a 1 GB exchange they would wait for it."""
        K, shard = self.K, self.model.shard
        bu0, bi0, bu1, bi1 = self.bits
        self._union_item_bits(bi0)
        if self.L >= 3 or self.light:
            K.expand_row_bits(shard.ui, bu0, bi1, self._list_i, self._list_n[1:2], bits_self=bi0)
            self._union_item_bits(bi1)
            K.expand_row_bits(shard.iu, bi0, bu1, self._list_u, self._list_n[0:1], bits_self=bu0)

    def __call__(self, users=None, pos=None, neg=None, single=False, full_result=False):
        """One replay = `steps_per_replay` training steps (single=True: exactly one) -> this rank's last batch loss
        (device scalar; the global loss is the mean over ranks).  users / pos / neg (shard-local ids, items as
        item + U_g) only in given_batch mode."""
        if self.edges is None:
            self.ids[0].copy_(users, non_blocking=True)
            torch.sub(pos.to(self.ids[1].device), self.U, out=self.ids[1])
            torch.sub(neg.to(self.ids[2].device), self.U, out=self.ids[2])
        full = bool(full_result) and self.light
        if self.graph is not None:
            (self.graph_full if full else self.graph1 if single else self.graph).replay()
        else:
            self._launch(light=False if full else None)
        self.replays += 1
        self._publish(full or not self.light)
        return self.static_loss

    def run(self, n_steps, full_last=True):
        """n_steps training steps: whole replays first, single-step replays for the remainder; with the light forward the
        last one is a full step (full_last: the evaluation comes next)."""
        k = self.steps_per_replay
        tail = 1 if (self.light and full_last and n_steps > 0) else 0
        n = n_steps - tail
        for _ in range(n // k):
            self()
        for _ in range(n % k):
            self(single=True)
        if tail:
            self(full_result=True)
        if self._compact:                          # (compact frontier exchanges ran: one sync per run() for their overflow flag)
            self.check_frontier()
        return self.static_loss


def build_weak_scaling_job(dataset, world, rank, D, L, reg, device, seed=42, group=None, synthetic=False):
    """bench.py at N GPUs (weak scaling): rank g owns ONE copy of the dataset's users -- global user id g * U1 + u has
    the interactions of user u of the real graph (Data/<dataset>/train.npy, packed in the repository) -- over the
    same I items, so per-rank work stays that of the N=1 configuration and an item's degree is N x its real degree.
    Every rank builds only its own shard (UserShard.from_local); the item degrees come from one all-reduce.
    -> dict(model, local_edges, num_user_local, shard, data)."""
    from . import dataload
    from .synthetic import DATASET_SHAPES, synthetic_interactions
    packed = None if synthetic else dataload.packed_interactions(dataset)
    if packed is not None:
        U1, I, edges1, kind = packed["num_user"], packed["num_item"], np.asarray(packed["train"], dtype=np.int64), "real"
    else:
        U1, I, E1 = DATASET_SHAPES[dataset]
        edges1, kind = synthetic_interactions(U1, I, E1, seed=seed).astype(np.int64), "synthetic"
    U = U1 * world
    mine = np.stack([edges1[:, 0] + rank * U1, edges1[:, 1] - U1 + U], 1)      # global user ids, items as item + U_global
    bounds = [k * U1 for k in range(world + 1)]
    shard = UserShard.from_local(mine, bounds, I, world, rank, device, group=group)
    model = ShardedLightGCN(shard, None, D, reg, L, device, seed=seed, group=group, global_init=False).to(device)
    return dict(model=model, local_edges=shard.local_edges, num_user_local=shard.num_user_local, shard=shard, data=kind,
                U1=U1, I=I)
