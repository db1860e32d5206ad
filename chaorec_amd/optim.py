"""Adam on the fused HIP kernel + whole-step hipGraph capture.

FusedAdam: torch.optim.Adam's update (defaults of main.py:397) as ONE launch per parameter tensor instead of
the ~8 foreach kernels torch issues; the step counter lives on the device so the launch is capturable.

GraphedTrainStep: captures zero_grad -> model.loss -> backward -> optimizer.step into one hipGraph (via
torch.cuda.CUDAGraph: every chaorec kernel is enqueued on the capturing stream) and replays it per batch from
static input buffers: the launch-bound inner loop of train_and_evaluate.py:43-48 without per-kernel host cost.
"""
import os

import torch

from . import ops


def _adjacent_run(first, candidates):
    """The parameters of `candidates` (in order) that continue `first`'s storage without a gap, `first` included."""
    run, end = [], None
    started = False
    for q in candidates:
        if q is first:
            started = True
            run, end = [q], q.data_ptr() + q.numel() * q.element_size()
        elif started and q.data_ptr() == end:
            run.append(q)
            end += q.numel() * q.element_size()
        elif started:
            break
    return run or [first]


# GraphedTrainStep: claimed tables' updates on a side stream beside the rest of the backward (opt-in: FREEDOM's captured step
# 0.496 -> 0.476 ms at best -- the table update fills every CU, so the backward's small launches queue behind its workgroups
# -- and 0.56 / 1.26 ms with other stream priorities: not worth a default)
EARLY_TABLES = os.environ.get("CHAOREC_EARLY_ADAM", "0") == "1"
BIAS_TABLE_STEPS = 1 << 16
MULTI_TENSOR_BELOW = 1 << 20      # parameter runs with fewer elements share one Adam launch (chaorec_adam_multi_f32)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (main.py:397) as one launch per parameter run, step count on the device.

    Feature tables that a model marks `_chaorec_projected_only` (read only through ops.linear_rows: FREEDOM's trainable image
    / text features, Model/FREEDOM.py:59-60, 209-213) are claimed: their [I, K] gradient is never materialised, the
    update comes from chaorec_adam_lowrank_f32 (gy [I, R] and the projection weight instead).  `lazy_rows` (default since
    round 5: on; CHAOREC_LAZY_ADAM=0 / lazy_rows=False: every row every step) additionally defers the zero-gradient updates
    of rows outside the batch until the row is next in a batch -- replayed then operation for operation, so after flush()
    the tables are bit-identical to the eager ones (tests/test_gpu_feature_adam.py: eager and captured steps); between
    flushes rows outside the recent batches are stale IN MEMORY: state_dict() and the training loop's end flush, a caller
    that reads such a table's rows directly calls optimizer.flush() first.  Only tables read through ops.linear_rows alone
    are ever lazy: one that ops.linear reads as a whole (VBPR, MGCN) takes the dense update whatever this says."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lazy_rows=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = None
        self.lazy_rows = (os.environ.get("CHAOREC_LAZY_ADAM", "1") == "1") if lazy_rows is None else bool(lazy_rows)
        self._pending = {}      # claimed parameter -> [gy_full, projection weight] of the backward that just ran
        self._claimed = {}      # id(parameter) -> its group
        self._dense_read = set()  # id(claimed parameter) read as a whole by ops.linear: never updated lazily (submit())
        self._bc_table = None
        # early_tables: the dense updates of claimed tables are launched on a side stream by submit(), step() joins them.
        # Only for loops with exactly one backward() per step() and no gradient exchange in between -- GraphedTrainStep
        # switches it on for itself.
        self._early, self._side, self.early_tables = set(), None, False
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_chaorec_projected_only", False) and p.dim() == 2 and p.shape[1] % 4 == 0 \
                        and p.dtype == torch.float32 and p.is_contiguous():
                    p._chaorec_lowrank_sink = self
                    self._claimed[id(p)] = group

    # -- the sink side of ops.linear_rows' backward -------------------------------------------------------------
    def accepts(self, p):
        return id(p) in self._claimed and p.is_cuda

    def submit(self, p, gy_full, weight, row_token=None, dense_reader=False):
        """dense_reader: the submission comes from ops.linear -- a forward that reads EVERY row of the table (VBPR's v_feat,
        MGCN's image / text tables).  Such a table cannot be updated lazily: its forward has no catch-up, so rows outside
        the batch would be read stale (and ranked stale by gene_ranklist).  It is brought up to date once and from then on
        takes the dense update (mode 0), whatever `lazy_rows` says."""
        if weight.shape[0] > 64:
            raise ValueError("FusedAdam: a claimed feature table needs a projection of at most 64 outputs")
        if dense_reader and id(p) not in self._dense_read:
            self._make_dense(p)
        cur = self._pending.get(p)
        if cur is not None and p in self._early:
            raise RuntimeError("FusedAdam.early_tables: a second backward() reached a claimed table before step() -- its "
                               "update is already running (set early_tables = False for gradient accumulation)")
        if cur is None:
            self._pending[p] = [gy_full, weight, row_token]
            if self.early_tables:
                self._launch_early(p)
        else:
            if cur[1] is not weight:
                raise ValueError("FusedAdam: one claimed feature table, two different projections")
            cur[0] = cur[0] + gy_full            # gradient accumulation over several backward() calls
            cur[2] = None                        # (more than one batch: the step scans gy for the rows instead)

    def _launch_early(self, p):
        """early_tables: the dense update of a claimed table starts the moment its gradient (gy, W) exists -- on a side
        stream, beside the rest of the backward pass (FREEDOM: the 46.6 M-element image table's update is HBM-bound for
        ~235 us while the propagate's backward is a chain of small launches).  Needs the moments and the step counter
        (from the second step on) and the dense mode; step() joins the stream before it advances the counter."""
        st = self.state.get(p)
        if not p.is_cuda or self._step_dev is None or not st or "exp_avg" not in st or (self.lazy_rows and "last" in st):
            return
        gy_full, weight, _ = self._pending[p]
        group = self._claimed[id(p)]
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=p.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side), torch.no_grad():
            # (step offset 1: the device counter still holds the previous step's number)
            ops.adam_lowrank(p.data, gy_full, weight, st["exp_avg"], st["exp_avg_sq"], 1, group["lr"], group["betas"],
                             group["eps"], group["weight_decay"], step_dev=self._step_dev, mode=0)
        self._early.add(p)

    def reduce_pending(self, p, reduce_fn):
        """Multi-process training (dist.ShardedFREEDOM.sync_grads): sum the ranks' gy of a claimed table in place; the
        rows to update are then the union of the ranks' batches, not this rank's list."""
        cur = self._pending.get(p)
        if cur is not None:
            reduce_fn(cur[0])
            cur[2] = None

    @torch.no_grad()
    def _make_dense(self, p):
        self._dense_read.add(id(p))
        st = self.state.get(p)
        if st and "last" in st:          # rows deferred so far: replay them now, then drop the lazy bookkeeping
            group = self._claimed[id(p)]
            ops.adam_lowrank(p.data, None, None, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                             group["eps"], group["weight_decay"], step_dev=self._step_dev, mode=2, last=st["last"],
                             bc_table=self._bc_table)
            for k in ("last", "claim", "stamp", "rowcount", "rowlist", "list_gen"):
                st.pop(k, None)

    def zero_grad(self, set_to_none=True):
        if self._early:          # (an update already running cannot be recalled: early_tables is for backward -> step loops)
            torch.cuda.current_stream().wait_stream(self._side)
            self._early.clear()
        self._pending.clear()
        super().zero_grad(set_to_none=set_to_none)

    def _lowrank_state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        if self.lazy_rows and "last" not in st and id(p) not in self._dense_read:
            strips = ops.adam_lowrank_strips(p.shape[1])
            st["last"] = self._step_dev.to(torch.int32).expand(strips * p.shape[0]).contiguous().view(strips, p.shape[0])
            st["claim"] = torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)   # chaorec_unique_rows' scratch
            st["stamp"] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st["rowcount"] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st["list_gen"] = 0
            if self._bc_table is None:
                self._bc_table = ops.adam_bias_table(BIAS_TABLE_STEPS, self.param_groups[0]["betas"], p.device)
        return st

    @torch.no_grad()
    def catch_up(self, p, rows):
        """Lazy rows: replay the zero-gradient steps the given rows of a claimed table sat out."""
        st = self.state.get(p)
        if not st or "last" not in st:
            return
        group = self._claimed[id(p)]
        # the distinct rows of the batch, listed once: this catch-up and the step's update visit exactly them
        rl = st.get("rowlist")
        if rl is None or rl.numel() < rows.numel():
            rl = st["rowlist"] = torch.zeros(rows.numel(), dtype=torch.int32, device=p.device)
        ops.unique_rows(rows, st["claim"], st["stamp"], rl, st["rowcount"])
        st["list_gen"] += 1
        ops.adam_lowrank(p.data, None, None, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                         group["eps"], group["weight_decay"], step_dev=self._step_dev, mode=3, last=st["last"],
                         bc_table=self._bc_table, rowlist=(rl, st["rowcount"]))
        return st["list_gen"]

    @torch.no_grad()
    def flush(self):
        """Bring every lazily updated row up to the current step (no-op without lazy_rows)."""
        if not self.lazy_rows or self._step_dev is None:
            return
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if id(p) in self._claimed and st and "last" in st:
                    ops.adam_lowrank(p.data, None, None, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                                     group["eps"], group["weight_decay"], step_dev=self._step_dev, mode=2,
                                     last=st["last"], bc_table=self._bc_table)

    def state_dict(self):
        """torch.optim.Adam keeps a `step` per parameter; here the count is one device scalar for the whole optimizer:
        saved as a top-level `chaorec_step` entry and restored by load_state_dict (the per-row `last` stamps of lazy
        rows are only meaningful next to it)."""
        self.flush()
        sd = super().state_dict()
        sd["chaorec_step"] = int(self._step_dev.item()) if self._step_dev is not None else 0
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        step = int(state_dict.pop("chaorec_step", 0))
        super().load_state_dict(state_dict)
        params = [p for g in self.param_groups for p in g["params"]]
        if params:
            if self._step_dev is None:
                self._step_dev = torch.zeros(1, dtype=torch.int32, device=params[0].device)
            self._step_dev.fill_(step)

    def _ensure_state(self, live):
        """Moments for the parameters of `live` that have none yet.  Parameters that sit back to back in one buffer
        (LightGCN's joined embedding tables) get their moments back to back too, so the whole run can be updated by
        ONE launch."""
        if live and self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=live[0].device)
        for p in live:
            st = self.state[p]
            if not st:
                run = [q for q in live if not self.state[q] and q.dtype == p.dtype and q.is_contiguous()]
                run = _adjacent_run(p, run)
                total = sum(q.numel() for q in run)
                flat_m = torch.zeros(total, dtype=p.dtype, device=p.device)
                flat_v = torch.zeros(total, dtype=p.dtype, device=p.device)
                o = 0
                for q in run:
                    self.state[q]["exp_avg"] = flat_m[o:o + q.numel()].view_as(q)
                    self.state[q]["exp_avg_sq"] = flat_v[o:o + q.numel()].view_as(q)
                    o += q.numel()

    @torch.no_grad()
    def make_moments_adjacent(self, params):
        """The Adam moments of `params` (tensors that sit back to back in one buffer) as views of ONE buffer each, in
        that order -- what the fused steps need to update the joined table in a single epilogue.  Moments that exist
        already (an optimizer that trained the tables separately, a loaded state_dict) are MIGRATED: copied into the
        joint buffers and rebound, values unchanged (the fused steps used to raise on such an optimizer)."""
        self._ensure_state(list(params))
        ms, vs = [self.state[p]["exp_avg"] for p in params], [self.state[p]["exp_avg_sq"] for p in params]

        def adjacent(ts):
            end = ts[0].data_ptr()
            for t in ts:
                if t.data_ptr() != end or not t.is_contiguous():
                    return False
                end += t.numel() * t.element_size()
            return True

        if adjacent(ms) and adjacent(vs):
            return
        total = sum(p.numel() for p in params)
        flat_m = torch.empty(total, dtype=params[0].dtype, device=params[0].device)
        flat_v = torch.empty_like(flat_m)
        o = 0
        for p, m, v in zip(params, ms, vs):
            n = p.numel()
            flat_m[o:o + n].copy_(m.reshape(-1))
            flat_v[o:o + n].copy_(v.reshape(-1))
            self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"] = flat_m[o:o + n].view_as(p), flat_v[o:o + n].view_as(p)
            o += n

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            self._ensure_state([p for p in group["params"] if p.grad is not None])
        if self._pending and self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=next(iter(self._pending)).device)
        for p in self._pending:
            self._lowrank_state(p)
        if self._early:          # updates already running on the side stream read the counter and W: join before either moves
            torch.cuda.current_stream().wait_stream(self._side)
        if self._step_dev is not None:
            self._step_dev.add_(1)
        # claimed feature tables first: their update reads the projection weight of THIS step, which the loop below
        # updates
        for p, (gy_full, weight, row_token) in self._pending.items():
            if p in self._early:
                continue
            group, st = self._claimed[id(p)], self.state[p]
            # the row list of this batch's catch-up serves the update too (the gradient is zero in every other row) --
            # unless another forward has re-listed since: then the update finds its rows by scanning gy
            rl = None
            lazy = self.lazy_rows and "last" in st       # (a table ops.linear reads as a whole has no "last": dense update)
            if lazy and row_token is not None and row_token == st.get("list_gen"):
                rl = (st["rowlist"], st["rowcount"])
            ops.adam_lowrank(p.data, gy_full, weight, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                             group["eps"], group["weight_decay"], step_dev=self._step_dev,
                             mode=1 if lazy else 0, last=st.get("last") if lazy else None, bc_table=self._bc_table, rowlist=rl)
        self._pending.clear()
        self._early.clear()
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            small = []
            i = 0
            while i < len(live):
                p = live[i]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                n, j = p.numel(), i + 1
                # extend over parameters whose data, gradient and moments all continue where this one ends
                while j < len(live):
                    q = live[j]
                    gq, sq = q.grad, self.state[q]
                    esz = p.element_size()
                    if not (q.is_contiguous() and gq.is_contiguous() and q.dtype == p.dtype
                            and q.data_ptr() == p.data_ptr() + n * esz and gq.data_ptr() == g.data_ptr() + n * esz
                            and sq["exp_avg"].data_ptr() == st["exp_avg"].data_ptr() + n * esz
                            and sq["exp_avg_sq"].data_ptr() == st["exp_avg_sq"].data_ptr() + n * esz):
                        break
                    n += q.numel()
                    j += 1
                if n >= MULTI_TENSOR_BELOW or p.dtype != torch.float32:
                    ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                                  group["eps"], group["weight_decay"], step_dev=self._step_dev, numel=n)
                else:
                    small.append((p.data, g, st["exp_avg"], st["exp_avg_sq"], n))
                i = j
            # the small tensors of the group (biases, 64 x 64 weights ...): one launch per 48 of them instead of one each
            cap = ops.adam_multi_max() if small else 0
            for k in range(0, len(small), cap or 1):
                chunk = small[k:k + cap]
                if len(chunk) == 1:
                    t = chunk[0]
                    ops.adam_step(t[0], t[1], t[2], t[3], 0, group["lr"], group["betas"], group["eps"],
                                  group["weight_decay"], step_dev=self._step_dev, numel=t[4])
                else:
                    ops.adam_multi(chunk, 0, group["lr"], group["betas"], group["eps"], group["weight_decay"],
                                   step_dev=self._step_dev)
        return loss


class GraphedTrainStep:
    """step(*batch) == { optimizer.zero_grad(); loss = model.loss(*batch); loss.backward(); optimizer.step() },
    replayed from one captured hipGraph.  Batch tensors must keep their shapes (the last, short batch of an
    epoch falls back to the eager path).  Returns the (device) loss of the step."""

    def __init__(self, model, optimizer, example_batch=None, warmup=3, batch_fn=None, loss_fn=None, after_backward=None):
        """`batch_fn` (optional): a capturable callable returning the batch tensors; it is captured INSIDE the
        graph (device-side sampling from a device counter), and the step is then called with no arguments.
        `after_backward` (optional): called between loss.backward() and optimizer.step() -- a sharded model's
        sync_grads() (the all-reduce of the replicated parameters' partial gradients), captured with the rest."""
        self.model, self.optimizer, self.batch_fn = model, optimizer, batch_fn
        self.loss_fn = loss_fn or model.loss
        self.after_backward = after_backward
        if isinstance(optimizer, FusedAdam) and after_backward is None and EARLY_TABLES:
            optimizer.early_tables = True      # this step is exactly zero_grad -> loss -> backward -> step, nothing between
        self._seed = None
        dev = next(model.parameters()).device
        self.static = [b.to(dev).clone() for b in example_batch] if example_batch is not None else None
        self.shapes = [tuple(b.shape) for b in self.static] if self.static is not None else None
        # A model that already trained eagerly keeps its last autograd graph alive through `self.result` (the
        # reference's stale-result quirk), and with it AccumulateGrad nodes bound to the default stream: capturing
        # a backward that reuses them on the capture stream breaks the capture.  Cut that reference first.
        if (not isinstance(getattr(type(model), "result", None), property)
                and torch.is_tensor(getattr(model, "result", None)) and model.result.grad_fn is not None):
            model.result = model.result.detach()     # (no local name may keep the old tensor alive either)
        # The warm-up steps (they build lazily cached state: graph schedules, Adam moments, allocator pools) must
        # not count as training: parameters and optimizer state are put back afterwards.
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        had_state = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                     for p in params if optimizer.state.get(p)}
        saved_step = optimizer._step_dev.clone() if getattr(optimizer, "_step_dev", None) is not None else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager(self.static)
        torch.cuda.current_stream().wait_stream(s)
        with torch.no_grad():
            for p, q in zip(params, saved_p):
                p.copy_(q)
            for p in params:
                for k, v in optimizer.state.get(p, {}).items():
                    if torch.is_tensor(v):                 # in place: the captured graph holds these addresses
                        old = had_state.get(id(p), {}).get(k)
                        if old is None and k == "last" and saved_step is not None:
                            v.copy_(saved_step.expand_as(v))       # (lazy rows: "current for the step count so far")
                        else:
                            v.copy_(old) if old is not None else v.zero_()
            if getattr(optimizer, "_step_dev", None) is not None:
                optimizer._step_dev.copy_(saved_step) if saved_step is not None else optimizer._step_dev.zero_()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        # (a step that holds collectives -- after_backward = a sharded model's sync_grads -- is captured in thread-local
        #  mode: RCCL's watchdog thread polls events while we capture, dist.capture_mode)
        import torch.distributed as _td
        mode = "thread_local" if (_td.is_available() and _td.is_initialized()) else "global"
        if mode == "thread_local":
            from .dist import capture_with_retry, settle_before_capture

            def capture():
                settle_before_capture()          # (RCCL's watchdog must have retired the warm-up's eager collectives)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode=mode):
                    self.static_loss = self._eager(self.static)

            def reset():                         # (a capture that failed has run nothing: only its leftovers go)
                torch.cuda.synchronize()
                self.optimizer.zero_grad(set_to_none=True)

            self.capture_attempts = capture_with_retry(capture, reset, what=type(model).__name__ + " train step")
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.static_loss = self._eager(self.static)
        # the captured forward's output buffer: every replay rewrites it, and model.result must keep pointing at it
        # (an eager step in between -- the short last batch of an epoch -- rebinds model.result to its own tensor)
        plain_attr = not isinstance(getattr(type(model), "result", None), property)   # (a sharded model derives it)
        res = getattr(model, "result", None) if plain_attr else None
        self._captured_result = res if torch.is_tensor(res) else None
        self._graph_generation = getattr(model, "graph_generation", 0)      # the graph arrays this capture holds
        self.replays = 0

    def _eager(self, batch):
        if self.batch_fn is not None:
            batch = self.batch_fn()
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(*batch)
        if self._seed is None or self._seed.shape != loss.shape or self._seed.device != loss.device:
            self._seed = torch.ones_like(loss)          # d(loss)/d(loss), kept: autograd would fill a fresh one per step
        loss.backward(self._seed)
        if self.after_backward is not None:
            self.after_backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, *batch):
        if getattr(self.model, "graph_generation", 0) != self._graph_generation:
            # FREEDOM / LayerGCN re-allocated their pruned graph AFTER this step was captured (another entry count than the
            # one captured): the hipGraph still holds the old arrays' addresses -- replaying it would train on freed
            # memory.  The training loop (train_and_evaluate) captures again; any other holder of a captured step must too.
            # (A rebind BEFORE the capture is harmless: the generation is recorded when the capture ends -- ADVICE r3.)
            raise RuntimeError("GraphedTrainStep: the model rebound its graph after this step was captured "
                               "(model.graph_generation moved): capture a new step")
        if self.batch_fn is not None:
            self.graph.replay()
            self.replays += 1
            if self._captured_result is not None:
                self.model.result = self._captured_result
            return self.static_loss
        if [tuple(b.shape) for b in batch] != self.shapes:
            return self._eager([b.to(self.static[0].device) for b in batch])
        for dst, src in zip(self.static, batch):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        if self._captured_result is not None:
            self.model.result = self._captured_result
        return self.static_loss


class FusedLightGCNStep:
    """One LightGCN training iteration -- train_and_evaluate.py:43-48: zero_grad, model.loss() (Model/LightGCN.py:
    123-135: L propagates + layer mean, BPR + L2), loss.backward(), Adam step -- as 2L + 2 kernel launches (2L + 1 + 1/k in a k-step replay) with no
    autograd tape and no optimizer launch, captured in one hipGraph:

        L x  SpMM (layer mean in the epilogue; the last one writes only the mean = model.result)
        1 x  BPR forward + backward (batch drawn in the launch or given): gradient rows added into G; one idle
             thread moves Adam's step count on and computes the step's bias corrections
        1 x  the loss scalar (single-block fixed-order reduction) + batch / permutation counters + loss bookkeeping
        L-1 x SpMM  g_l = A g_{l+1} + w G
        1 x  SpMM  g_0 = A g_1 + w G with the Adam update of the embedding table in its epilogue

    Same arithmetic, kernel for kernel, as LightGCN.loss_drawn()/loss_local() + backward() + FusedAdam.step(); what
    is gone: the Adam launch and its pass over the gradient (chaorec_spmm_csr_adam_f32), the per-step zero fill of the
    [N, D] batch-gradient buffer G (it is kept all-zero between steps: the last SpMM clears the rows it read),
    the separate BPR backward launch, three counter launches and autograd's bookkeeping.  `steps_per_replay` steps
    are captured back to back in ONE hipGraph: consecutive replays of a graph are ~5 us apart on the device, kernels
    inside one follow each other without a gap.  (The loss reduction stays in line: a forked capture branch for it
    cost ~15 us of cross-queue fork/join on this stack and did not overlap.)  model.result is the
    propagated table of THIS step's forward (the reference's stale-result quirk Q4 is kept).
    The optimizer's state lives in the FusedAdam instance (same tensors, same step counter), so fused and unfused
    steps can be mixed (the short last batch of an epoch runs through the ordinary path)."""

    def __init__(self, model, optimizer, batch_size=1024, edges=None, seed=42, step_dev=None, perm=None, perm_pos=None,
                 given_batch=False, loss_accum=None, capture=True, steps_per_replay=1, light_forward=None):
        """light_forward: None = by size (with the row-sparse backward; CHAOREC_LIGHT_FORWARD=auto/0/1), True / False: the
        forward propagates of a step restricted to the rows its loss reads (see _launch) / every row of every layer."""
        if not isinstance(optimizer, FusedAdam) or len(optimizer.param_groups) != 1:
            raise TypeError("FusedLightGCNStep needs a FusedAdam with one parameter group")
        L = model.n_layers
        if L < 1:
            raise ValueError("FusedLightGCNStep: n_layers >= 1")
        if not model.graph.symmetric:
            raise ValueError("FusedLightGCNStep: the propagate graph must be its own transpose")
        if (edges is None) == (not given_batch):
            raise ValueError("FusedLightGCNStep: either edges (in-launch draw) or given_batch=True")
        self.model, self.optimizer, self.B, self.L = model, optimizer, int(batch_size), L
        self.edges, self.seed, self.step_dev, self.perm, self.perm_pos = edges, seed, step_dev, perm, perm_pos
        group = optimizer.param_groups[0]
        uw, iw = model.user_embedding.weight, model.item_embedding.weight
        if [id(p) for p in group["params"]] != [id(uw), id(iw)]:
            raise ValueError("FusedLightGCNStep: the optimizer must hold exactly the two embedding tables")
        optimizer.make_moments_adjacent([uw, iw])
        st_u, st_i = optimizer.state[uw], optimizer.state[iw]
        flat = model._flat
        N, D = flat.shape
        esz = 4
        if not (uw.data_ptr() == flat.data_ptr() and iw.data_ptr() == flat.data_ptr() + uw.numel() * esz
                and st_i["exp_avg"].data_ptr() == st_u["exp_avg"].data_ptr() + uw.numel() * esz
                and st_i["exp_avg_sq"].data_ptr() == st_u["exp_avg_sq"].data_ptr() + uw.numel() * esz):
            raise ValueError("FusedLightGCNStep: embedding tables / moments are not one contiguous run")
        dev = flat.device
        self.N, self.D = N, D
        self.m = torch.as_strided(st_u["exp_avg"], (N, D), (D, 1))
        self.v = torch.as_strided(st_u["exp_avg_sq"], (N, D), (D, 1))
        self.fbuf = [torch.empty((N, D), dtype=torch.float32, device=dev) for _ in range(max(L - 1, 0))]   # x_1 .. x_{L-1}
        self.buf = (self.fbuf + [torch.empty((N, D), dtype=torch.float32, device=dev) for _ in range(2)])[:2]   # g_l
        self.final = torch.empty((N, D), dtype=torch.float32, device=dev)
        self.G = torch.zeros((N, D), dtype=torch.float32, device=dev)       # all-zero between steps
        self.ids = tuple(torch.zeros(self.B, dtype=torch.int64, device=dev) for _ in range(3))
        self.coef = torch.empty(self.B, dtype=torch.float32, device=dev)
        self.ws = torch.empty(4 * self.B, dtype=torch.float32, device=dev)
        self.ws_steps = None                # [steps_per_replay, 4 B]: one workspace per step of a multi-step replay
        self.out = torch.zeros(3, dtype=torch.float32, device=dev)
        self.static_loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.bc = torch.ones(2, dtype=torch.float32, device=dev)
        self.loss_accum = loss_accum
        # Row-sparse backward.  The batch gradient G has 3 B non-zero rows R0 out of N; g_{L-1} = w (A G) + w G is non-zero in
        # N1 = R0 + nbr(R0) only.  With a real BPR batch (positives drawn in proportion to their popularity) N1 is ~13-26 % of a
        # BASELINE-configs[4]-sized graph and N2 = nbr(N1) nearly all of it, so:
        #   launch 1  runs over the LIST of N1's rows (ops.expand_row_bits emits it from R0's few rows, ops.spmm_rowlist_raw
        #             computes exactly those rows and leaves the others untouched): 33.8 -> 8.7 ms at 12 M rows;
        #   launch 2  is the ordinary launch with its GATHERS gated by N1's bitmap (ops.spmm_rowsparse_raw: an unflagged source
        #             row is not fetched -- which is also why launch 1 need not write it): 33.1 -> 27.3 ms;
        #   launch 3  (Adam epilogue) is dense, and clears the bitmaps and the list's length as a side job.
        # Same sums bit for bit (the skipped terms are val * (+0)).  bits[0] = R0 (set by the BPR launch), bits[1] = N1.  It pays
        # where the frontier is a part of the graph and costs launches where it is not (sports: N1 is half the graph, N2 all of
        # it), hence by size.  CHAOREC_SPARSE_BACKWARD=0 / 1: off / forced.
        wide_ok, self.sparse_bwd, light_by_size = self.frontier_modes(N, L, D)
        # Light forward.  train_and_evaluate.py:43-48 reads the propagated table (Model/LightGCN.py:95's mean) in the batch's
        # rows R0 only; the other rows of `model.result` are a by-product that nothing looks at before the next evaluation.
        # A LIGHT step therefore draws its batch first (ops.batch_rows: the same triples, flagged and listed), expands R0 to
        # N1 once (the backward's launch 1 needs the same list) and runs
        #   layers 1 .. L-2  dense                       (x_{L-2} is gathered by N1's rows: nearly every row has a reader)
        #   layer  L-1       over N1's row list          (x_{L-1} is gathered by R0's rows only)
        #   layer  L         over R0's row list, with the layer mean of those rows in its epilogue
        # -- the same arithmetic for every row it computes, so loss, gradient and updated tables are the full step's bit for
        # bit.  The step BEFORE AN EVALUATION must be a full one (`full_result=True`; run() does it): gene_ranklist reads the
        # whole table of the last training forward (the reference's stale-result quirk Q4).
        if light_forward is None:
            light_forward = light_by_size
        elif light_forward and not (wide_ok and L <= 4):
            raise ValueError("FusedLightGCNStep: the light forward needs 2 <= n_layers <= 4 and D in {64, 128, 256}")
        self.light = bool(light_forward)
        if self.light and not self.sparse_bwd:
            self.sparse_bwd = True              # (its bitmaps and N1's list are the forward's as well)
        self.bits = None
        if self.sparse_bwd:
            words = (N + 31) // 32 + 1
            self._bits_all = torch.zeros(2 * words + 2, dtype=torch.int32, device=dev)      # (+ the two row lists' lengths)
            self.bits = [self._bits_all[k * words:(k + 1) * words] for k in range(2)]
            self._list_n = self._bits_all[2 * words:2 * words + 1]
            self._list0_n = self._bits_all[2 * words + 1:]
            self._row_list = torch.empty(N, dtype=torch.int32, device=dev)                  # N1's rows, in no particular order
            self._list0 = torch.empty(3 * self.B, dtype=torch.int32, device=dev)            # R0's rows
            self._long = ops.long_row_buffers(model.graph)                                  # (the list launches' long rows)
        self.result_complete = True
        self.steps_per_replay = int(steps_per_replay) if (capture and edges is not None) else 1
        self.replays = 0
        self.graph = self.graph1 = self.graph_full = None
        if capture:
            # schedules, lazily built by the first SpMM call, must exist before capture
            model.graph.schedule(D)
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            saved = self._save_state()
            with torch.cuda.stream(s):
                self._launch()
                if self.light:
                    self._launch(light=False)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._restore_state(saved)
            self.graph1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph1):
                self._launch()
            if self.light:
                self.graph_full = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_full):
                    self._launch(light=False)
            self.graph = self.graph1
            if self.steps_per_replay > 1:
                # k steps per replay: the loss bookkeeping (reduction of the per-sample terms, epoch loss, batch counter
                # and permutation cursor) runs ONCE, after the last step -- nothing inside a step reads it, only the next
                # step's batch draw does, and that takes its position as cursor + j B with j fixed per captured launch
                k = self.steps_per_replay
                self.ws_steps = torch.empty((k, 4 * self.B), dtype=torch.float32, device=dev)
                self._fin_scratch = torch.zeros(2 * k + 1, dtype=torch.float32, device=dev)    # (+ the launch's ticket)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    for j in range(k):
                        self._launch(j, k)
            # the first launch of a captured graph uploads it (tens of us for the k-step one): do it here, on a copy
            # of the state, so that the first replay a caller times is like every other
            saved = self._save_state()
            for gph in {id(g_): g_ for g_ in (self.graph1, self.graph, self.graph_full) if g_ is not None}.values():
                gph.replay()
            torch.cuda.synchronize()
            self._restore_state(saved)

    def _counters(self):
        return [t for t in (self.step_dev, self.perm_pos, self.loss_accum, self.optimizer._step_dev) if t is not None]

    def _save_state(self):
        return ([self.model._flat.clone(), self.m.clone(), self.v.clone()], [t.clone() for t in self._counters()])

    def _restore_state(self, saved):
        with torch.no_grad():
            for dst, src in zip((self.model._flat, self.m, self.v), saved[0]):
                dst.copy_(src)
            for dst, src in zip(self._counters(), saved[1]):
                dst.copy_(src)
            self.G.zero_()
            if self.bits is not None:
                self._bits_all.zero_()

    @staticmethod
    def frontier_modes(n_rows, n_layers, D):
        """-> (the shapes the frontier launches are built for, row-sparse backward by size, light forward by size) for a
        graph of n_rows rows: what __init__ decides when it is not told (CHAOREC_SPARSE_BACKWARD / CHAOREC_LIGHT_FORWARD =
        auto / 0 / 1; CHAOREC_SPARSE_BACKWARD_MIN_ROWS).  Also for callers that plan around a light step (the training
        loop picks its replay length so that a light epoch's steps fill whole replays)."""
        mode = os.environ.get("CHAOREC_SPARSE_BACKWARD", "auto")
        wide_ok = n_layers >= 2 and D % 4 == 0 and (D // 4) in (16, 32, 64)
        sparse_bwd = wide_ok and mode != "0" and (mode == "1" or n_rows >= int(os.environ.get("CHAOREC_SPARSE_BACKWARD_MIN_ROWS",
                                                                                              "400000")))
        light = sparse_bwd and n_layers <= 4 and os.environ.get("CHAOREC_LIGHT_FORWARD", "auto") != "0"
        return wide_ok, sparse_bwd, light

    @torch.no_grad()
    def _launch(self, j=0, k=1, light=None):
        """Step j of a k-step replay (k = 1: a step with its own finalize launch).  light: None = as built."""
        model, opt, L, B, D = self.model, self.optimizer, self.L, self.B, self.D
        group = opt.param_groups[0]
        csr, x0, w = model.graph, model._flat, 1.0 / (L + 1)
        light = self.light if light is None else bool(light)
        draw = self.edges is not None
        ws = self.ws if k == 1 else self.ws_steps[j]
        if light:
            ops.batch_rows(self.ids, self.bits[0], model.num_user, self._list0, self._list0_n, edges=self.edges,
                           hist=model.hist if draw else None, num_user=model.num_user, num_item=model.num_item, seed=self.seed,
                           step=j, step_dev=self.step_dev, perm=self.perm, perm_pos=self.perm_pos, pos_offset=j * B)
            ops.expand_row_bits(csr, self.bits[0], self.bits[1], self._row_list, self._list_n)
            xs = [x0]
            for l in range(L - 2):
                xs.append(ops.spmm_raw(csr, xs[-1], y=self.fbuf[l]))
            ops.spmm_rowlist_raw(csr, xs[-1], self.fbuf[L - 2], self._row_list, self._list_n, long_rows=self._long)
            xs.append(self.fbuf[L - 2])
            ops.spmm_rowlist_raw(csr, xs[-1], None, self._list0, self._list0_n, mean_out=self.final, mean_terms=xs, mean_w=w,
                                 long_rows=self._long)
            ops.bpr_fwd_bwd(self.final, model.num_user, self.G, B, ops.VARIANT_LOG_SIGMOID_EPS, model.reg_weight, self.coef,
                            ws, self.ids, num_user=model.num_user, num_item=model.num_item, adam_step=opt._step_dev,
                            betas=group["betas"], adam_bc=self.bc)
        else:
            ops.forward_layers(csr, x0, L, self.final, self.fbuf)
            ops.bpr_fwd_bwd(self.final, model.num_user, self.G, B, ops.VARIANT_LOG_SIGMOID_EPS, model.reg_weight, self.coef,
                            ws, self.ids, edges=self.edges, hist=model.hist if draw else None, num_user=model.num_user,
                            num_item=model.num_item, seed=self.seed, step=j, step_dev=self.step_dev, perm=self.perm,
                            perm_pos=self.perm_pos, adam_step=opt._step_dev, betas=group["betas"], adam_bc=self.bc,
                            pos_offset=j * B, row_bits=self.bits[0] if self.sparse_bwd else None)
        book = dict(out_total=self.static_loss, loss_accum=self.loss_accum, advance=self.step_dev if draw else None,
                    perm_pos=self.perm_pos if (draw and self.perm is not None) else None)
        if k == 1:
            ops.bpr_finalize(ws, B, D, model.reg_weight, self.out, **book)
        elif j == k - 1:
            ops.bpr_finalize_steps(self.ws_steps, k, B, D, model.reg_weight, self.out, scratch=self._fin_scratch, **book)
        g, alpha = self.G, w                    # g_{L-1} = w (A G) + w G, then g_l = A g_{l+1} + w G
        for l in range(L - 1):
            y = self.buf[l & 1]
            if self.sparse_bwd and l == 0 and L >= 3:
                # g = G (rows R0): the output's rows N1 as a list; rows outside it stay unwritten (the next launch, gated by
                # N1's bitmap, never gathers them)
                if not light:                   # (a light step expanded R0 before its forward)
                    ops.expand_row_bits(csr, self.bits[0], self.bits[1], self._row_list, self._list_n)
                ops.spmm_rowlist_raw(csr, g, y, self._row_list, self._list_n, alpha=alpha, z=self.G, beta=w,
                                     src_bits=self.bits[0], z_bits=self.bits[0], long_rows=self._long)
            elif self.sparse_bwd and l < 2:
                # every row computed and written (its reader is dense), gathers gated by the source's bitmap
                ops.spmm_rowsparse_raw(csr, g, y, alpha=alpha, z=self.G, beta=w, src_bits=self.bits[l], z_bits=self.bits[0])
            else:
                ops.spmm_raw(csr, g, y=y, alpha=alpha, z=self.G, beta=w)
            g, alpha = y, 1.0
        ops.spmm_adam_raw(csr, g, x0, self.m, self.v, self.bc, group["lr"], group["betas"], group["eps"],
                          group["weight_decay"], alpha=alpha, z=self.G, beta=w, clear_z=L >= 2,
                          clear_bits=(self._bits_all,) if self.sparse_bwd else ())
        if L < 2:
            self.G.zero_()                      # (the single backward SpMM gathers from G: it cannot clear it)
        self._publish(not light)

    def _publish(self, complete):
        """model.result = this step's propagated table -- or, after a light step, nothing: only the batch's rows of it exist,
        and a reader of the rest must fail (Model.LightGCN.gene_ranklist says how to get the table)."""
        self.result_complete = bool(complete)
        self.model.result = self.final if complete else None

    def __call__(self, users=None, pos=None, neg=None, single=False, full_result=False):
        """One replay = `steps_per_replay` training steps (single=True: exactly one, whatever the replay size) -> the
        LAST step's loss (device scalar, rewritten by the next call; per-step sums go to loss_accum).  users / pos / neg
        (GLOBAL item ids, the reference's batch format) only in given_batch mode.  full_result=True (a step built with the
        light forward): ONE step that leaves the whole propagated table in model.result -- the step before an evaluation."""
        if self.edges is None:
            self.ids[0].copy_(users, non_blocking=True)
            torch.sub(pos.to(self.ids[1].device), self.model.num_user, out=self.ids[1])
            torch.sub(neg.to(self.ids[2].device), self.model.num_user, out=self.ids[2])
        full = bool(full_result) and self.light
        if self.graph is not None:
            (self.graph_full if full else self.graph1 if single else self.graph).replay()
        else:
            self._launch(light=False if full else None)
        self.replays += 1
        self._publish(full or not self.light)
        return self.static_loss

    def run(self, n_steps, full_last=True):
        """n_steps training steps: whole replays first, single-step replays for the remainder.  full_last (light forward
        only): the last of them leaves the whole propagated table behind -- an epoch's steps, with the evaluation next."""
        k = self.steps_per_replay
        tail = 1 if (self.light and full_last and n_steps > 0) else 0
        n = n_steps - tail
        for _ in range(n // k):
            self()
        for _ in range(n % k):
            self(single=True)
        if tail:
            self(full_result=True)
        return self.static_loss
