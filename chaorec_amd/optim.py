"""Adam on the fused HIP kernel + whole-step hipGraph capture.

FusedAdam: torch.optim.Adam's update (defaults of main.py:397) as ONE launch per parameter tensor instead of
the ~8 foreach kernels torch issues; the step counter lives on the device so the launch is capturable.

GraphedTrainStep: captures zero_grad -> model.loss -> backward -> optimizer.step into one hipGraph (via
torch.cuda.CUDAGraph: every chaorec kernel is enqueued on the capturing stream) and replays it per batch from
static input buffers: the launch-bound inner loop of train_and_evaluate.py:43-48 without per-kernel host cost.
"""
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


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if self._step_dev is None:
                    self._step_dev = torch.zeros(1, dtype=torch.int32, device=p.device)
                break
        if self._step_dev is not None:
            self._step_dev.add_(1)
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            for p in live:
                st = self.state[p]
                if not st:
                    # parameters that sit back to back in one buffer (LightGCN's joined embedding tables) get their
                    # moments back to back too, so the whole run can be updated by ONE launch
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
                ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], 0, group["lr"], group["betas"],
                              group["eps"], group["weight_decay"], step_dev=self._step_dev, numel=n)
                i = j
        return loss


class GraphedTrainStep:
    """step(*batch) == { optimizer.zero_grad(); loss = model.loss(*batch); loss.backward(); optimizer.step() },
    replayed from one captured hipGraph.  Batch tensors must keep their shapes (the last, short batch of an
    epoch falls back to the eager path).  Returns the (device) loss of the step."""

    def __init__(self, model, optimizer, example_batch=None, warmup=3, batch_fn=None, loss_fn=None):
        """`batch_fn` (optional): a capturable callable returning the batch tensors; it is captured INSIDE the
        graph (device-side sampling from a device counter), and the step is then called with no arguments."""
        self.model, self.optimizer, self.batch_fn = model, optimizer, batch_fn
        self.loss_fn = loss_fn or model.loss
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
                        v.copy_(old) if old is not None else v.zero_()
            if getattr(optimizer, "_step_dev", None) is not None:
                optimizer._step_dev.copy_(saved_step) if saved_step is not None else optimizer._step_dev.zero_()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager(self.static)
        # the captured forward's output buffer: every replay rewrites it, and model.result must keep pointing at it
        # (an eager step in between -- the short last batch of an epoch -- rebinds model.result to its own tensor)
        plain_attr = not isinstance(getattr(type(model), "result", None), property)   # (a sharded model derives it)
        res = getattr(model, "result", None) if plain_attr else None
        self._captured_result = res if torch.is_tensor(res) else None
        self.replays = 0

    def _eager(self, batch):
        if self.batch_fn is not None:
            batch = self.batch_fn()
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(*batch)
        if self._seed is None or self._seed.shape != loss.shape or self._seed.device != loss.device:
            self._seed = torch.ones_like(loss)          # d(loss)/d(loss), kept: autograd would fill a fresh one per step
        loss.backward(self._seed)
        self.optimizer.step()
        return loss.detach()

    def __call__(self, *batch):
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
