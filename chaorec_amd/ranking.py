"""The one `gene_ranklist` behind the 49 near-identical copies in the reference's Model/*.py:
score = user @ item.T, history -> mask_value (1e-6 in 46 models, 1e-5 in 3), top-k, + num_user, LongTensor on CPU.

    from chaorec_amd.ranking import gene_ranklist, history_csr
    self.hist = history_csr(user_item_dict, num_user, device)          # once, in __init__
    return gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk)
"""
import torch

from . import graph, ops


def history_csr(user_item_dict, num_user, device):
    rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
    return rowptr.to(device), col.to(device)


def _to_host(idx):
    """[U, K] int64 rank list -> CPU LongTensor through page-locked memory (torch's caching host allocator hands the
    same pages out again call after call; a pageable .cpu() copy of sports' 11.6 MB list costs several times the
    ranking itself)."""
    buf = torch.empty(idx.shape, dtype=idx.dtype, pin_memory=True)
    buf.copy_(idx, non_blocking=True)
    torch.cuda.current_stream(idx.device).synchronize()
    return buf


class RankState:
    """What one model's evaluations carry from call to call: the per-user candidate thresholds of
    chaorec_score_topk_hinted_f32 (each gene_ranklist() leaves, per user, the exact score of rank 2.2 K for the next
    one: one epoch of training moves the scores little, so the next call needs no sampling pass and re-scores about
    half the candidates).  The thresholds never change a result; stale ones only cost a retry."""

    LIGHT_BELOW = 16     # a call whose predecessor queued at most this many users for the retry pass runs without one
    STALE_SHARE = 0.1    # ... and one whose predecessor queued more than this share of the users runs WITHOUT hints

    def __init__(self):
        self.hint, self.valid = None, False
        self.counters = self.counters_host = None
        self.copied = None          # event: the previous call's counters have reached the host
        self.cooldown = self.backoff = 0

    def buffer(self, num_user, device):
        if self.hint is None or self.hint.numel() != num_user or self.hint.device != device:
            self.hint, self.valid = torch.empty(num_user, dtype=torch.float32, device=device), False
            # the call's queue lengths land in PAGE-LOCKED HOST memory, written by the call's last launch itself (pinned
            # memory is mapped into the device's address space, like the rank list's zero-copy output): a device buffer +
            # an asynchronous D2H copy was one more launch on the call's stream (a 7 us copy kernel: 4 % of the steady call)
            self.counters = self.counters_host = torch.zeros(4, dtype=torch.int32).pin_memory()
            self.copied = None
        return self.hint

    def light(self):
        """No retry pass this time?  Yes when the previous call's queue lengths are on the host already (they were
        copied asynchronously: no sync here) and that call queued only a handful of users."""
        return self.prev_queue is not None and self.prev_queue <= self.LIGHT_BELOW

    def use_hints(self, num_user):
        """Carried thresholds pay off when the tables moved little since the last call.  Early in training they move a
        lot -- every user fails pass A, which then only costs time (a full sweep + selection before the retry pass).
        The previous call's queue length tells: after a call that queued more than STALE_SHARE of the users the next
        calls run without hints, 1, 2, 4 ... up to 16 of them, until a hinted call succeeds again."""
        if not self.valid:
            return False
        self.prev_queue = None                      # the previous call's pass-A queue, if it had a pass A and we know it
        if self.copied is not None and self.copied.query() and self.last_hinted:
            self.prev_queue = int(self.counters_host[0])
            if self.prev_queue > self.STALE_SHARE * num_user:
                self.backoff = min(max(2 * self.backoff, 1), 16)
                self.cooldown = self.backoff
            else:
                self.backoff = 0
        self.last_hinted = False
        if self.cooldown > 0:
            self.cooldown -= 1
            return False
        return True

    last_hinted = False
    last_light = False       # (what the last call ran as: read by bench.py for its line, never by the product)
    prev_queue = None

    def after_call(self, hinted, light=False):
        self.copied = torch.cuda.Event()            # (once it has passed, the call's counters are in counters_host)
        self.copied.record()
        self.valid = True
        self.last_hinted = bool(hinted)
        self.last_light = bool(light)


def hint_rank_for(topk):
    """The rank whose exact score becomes the next call's threshold: 2.2 K.  At 2 K a handful of sports' 28 940 users
    (0-5 per epoch) drop below K candidates one epoch later and take the exact route, which costs the call 23 us
    however few they are; at 2.2 K that queue is empty in most epochs, for 11 % more candidates (tools/score_profile.py
    with HINT_RANKS)."""
    return min(int(2.2 * topk), 128)


def state_of(model):
    """The RankState of `model` (created on first use; kept out of state_dict / parameters)."""
    st = model.__dict__.get("_rank_state")
    if st is None:
        st = model.__dict__["_rank_state"] = RankState()
    return st


ZERO_COPY = True     # to_cpu: the selection writes the rank list into pinned host memory itself (no D2H copy afterwards)


def gene_ranklist(result, num_user, num_item, hist, mask_value=1e-6, topk=50, to_cpu=True, state=None, id_offset=None,
                  items=None):
    """result [N, D] (users first) on the GPU -> LongTensor [num_user, topk] of GLOBAL item ids on the CPU (the
    reference's contract); to_cpu=False keeps it in HBM for utils.gene_metrics_device.
    id_offset (default num_user): what is added to an item's row number -- a user shard ranks its own users against the
    replicated items and reports item + num_user_GLOBAL.  items: the item table where it is not part of `result`."""
    if items is not None and result.shape[1] not in (8, 16, 32, 64, 128) and not (result.shape[1] > 128 and result.shape[1] % 64 == 0):
        result, items = torch.cat((result[:num_user].detach(), items.detach()), 0), None     # (odd width: padded below)
    id_offset = num_user if id_offset is None else int(id_offset)
    host = None
    if to_cpu and ZERO_COPY:
        host = torch.empty((num_user, topk), dtype=torch.int64, pin_memory=True)
    with torch.no_grad():
        result = result.detach()
        D = result.shape[1]
        if D not in (8, 16, 32, 64, 128) and not (D > 128 and D % 64 == 0):
            # a width the scoring kernels do not tile (VBPR with a 16-wide id embedding: 80): zero columns up to the next
            # one that they do -- a dot product's k-ascending chain only gains +0 * 0 terms, the scores keep their bits
            wide = next(w for w in (8, 16, 32, 64, 128) if w >= D) if D < 128 else (D + 63) // 64 * 64
            result = torch.cat((result, result.new_zeros(result.shape[0], wide - D)), 1)
        ue = result[:num_user]
        ie = items.detach() if items is not None else result[num_user:num_user + num_item]
        if state is not None:
            hint = state.buffer(num_user, result.device)
            hinted = state.use_hints(num_user)          # (once per call: it consumes the previous call's counters)
            light = hinted and state.light()
            idx, _ = ops.score_topk(ue, ie, hist, mask_value, topk, id_offset=id_offset, hint=hint, hint_valid=hinted,
                                    hint_rank=hint_rank_for(topk), light=light, counters=state.counters, idx_out=host)
            state.after_call(hinted, light)
        else:
            idx, _ = ops.score_topk(ue, ie, hist, mask_value, topk, id_offset=id_offset, idx_out=host)
    if host is not None:
        torch.cuda.current_stream(result.device).synchronize()
        return host
    return _to_host(idx) if to_cpu else idx
