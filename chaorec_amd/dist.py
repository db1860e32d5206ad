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

from . import _lib, graph, ops


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
        """From the WHOLE edge list (every rank holds it: small graphs, tests).  self_loops: BasicGCN's
        D^-1/2 (A + I) D^-1/2 (BasicGCN.py:37-46): degrees count the loop, the loop's own weight 1/(d+1) is kept as the
        diagonals `diag_u` (local users) / `diag_i` (items)."""
        e = np.asarray(edges, dtype=np.int64)
        u, i = e[:, 0], e[:, 1] - num_user
        bounds = partition_users_by_nnz(np.bincount(u, minlength=num_user), world)
        sel = (u >= bounds[rank]) & (u < bounds[rank + 1])
        deg_i = np.bincount(i, minlength=num_item)
        self._build(u[sel], i[sel], deg_i, bounds, num_user, num_item, world, rank, device, self_loops)

    @classmethod
    def from_local(cls, local_edges, bounds, num_item, world, rank, device, group=None, self_loops=False):
        """Per-rank construction: `local_edges` [E_g, 2] holds ONLY this rank's users -- GLOBAL user ids
        in [bounds[rank], bounds[rank+1]), item ids as item + num_user_global -- so no rank ever generates, loads or
        sorts the whole graph.  The one global quantity the normalisation needs, the item degrees, is an all-reduce
        of the ranks' local counts ([I] int64: 16 MB at 2 M items)."""
        self = cls.__new__(cls)
        num_user = int(bounds[-1])
        e = np.asarray(local_edges, dtype=np.int64)
        u, i = e[:, 0], e[:, 1] - num_user
        if len(u) and (u.min() < bounds[rank] or u.max() >= bounds[rank + 1]):
            raise ValueError("from_local: an edge of a user this rank does not own")
        deg_i = torch.from_numpy(np.bincount(i, minlength=num_item).astype(np.int64))
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            backend = dist.get_backend(group)
            t = deg_i.to(device) if backend == "nccl" else deg_i
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            deg_i = t.cpu()
        self._build(u, i, deg_i.numpy(), list(bounds), num_user, num_item, world, rank, device, self_loops)
        return self

    def _build(self, u, il, deg_i, bounds, num_user, num_item, world, rank, device, self_loops):
        self.bounds = [int(b) for b in bounds]
        self.u0, self.u1 = self.bounds[rank], self.bounds[rank + 1]
        self.num_user_global, self.num_item, self.world, self.rank = num_user, num_item, world, rank
        n_local = self.u1 - self.u0
        ul = u - self.u0
        loop = 1 if self_loops else 0
        deg_u = np.bincount(ul, minlength=n_local) + loop            # this rank's users only: all their edges are here
        deg_i = np.asarray(deg_i) + loop
        # same fp32 normalisation as graph.lightgcn_csr, with GLOBAL degrees
        dis_u = torch.from_numpy(deg_u.astype(np.float32)).pow(-0.5)
        dis_i = torch.from_numpy(deg_i.astype(np.float32)).pow(-0.5)
        w = dis_u[torch.from_numpy(ul)] * dis_i[torch.from_numpy(il)]
        self.ui = graph.coo_to_csr(ul, il, w, n_local, num_item).to(device)        # y_u = B_g x_i
        self.iu = graph.coo_to_csr(il, ul, w, num_item, n_local).to(device)        # p_i = B_g^T x_u
        self.ui._t, self.iu._t = self.iu, self.ui
        self.local_edges = np.stack([ul, il + n_local], 1).astype(np.int32)        # shard-local id convention
        self.num_user_local = n_local
        self.nnz = int(len(ul))
        self.diag_u = self.diag_i = None
        if self_loops:
            self.diag_u = (dis_u * dis_u).view(-1, 1).to(device)
            self.diag_i = (dis_i * dis_i).view(-1, 1).to(device)


def joined_shard_csr(shard):
    """The rank's blocks as ONE symmetric graph over its own joined table [local users; items]: rows 0 .. U_g-1 are
    B_g's rows (columns shifted by U_g), rows U_g .. are B_g^T's.  One SpMM launch over it computes a layer's user rows
    (complete) and the rank's partial of the item rows -- the launch count of the unsharded step."""
    if getattr(shard, "_joined", None) is None:
        ui, iu = shard.ui, shard.iu
        U = shard.num_user_local
        rowptr = torch.cat([ui.rowptr, iu.rowptr[1:] + ui.nnz])
        col = torch.cat([ui.col + U, iu.col])
        val = torch.cat([ui.val, iu.val])
        shard._joined = graph.CSR(rowptr, col, val, U + shard.num_item, U + shard.num_item, symmetric=True)
    return shard._joined


# CHAOREC_FORCE_COLLECTIVES=1: issue the exchanges on a 1-rank group too, so that a 1-GPU box exercises the
# RCCL launch (and hipGraph capture of it) that the N>1 job uses
import os as _os
_FORCE_COLLECTIVES = _os.environ.get("CHAOREC_FORCE_COLLECTIVES", "0") == "1"

# How the ranks' [I, D] item partials are summed (SURVEY 8(e) "Collective choice"; CHAOREC_DIST_EXCHANGE):
#   allreduce  one all-reduce (RCCL picks ring / tree: a ring is bound by ONE xGMI link, 2 * 7/8 * bytes / 153 GB/s)
#   rs_ag      reduce-scatter + all-gather of 1/world row blocks (RCCL's own algorithms for each half)
#   direct     the reduce-scatter as ONE all-to-all -- every rank sends block j straight to rank j, all 7 links at
#              once, and sums the 8 blocks it receives itself -- then the all-gather: 2 * bytes / 8 / 153 GB/s per
#              half on a fully connected xGMI node (1.7 ms instead of 11.7 ms for config 5's 1.02 GB)
# Same sums up to fp32 association.  No multi-GPU node was available to time them against each other.
EXCHANGE_MODES = ("auto", "allreduce", "rs_ag", "direct", "p2p")
DIRECT_FELL_BACK = False       # set when a captured step replaced `direct` by `rs_ag` (see _sum_exchange_async)
DIRECT_WENT_P2P = False        # set when a captured step ran `direct` as the hand-written peer-to-peer exchange instead
# `auto` (the default) picks by buffer size: below AUTO_BIG_BYTES one RCCL all-reduce (launch-latency-bound: one launch);
# from there on whatever calibrate_exchange() measured fastest among the modes that passed its first-contact check
# on this node -- and RCCL's all-reduce when nobody calibrated (the conservative choice: the hand-written p2p exchange
# has to prove itself against an all-reduce on the node it runs on before a step trusts it).
AUTO_BIG_BYTES = int(_os.environ.get("CHAOREC_DIST_BIG_BYTES", str(64 << 20)))
_AUTO_BIG_CHOICE = {}          # numel -> mode, filled by calibrate_exchange()
_VETOED = set()                # modes that failed a first-contact check in this process (or were vetoed by the launcher)
MODES_USED = set()             # what the exchanges of this process actually ran (bench line)
CALIBRATION = {}               # numel -> the table calibrate_exchange() measured (bench line)
_FORCED = [None]               # calibrate_exchange() runs one named mode at a time through the product path
STATS = {"exchanges": 0, "bytes": 0}     # every buffer this process handed to a collective (host-side count, bench line)


def _count(t):
    STATS["exchanges"] += 1
    STATS["bytes"] += t.numel() * t.element_size()


def exchange_mode():
    m = _os.environ.get("CHAOREC_DIST_EXCHANGE", "auto")
    if m not in EXCHANGE_MODES:
        raise ValueError(f"CHAOREC_DIST_EXCHANGE={m}: one of {EXCHANGE_MODES}")
    return m


def veto(mode):
    _VETOED.add(mode)
    for k in [k for k, v in _AUTO_BIG_CHOICE.items() if v == mode]:
        del _AUTO_BIG_CHOICE[k]


for _m in _os.environ.get("CHAOREC_DIST_VETO", "").split(","):
    if _m:
        _VETOED.add(_m)


def resolve_mode(buf):
    """The exchange mode for THIS buffer: the mode asked for, `auto` resolved by size, vetoed modes replaced by rs_ag."""
    m = _FORCED[0] or exchange_mode()
    if m == "auto":
        big = buf.numel() * buf.element_size() >= AUTO_BIG_BYTES
        m = _AUTO_BIG_CHOICE.get(buf.numel(), "allreduce") if big else "allreduce"
    if m in _VETOED:
        m = "rs_ag" if "rs_ag" not in _VETOED else "allreduce"
    return m


def exchange_mode_used():
    """What the bench line reports: the mode asked for, what `auto` resolved to, and what captured steps ran instead
    where that differs."""
    m = exchange_mode()
    if m == "auto" or _VETOED:
        m += " -> " + "/".join(sorted(MODES_USED) or ["(no exchange ran)"])
        if _VETOED:
            m += f" (failed their first-contact check here: {sorted(_VETOED)})"
    if DIRECT_WENT_P2P:
        return m + " (captured steps: p2p -- RCCL's all-to-all is not capturable on this stack; the same direct pattern as " \
                   "pull kernels over IPC-mapped peer buffers)"
    return m + (" (captured steps: rs_ag -- RCCL's all-to-all is not capturable on this stack)" if DIRECT_FELL_BACK else "")


def _active(group):
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES)


def padded_rows(n_rows, group=None):
    """Rows of an exchange buffer: a multiple of the world size, so that it splits into equal row blocks."""
    w = dist.get_world_size(group) if dist.is_initialized() else 1
    return (n_rows + w - 1) // w * w


def exchange_buffer(n_rows, D, like, group=None):
    """-> (padded [rows_pad, D] buffer, its [n_rows, D] view).  The pad rows take part in the sums and are never read."""
    buf = torch.empty((padded_rows(n_rows, group), D), dtype=like.dtype, device=like.device)
    if buf.shape[0] > n_rows:
        buf[n_rows:].zero_()
    return buf, buf[:n_rows]


def _all_reduce(t, group):
    if _active(group):
        _count(t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _Pending:
    """Handle of an exchange in flight (RCCL runs it on its own stream, so the SpMM launched next on the compute
    stream overlaps it; wait() makes the compute stream depend on the result).  `then` runs the second half of a
    two-step exchange once the first has arrived."""

    def __init__(self, work, then=None):
        self.work, self.then = work, then

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.then is not None:
            nxt, self.then = self.then, None
            nxt().wait()


def capture_mode():
    """capture_error_mode for a hipGraph capture that holds collectives: "thread_local" while a process group exists --
    RCCL's watchdog THREAD polls the events of earlier eager collectives (hipEventQuery), which a capture in the default
    global mode forbids to every thread of the process: the poll then throws inside the watchdog and takes the process
    down (seen on the first captured exchange that followed eager ones closely).  Thread-local mode only restricts the
    capturing thread."""
    return "thread_local" if dist.is_initialized() else "global"


WATCHDOG_SETTLE_S = float(_os.environ.get("CHAOREC_CAPTURE_SETTLE_S", "0.3"))


def settle_before_capture():
    """Call right before a hipGraph capture that will hold collectives, after the eager launches that precede it.  RCCL's
    watchdog thread keeps polling the end events of EAGER collectives until it has seen them complete (its loop sleeps
    100 ms); a capture pulls RCCL's stream (and the p2p exchange's side stream) into capture mode, and on this stack
    (ROCm 7.2) hipEventQuery on an event whose stream is capturing NOW -- although the event was recorded before the
    capture -- fails with hipErrorCapturedEvent and invalidates the capture ("operation failed due to a previous error
    during capture" on the next launch; seen once in ~30 captures that followed eager exchanges within milliseconds).
    So: let the device finish, then give the watchdog time to retire what it holds."""
    if dist.is_initialized() and torch.cuda.is_available():
        import time as _time
        torch.cuda.synchronize()
        if WATCHDOG_SETTLE_S > 0:
            _time.sleep(WATCHDOG_SETTLE_S)


CAPTURE_ATTEMPTS = int(_os.environ.get("CHAOREC_CAPTURE_ATTEMPTS", "3"))
CAPTURE_LOG = []          # (what, attempts it took) of every capture that went through capture_with_retry


def _is_capture_race(exc):
    """The watchdog race of settle_before_capture(): hipErrorCapturedEvent / hipErrorStreamCaptureInvalidated and what torch
    makes of them ("operation failed due to a previous error during capture", "... when stream is capturing")."""
    msg = str(exc).lower()
    return "captur" in msg


def capture_with_retry(capture, reset, what="step", attempts=None):
    """settle_before_capture() lowers the odds of RCCL's watchdog polling an eager collective's event while its stream is
    being captured; it cannot exclude it (the watchdog may still hold events under load).  A capture lost to that race is
    simply taken again: `capture()` (which settles, then captures) up to `attempts` times, `reset()` in between (device
    idle, state restored -- a failed capture has run nothing, but the caller's eager warm-up may have).  Any other error, or
    the last attempt's, propagates.  -> the number of attempts it took (also appended to CAPTURE_LOG)."""
    attempts = CAPTURE_ATTEMPTS if attempts is None else attempts
    for k in range(attempts):
        try:
            capture()
            CAPTURE_LOG.append((what, k + 1))
            return k + 1
        except RuntimeError as exc:
            if k + 1 >= attempts or not _is_capture_race(exc):
                raise
            import sys as _sys
            print(f"[chaorec_amd.dist] capture of {what} lost to the watchdog race ({str(exc)[:120]}): attempt {k + 2} of "
                  f"{attempts}", file=_sys.stderr, flush=True)
            reset()
    return attempts


_SIDE_GROUPS = {}         # id of the main group (None: the default group) -> (the main group object, its side communicator)


def side_group(group=None):
    """A second communicator over the same ranks as `group`, for collectives that run CONCURRENTLY with the group's own (a
    model whose branches compute on two streams: two collectives of one communicator must never be in flight together).
    Created once per group and reused by every model built on it -- dist.new_group is collective over the WHOLE default
    group (every rank of the job must call this at the same point, ranks outside `group` included) and a communicator is
    never given back until destroy_side_groups(); building one per model leaked one per bench loop / test.  No process
    group, or a single rank without forced collectives: `group` itself."""
    if not (dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES)):
        return group
    key = id(group) if group is not None else None
    held = _SIDE_GROUPS.get(key)
    if held is not None and held[0] is group:
        return held[1]
    ranks = dist.get_process_group_ranks(group) if group is not None else None
    side = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
    _SIDE_GROUPS[key] = (group, side)
    return side


def destroy_side_groups():
    """Give the side communicators back (call before dist.destroy_process_group(), next to P2PExchange.forget_all())."""
    for _, side in _SIDE_GROUPS.values():
        try:
            dist.destroy_process_group(side)
        except Exception:      # noqa: BLE001 -- the default group may already be gone
            pass
    _SIDE_GROUPS.clear()


class _PendingStream:
    """Handle of an exchange made of plain launches on a side stream (the hand-written p2p exchange): wait() makes the
    compute stream depend on everything the side stream was given so far."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.stream = None


def _mean_all(x):
    """Scalar mean that survives hipGraph replay on the GPU (ops.mean_all); the gloo/CPU tests of the exchange logic
    run the same expression through torch."""
    return ops.mean_all(x) if x.is_cuda else x.mean()


class P2PExchange:
    """The `direct` exchange pattern (every row block crosses one xGMI link once: reduce-scatter by pull, all-gather by pull)
    written by hand over IPC-mapped peer buffers -- plain kernel launches (csrc/exchange.hip), so a captured step can contain
    it, which RCCL's all-to-all cannot on this stack.  Per buffer size every rank owns two "mailboxes" that all its peers map
    once (torch's storage sharing = hipIpcGetMemHandle / hipIpcOpenMemHandle; set-up is a collective on the host and cannot
    happen inside a capture): P for its partial, R (one row block) for the block it reduces.  One exchange: P <- partial;
    barrier; R = sum of the ranks' P rows of my block, rank order; barrier; every rank's R into the caller's buffer.  Two
    barriers per exchange are enough for any sequence of exchanges (csrc/exchange.hip).  The barrier is a one-element
    all-reduce on the stream with RCCL (a kernel boundary on every rank); with gloo (the ranks-on-one-GPU tests) a stream
    synchronisation + host barrier, eager only.
    Needs every rank to see all the node's GPUs under the same indices (torchrun's default: LOCAL_RANK picks the device).
    Never run across xGMI: the boxes of this build have one GPU; opt-in."""

    _by_group = {}

    @classmethod
    def of(cls, group):
        """The exchange of this process group.  Keyed on the group AND what it looks like now: after
        destroy_process_group() + a new init (or a new group object re-using an id) the old instance -- old world size,
        rank, mailboxes, peer pointers -- must not be handed out again."""
        pg = group if group is not None else dist.group.WORLD
        key = id(pg)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        inst = cls._by_group.get(key)
        if inst is None or inst.pg is not pg or (inst.world, inst.rank) != (world, rank):
            inst = cls._by_group[key] = cls(group)
            inst.pg = pg                           # (a strong reference: the id cannot be re-used while this entry lives)
        return inst

    @classmethod
    def forget_all(cls):
        """Drop every cached exchange (call after dist.destroy_process_group())."""
        cls._by_group.clear()

    def __init__(self, group):
        self.group = group
        self.pg = None
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rccl = dist.get_backend(group) == "nccl"
        self.boxes = {}          # numel -> dict(P, R: this rank's mailboxes; keep: the peers' mapped tensors; pP, pR: pointer arrays)
        self.flag = None
        self._side = None

    def side_stream(self, device):
        """The stream the exchange's launches run on, so that the compute stream's next SpMM overlaps them."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def ready(self, numel):
        return numel in self.boxes

    def setup(self, numel, device):
        """Collective, host-synchronous: allocate this rank's mailboxes for buffers of `numel` floats and map every peer's."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("P2PExchange.setup inside a hipGraph capture (run one eager step first)")
        import ctypes
        block = numel // self.world
        mail = [torch.zeros(numel, dtype=torch.float32, device=device), torch.zeros(block, dtype=torch.float32, device=device)]
        # every rank must see all the node's GPUs under the same indices (torchrun's default): a handle is opened on the
        # OWNER's device index -- with per-rank HIP_VISIBLE_DEVICES that index names another (or no) device here
        n_vis = torch.cuda.device_count()
        mine = [(m.untyped_storage()._share_cuda_(), m.storage_offset()) for m in mail] + [torch.device(device).index or 0, n_vis]
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        for r, entry in enumerate(everyone):
            if entry[2] >= n_vis or entry[3] != n_vis:
                raise RuntimeError(f"P2PExchange: rank {r} lives on device index {entry[2]} of {entry[3]} visible devices, "
                                   f"this rank sees {n_vis}: the ranks do not share one device numbering")
        keep, ptrs = [], []
        for k, n in enumerate((numel, block)):
            row = []
            for r in range(self.world):
                if r == self.rank:
                    row.append(mail[k])
                    continue
                handle, off = everyone[r][k]
                st = torch.UntypedStorage._new_shared_cuda(*handle)      # opened on the OWNER's device index
                t = torch.empty(0, dtype=torch.float32, device=st.device).set_(st, off, (n,))
                if t.device != torch.device(device):
                    # the pull kernels run on this rank's device and read the peer's memory through the mapping: peer
                    # access must be on (torch switches it on in its device-to-device copy path)
                    torch.empty(1, dtype=torch.float32, device=device).copy_(t[:1])
                row.append(t)
            keep.append(row)
            ptrs.append((ctypes.c_void_p * self.world)(*[t.data_ptr() for t in row]))
        if self.flag is None:
            self.flag = torch.zeros(1, dtype=torch.float32, device=device)
        self.boxes[numel] = dict(P=mail[0], R=mail[1], keep=keep, pP=ptrs[0], pR=ptrs[1])
        torch.cuda.synchronize()
        dist.barrier(group=self.group)

    def _barrier(self):
        if self.rccl:
            dist.all_reduce(self.flag, group=self.group)        # on the stream; capturable
        else:
            torch.cuda.current_stream().synchronize()
            dist.barrier(group=self.group)

    def exchange(self, buf, bits=None, n_rows=None):
        """buf [rows_pad, D] fp32 on the GPU, rows_pad a multiple of the world size: summed over the ranks in place.
        bits (with n_rows = the rows it covers): a bitmap over buf's rows, IDENTICAL on every rank, outside of which buf is
        all-zero on every rank (a frontier buffer) -- only the flagged rows are copied, pulled and written back."""
        numel = buf.numel()
        if not self.ready(numel):
            self.setup(numel, buf.device)
        box = self.boxes[numel]
        block = numel // self.world
        lib = _lib.load()
        stream = ops._stream()
        if bits is not None:
            D = buf.shape[1]
            nb = buf.shape[0] // self.world
            _lib.check(lib.chaorec_rows_copy_by_bits_f32(ops._ptr(box["P"]), ops._ptr(buf), n_rows, D, ops._ptr(bits), stream),
                       "chaorec_rows_copy_by_bits_f32")
            self._barrier()
            _lib.check(lib.chaorec_exchange_pull_sum_rows_f32(box["pP"], self.world, self.rank * nb, nb, n_rows, D, ops._ptr(bits),
                                                              ops._ptr(box["R"]), stream), "chaorec_exchange_pull_sum_rows_f32")
            self._barrier()
            _lib.check(lib.chaorec_exchange_pull_gather_rows_f32(box["pR"], self.world, nb, n_rows, D, ops._ptr(bits), ops._ptr(buf),
                                                                 stream), "chaorec_exchange_pull_gather_rows_f32")
            return
        box["P"].copy_(buf.reshape(-1))
        self._barrier()
        _lib.check(lib.chaorec_exchange_pull_sum_f32(box["pP"], self.world, self.rank * block, block, ops._ptr(box["R"]),
                                                     stream), "chaorec_exchange_pull_sum_f32")
        self._barrier()
        _lib.check(lib.chaorec_exchange_pull_gather_f32(box["pR"], self.world, block, ops._ptr(buf), stream),
                   "chaorec_exchange_pull_gather_f32")


def _p2p_usable(buf, group):
    return buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and \
        (buf.numel() // dist.get_world_size(group)) % 4 == 0 and buf.shape[0] % dist.get_world_size(group) == 0 and \
        dist.get_world_size(group) <= 16


def _sum_exchange_async(buf, group, sync=False, bits=None, n_rows=None):
    """Sum `buf` ([rows_pad, D], rows_pad a multiple of the world size) over the ranks, in place; -> a handle to wait on.
    bits / n_rows: a FRONTIER buffer -- all-zero on every rank outside the rows flagged in `bits` (a bitmap over its first
    n_rows rows, identical on every rank).  The hand-written p2p exchange then moves the flagged rows only; RCCL's dense
    collectives sum the whole buffer (same result).
    sync=True: c10d's synchronous form (the CURRENT stream waits for the collective, nothing to wait on afterwards) -- what a
    model whose compute runs on SEVERAL streams must use inside a captured step: on this stack (ROCm 7.2, RCCL 2.26 of torch
    2.10) `async_op=True` collectives issued from a second capturing stream segfault in the capture, the synchronous form
    from the same streams is fine, eagerly both are (tools/rccl_streams_repro.py, profiles/r04_f_rccl_streams_repro.txt)."""
    if not _active(group):
        return _Pending(None)
    return _sum_exchange_issue(buf, group, sync, bits, n_rows)


def _sum_exchange_issue(buf, group, sync=False, bits=None, n_rows=None):
    _count(buf)
    mode = resolve_mode(buf)
    if mode == "direct" and buf.is_cuda and torch.cuda.is_current_stream_capturing():
        # RCCL's all-to-all cannot be captured on this stack (ROCm 7.2 / RCCL of torch 2.10: a captured
        # all_to_all_single hangs or segfaults even alone in a graph, tools/direct_capture_repro.py,
        # profiles/r03_a_all_to_all_capture_repro.log; reduce-scatter and all-gather capture fine).  A captured step that
        # asked for `direct` gets the same pattern from the hand-written peer-to-peer exchange when its mailboxes exist
        # (an eager step ran first and CHAOREC_DIST_DIRECT_CAPTURE=p2p), else the two-phase exchange through RCCL's
        # reduce-scatter
        global DIRECT_FELL_BACK, DIRECT_WENT_P2P
        if _os.environ.get("CHAOREC_DIST_DIRECT_CAPTURE", "rs_ag") == "p2p" and _p2p_usable(buf, group) and \
                P2PExchange.of(group).ready(buf.numel()):
            DIRECT_WENT_P2P = True
            mode = "p2p"
        else:
            DIRECT_FELL_BACK = True
            mode = "rs_ag"
    elif mode == "direct" and buf.is_cuda and _os.environ.get("CHAOREC_DIST_DIRECT_CAPTURE", "rs_ag") == "p2p" and \
            _p2p_usable(buf, group) and not P2PExchange.of(group).ready(buf.numel()):
        P2PExchange.of(group).setup(buf.numel(), buf.device)       # (eager call: the mailboxes a later capture will use)
    if mode == "p2p":
        if not _p2p_usable(buf, group):
            mode = "allreduce"
        else:
            ex = P2PExchange.of(group)
            if not ex.ready(buf.numel()):
                ex.setup(buf.numel(), buf.device)          # (collective, eager only: raises inside a capture)
            MODES_USED.add("p2p")
            if bits is not None:
                STATS["frontier_exchanges"] = STATS.get("frontier_exchanges", 0) + 1
            if sync:
                ex.exchange(buf, bits, n_rows)            # (on the caller's stream: nothing to wait on)
                return _Pending(None)
            side = ex.side_stream(buf.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ex.exchange(buf, bits, n_rows)
            return _PendingStream(side)
    MODES_USED.add("allreduce" if (mode == "allreduce" or buf.shape[0] % dist.get_world_size(group)) else mode)
    lazy = not sync                                    # (async_op=False returns no handle: _Pending(None) waits for nothing)
    if mode == "allreduce" or buf.shape[0] % dist.get_world_size(group):
        return _Pending(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=lazy))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = buf.shape[0] // world
    mine = buf[rank * rows:(rank + 1) * rows]
    if mode == "rs_ag":
        chunk = torch.empty_like(mine)
        rs = dist.reduce_scatter_tensor(chunk, buf, op=dist.ReduceOp.SUM, group=group, async_op=lazy)
        return _Pending(rs, lambda: _Pending(dist.all_gather_into_tensor(buf, chunk, group=group, async_op=lazy)))
    recv = torch.empty_like(buf)                       # block j of `recv` = rank j's partial of MY row block
    a2a = dist.all_to_all_single(recv, buf, group=group, async_op=lazy)

    def gather():
        # (the two-pass column sum of the kernels library on the GPU: fixed order, no memset node under capture)
        blocks = recv.view(world, -1)
        chunk = (ops.col_sum(blocks) if recv.is_cuda else blocks.sum(0)).view(rows, -1)
        return _Pending(dist.all_gather_into_tensor(buf, chunk, group=group, async_op=lazy))

    return _Pending(a2a, gather)


def _all_reduce_async(t, group):
    """Whole-tensor all-reduce (small / unpadded tensors)."""
    if _active(group):
        _count(t)
        return _Pending(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
    return _Pending(None)


def calibrate_exchange(n_rows, D, device, group=None, captured=False, reps=3, candidates=("allreduce", "rs_ag", "p2p")):
    """First contact + choice for exchange buffers of [n_rows (padded), D] floats on THIS node, before a step trusts them:
    every candidate mode sums a seeded random buffer through the product path (_sum_exchange_async: eagerly, and --
    captured=True -- from a replayed hipGraph on fresh contents, twice) and is compared with dist.all_reduce of the same
    data; the result must also be the SAME BITS on every rank (the replicated item rows must not drift apart).  A mode
    that raises or differs is vetoed for the rest of the process (resolve_mode then gives rs_ag / allreduce); among the
    ones that passed, the fastest (max over ranks of the median of `reps` timed exchanges) becomes what `auto` picks
    for buffers of this size.  Collective: every rank calls it with the same arguments.  -> the table (also kept in
    CALIBRATION for the bench line).  What it covers that no 1-GPU test can: the p2p exchange's cross-device visibility
    (peer kernels' writes read through an IPC mapping after a stream-ordered RCCL barrier)."""
    if not _active(group):
        return {}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = padded_rows(n_rows, group)
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"
    gen = torch.Generator(device=dev).manual_seed(977 + rank)

    def fresh():
        return torch.rand((rows, D), generator=gen, device=dev, dtype=torch.float32) - 0.5

    def agreed(ok):
        t = torch.tensor([1.0 if ok else 0.0], device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(t.item() > 0)

    def same_on_all_ranks(t):
        hi, lo = t.clone(), t.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        return bool(torch.equal(hi, lo))

    def check(out, src):
        ref = src.clone()
        dist.all_reduce(ref, group=group)
        same = same_on_all_ranks(out)               # (collective: never behind a rank-local condition)
        return bool(torch.allclose(out, ref, rtol=0, atol=2e-6 * world)) and same

    table = {}
    for mode in candidates:
        if mode in _VETOED:
            table[mode] = {"ok": False, "why": "vetoed before calibration"}
            continue
        why, ms = None, None
        _FORCED[0] = mode

        def stage(fn):
            """One stage on every rank, then ONE agreement on whether it worked everywhere: the ranks enter the next
            stage together or not at all (a rank-local exception must not leave the others inside a collective alone)."""
            nonlocal why
            try:
                good = fn()
            except Exception as exc:      # noqa: BLE001 -- "this mode does not work here", never a wrong step
                good, why = False, why or repr(exc)[:200]
            return agreed(bool(good))

        def eager():
            nonlocal why
            src = fresh()
            out = src.clone()
            _sum_exchange_async(out, group).wait()
            if on_gpu:
                torch.cuda.synchronize()
            if not check(out, src):
                why = "eager sum differs from all_reduce"
                return False
            return True

        def replayed():
            nonlocal why
            static_src, static_out = fresh(), torch.empty((rows, D), device=dev)
            g = torch.cuda.CUDAGraph()
            settle_before_capture()
            with torch.cuda.graph(g, capture_error_mode=capture_mode()):
                static_out.copy_(static_src)
                _sum_exchange_async(static_out, group).wait()
            good = True
            for _ in range(2):
                static_src.copy_(fresh())
                g.replay()
                torch.cuda.synchronize()
                if not check(static_out, static_src) and good:        # (check() is collective: run it both times)
                    good, why = False, "captured sum differs from all_reduce on replay"
            return good

        def frontier():
            """The p2p exchange's FRONTIER form (flagged rows only: the row-sparse backward / light forward of a shard):
            a bitmap that is the same on every rank, values in a rank's own part of it, zeros elsewhere -- eagerly and
            replayed from a hipGraph, against all_reduce."""
            nonlocal why
            g0 = torch.Generator(device=dev).manual_seed(4242)          # (the same on every rank)
            flagged = torch.rand(n_rows, generator=g0, device=dev) < 0.03
            idx = torch.nonzero(flagged).flatten().cpu().numpy()
            words = np.zeros((n_rows + 31) // 32 + 1, dtype=np.uint32)
            np.bitwise_or.at(words, idx >> 5, np.uint32(1) << (idx & 31).astype(np.uint32))
            bits = torch.from_numpy(words.view(np.int32)).to(dev)

            def mine():
                t = torch.zeros((rows, D), device=dev)
                own = flagged & (torch.rand(n_rows, generator=gen, device=dev) < 0.6)
                t[:n_rows][own] = (torch.rand((int(own.sum()), D), generator=gen, device=dev) - 0.5)
                return t

            src = mine()
            out = src.clone()
            _sum_exchange_async(out, group, bits=bits, n_rows=n_rows).wait()
            torch.cuda.synchronize()
            if not check(out, src):
                why = "eager frontier sum differs from all_reduce"
                return False
            if not captured:
                return True
            static_src, static_out = mine(), torch.empty((rows, D), device=dev)
            g = torch.cuda.CUDAGraph()
            settle_before_capture()
            with torch.cuda.graph(g, capture_error_mode=capture_mode()):
                static_out.copy_(static_src)
                _sum_exchange_async(static_out, group, bits=bits, n_rows=n_rows).wait()
            good = True
            for _ in range(2):
                static_src.copy_(mine())
                g.replay()
                torch.cuda.synchronize()
                if not check(static_out, static_src) and good:
                    good, why = False, "captured frontier sum differs from all_reduce on replay"
            return good

        def timed():
            nonlocal ms
            import time as _time
            out = fresh()
            times = []
            for _ in range(reps):
                if on_gpu:
                    torch.cuda.synchronize()
                dist.barrier(group=group)
                t0 = _time.perf_counter()
                _sum_exchange_async(out, group).wait()
                if on_gpu:
                    torch.cuda.synchronize()
                times.append(_time.perf_counter() - t0)
            t = torch.tensor([float(np.median(times))], dtype=torch.float64,
                             device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            ms = float(t.item()) * 1e3
            return True

        try:
            ok = stage(eager)
            if ok and captured and on_gpu:
                ok = stage(replayed)
            frontier_checked = False
            if ok and mode == "p2p" and on_gpu:
                ok = frontier_checked = stage(frontier)
            if ok:
                ok = stage(timed)
        finally:
            _FORCED[0] = None
        if not ok:
            veto(mode)
        table[mode] = {"ok": ok, "ms": ms if ok else None}
        if mode == "p2p" and on_gpu:
            table[mode]["frontier_form_ok"] = bool(frontier_checked)
        if why:
            table[mode]["why"] = why
    good = {m: e["ms"] for m, e in table.items() if e["ok"] and e["ms"] is not None}
    if good:
        best = min(good, key=good.get)
        _AUTO_BIG_CHOICE[rows * D] = best
        table["chosen"] = best
        failed = [m for m, e in table.items() if isinstance(e, dict) and not e.get("ok")]
        table["why_chosen"] = ("fastest of the modes that passed every check on this node (max over ranks of the median of %d "
                               "timed exchanges of %d x %d floats): " % (reps, rows, D)
                               + " < ".join("%s %.3f ms" % (m, good[m]) for m in sorted(good, key=good.get))
                               + ("; failed / vetoed: " + ", ".join("%s (%s)" % (m, table[m].get("why", "?")) for m in failed)
                                  if failed else ""))
    else:
        table["chosen"] = None
        table["why_chosen"] = "no candidate passed: `auto` stays on the plain all-reduce"
    CALIBRATION[rows * D] = dict(table, rows=rows, D=D, bytes=rows * D * 4, captured=bool(captured and on_gpu))
    return table


# ---- the sharded models and steps live in their own modules since round 5 (dist.py had grown to 1 900 lines);
# ---- everything is still reachable as chaorec_amd.dist.<name>: resolved on first use (PEP 562), because those modules import
# ---- names from this one -- an import at the end of this file would break whoever imports one of them FIRST.
_MOVED = {
    "dist_lightgcn": (
        "_ShardedLayerMean", "sharded_layer_mean_propagate", "ShardedLightGCN", "gather_ranklists",
        "_HipStepKernels", "SPLIT_BYTES", "FusedShardedLightGCNStep", "build_weak_scaling_job",
    ),
    "dist_models": (
        "_ShardedPropagate", "joined_loop_csr", "_ShardedPropagateJoined", "propagate_joined_fwd",
        "propagate_joined_bwd", "ShardedGraph", "GradBucket", "allreduce_grads", "SHARDED_MMGCN_STREAMS_DEFAULT",
        "ShardedMMGCN", "_SumGradAcrossRanks", "global_kth_smallest", "ShardedFREEDOM",
    ),
}


def __getattr__(name):
    for module, names in _MOVED.items():
        if name in names:
            import importlib
            value = getattr(importlib.import_module("." + module, __package__), name)
            globals()[name] = value
            return value
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(set(globals()) | {n for names in _MOVED.values() for n in names})
