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
