"""Optimizer and model-specific wrappers: Adam launches (single tensor, multi-tensor, the low-rank feature-table update),
ops.linear_rows (a Linear over gathered rows), and the kernels of the sparse.mm model family (edge dropout, dynamic-value
SpMM, NGCF's layer, weighted sampling without replacement, LayerGCN's row re-weighting).  Moved out of ops.py in round 5
 with no behaviour change; `chaorec_amd.ops` re-exports every name here."""
import ctypes
import os  # noqa: F401

import torch

from . import _lib
from . import ops as _ops
from .ops import RowScatterToken, _f32c, _need_cuda, _ptr, _stream, col_sum, spmm_raw  # noqa: F401
from .ops_dense import gemm_nn_bf16x3, gemm_nt_bf16x3, gemm_raw  # noqa: F401


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              step_dev=None, numel=None):
    """One fused Adam launch over a flat fp32 tensor.  `step_dev` (int32 device scalar) overrides `step` so the
    launch can be captured in a hipGraph.  `numel` > param.numel(): the four arrays continue contiguously past this
    tensor (adjacent parameters updated by one launch, see optim.FusedAdam)."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq, step_dev)
    rc = _lib.load().chaorec_adam_step_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                           param.numel() if numel is None else int(numel),
                                           lr, betas[0], betas[1], eps, weight_decay, int(step), _ptr(step_dev),
                                           _stream())
    _lib.check(rc, "chaorec_adam_step_f32")


def adam_bias_table(n_steps, betas, device):
    """float [n_steps, 2]: (1 - beta1^s, sqrt(1 - beta2^s)) per step s (entry 0 unused), computed by the device with the
    same double-precision expression every Adam launch uses for its own step."""
    t = torch.empty((int(n_steps), 2), dtype=torch.float32, device=device)
    _lib.check(_lib.load().chaorec_adam_bias_table(_ptr(t), int(n_steps), betas[0], betas[1], _stream()),
               "chaorec_adam_bias_table")
    return t


def unique_rows(rows, claim, stamp_dev, out_list, out_count):
    """out_list[0 .. out_count[0]) = the distinct ids in `rows` (int64, duplicates allowed), any order; `claim` int32
    [n_rows] and `stamp_dev` int32 [1]: scratch the launches keep between them (zero-initialised once)."""
    _need_cuda(rows, claim, stamp_dev, out_list, out_count)
    rows = rows.to(torch.int64).contiguous()
    if out_list.numel() < rows.numel():
        raise ValueError("unique_rows: the list must hold as many ids as `rows`")
    _lib.check(_lib.load().chaorec_unique_rows(_ptr(rows), rows.numel(), claim.numel(), _ptr(claim), _ptr(stamp_dev),
                                               _ptr(out_list), _ptr(out_count), _stream()), "chaorec_unique_rows")


def adam_lowrank_strips(K):
    return int(_lib.load().chaorec_adam_lowrank_strips(int(K)))


def adam_lowrank(param, gy, weight, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 step_dev=None, mode=0, last=None, bc_table=None, rowlist=None, rows_given=True):
    """Adam on a feature table [n, K] whose gradient is gy [n, R] @ weight [R, K], never materialised
    (chaorec_adam_lowrank_f32).  mode 0: every row, every step; 1: only the rows with a non-zero gy row, after they
    caught up on the zero-gradient steps they sat out (`last` int32 [strips, n]); 2: flush -- every row catches up;
    3: the rows flagged by a non-zero row of `gy` catch up (before a forward reads them).
    rowlist (modes 1, 3): (list int32 [cap], count int32 [1]); rows_given: filled by unique_rows -- visit exactly these
    rows; else scratch (cap >= n) the launch fills with the rows whose gy (flag) row is non-zero."""
    _need_cuda(param, exp_avg, exp_avg_sq, step_dev, last, bc_table)
    if rowlist is None and mode in (1, 3):               # scratch for the launch's own row scan
        rowlist = (torch.empty(param.shape[0], dtype=torch.int32, device=param.device),
                   torch.empty(1, dtype=torch.int32, device=param.device))
        rows_given = False
    rl, rc_, cap = (rowlist[0], rowlist[1], rowlist[0].numel()) if rowlist is not None else (None, None, 0)
    _need_cuda(rl, rc_)
    n, K = param.shape
    R = 1
    if mode <= 1:
        _need_cuda(gy, weight)
        gy, weight = _f32c(gy), _f32c(weight)
        if tuple(gy.shape) != (n, weight.shape[0]) or weight.shape[1] != K:
            raise ValueError(f"adam_lowrank: param {tuple(param.shape)} gy {tuple(gy.shape)} weight {tuple(weight.shape)}")
        R = weight.shape[0]
    elif mode == 3 and gy is not None:
        _need_cuda(gy)
        gy = _f32c(gy)                                   # row flags [n, R]
        if gy.dim() != 2 or gy.shape[0] != n:
            raise ValueError(f"adam_lowrank: flags {tuple(gy.shape)} for a table of {n} rows")
        R = gy.shape[1]
    if not (param.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()):
        raise ValueError("adam_lowrank: contiguous tables")
    rc = _lib.load().chaorec_adam_lowrank_f32(_ptr(param), _ptr(gy if mode != 2 else None),
                                              _ptr(weight if mode <= 1 else None), _ptr(exp_avg), _ptr(exp_avg_sq),
                                              n, K, R, lr, betas[0], betas[1], eps, weight_decay, int(step),
                                              _ptr(step_dev), int(mode), _ptr(last), _ptr(bc_table),
                                              0 if bc_table is None else bc_table.shape[0], _ptr(rl), _ptr(rc_), cap,
                                              int(bool(rows_given)), _stream())
    _lib.check(rc, "chaorec_adam_lowrank_f32")


class _LinearRows(torch.autograd.Function):
    """y = (x W^T + b)[rows], computed on the gathered rows only (a row of a Linear depends on that row alone).
    Model/FREEDOM.py:209-213 projects the whole trainable feature table every step and then reads the 2 B rows of the
    batch: 2 B x K instead of I x K of reads, and a gradient  gy W  that is non-zero in those rows only.  When the
    optimizer has claimed x (optim.FusedAdam, chaorec_adam_lowrank_f32) the [I, K] gradient is never formed: the
    optimizer receives gy (scattered to [I, R]) and W instead; otherwise x.grad is the usual dense tensor."""

    @staticmethod
    def forward(ctx, x, rows, weight, bias, scatter_token=None):
        sink = getattr(x, "_chaorec_lowrank_sink", None)
        ctx.scatter_token = scatter_token
        ctx.row_token = None
        if sink is not None and sink.lazy_rows and sink.accepts(x):
            ctx.row_token = sink.catch_up(x, rows)       # lazily updated table: these rows must be current first
        xg = x.index_select(0, rows)
        if _ops.LINEAR_FORWARD == "bf16x3" and xg.shape[1] >= 64 and xg.shape[0] >= 256:
            y = gemm_nt_bf16x3(xg, weight, bias=bias)
        else:
            y = gemm_raw(xg, weight, transB=True, bias=bias)
        ctx.save_for_backward(xg, rows, weight)
        ctx.x_param, ctx.has_bias = x, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        xg, rows, weight = ctx.saved_tensors
        x = ctx.x_param
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gy_full = ctx.scatter_token.take(x.shape[0], gy.shape[1], rows) if ctx.scatter_token is not None else None
            if gy_full is None:                          # (else: already scattered by the multi-term BPR backward's launch)
                gy_full = torch.zeros((x.shape[0], gy.shape[1]), dtype=gy.dtype, device=gy.device)
                gy_full.index_add_(0, rows, gy)          # an item can sit in the batch more than once
            sink = getattr(x, "_chaorec_lowrank_sink", None)
            if sink is not None and sink.accepts(x):
                sink.submit(x, gy_full, weight, ctx.row_token)
            elif _ops.LINEAR_FORWARD == "bf16x3" and weight.shape[0] >= 64 and gy_full.shape[0] >= 256:
                gx = gemm_nn_bf16x3(gy_full, weight)
            else:
                gx = gemm_raw(gy_full, weight)
        gw = gemm_raw(gy, xg, transA=True) if ctx.needs_input_grad[2] else None
        gb = col_sum(gy) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return gx, None, gw, gb, None


def linear_rows(x, rows, weight, bias=None):
    """== linear(x, weight, bias)[rows].  The result carries a RowScatterToken (`_chaorec_row_scatter`): a bpr_loss_multi
    that takes it as a gathered term hands this node the already scattered row gradient through it."""
    token = RowScatterToken()
    y = _LinearRows.apply(x, rows, weight, bias, token)
    y._chaorec_row_scatter = token
    return y


def adam_multi(tensors, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step_dev=None):
    """One Adam launch over several small tensors: `tensors` = [(param, grad, exp_avg, exp_avg_sq, numel)], at most
    adam_multi_max() of them (chaorec_adam_multi_f32; same arithmetic as adam_step)."""
    n = len(tensors)
    if n == 0:
        return
    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
    for t in tensors:
        _need_cuda(t[0], t[1], t[2], t[3])
    numel = (ctypes.c_int64 * n)(*[int(t[4]) for t in tensors])
    rc = _lib.load().chaorec_adam_multi_f32(n, arr([t[0].data_ptr() for t in tensors]),
                                            arr([t[1].data_ptr() for t in tensors]),
                                            arr([t[2].data_ptr() for t in tensors]),
                                            arr([t[3].data_ptr() for t in tensors]), numel, lr, betas[0], betas[1], eps,
                                            weight_decay, int(step), _ptr(step_dev), _stream())
    _lib.check(rc, "chaorec_adam_multi_f32")


def adam_multi_max():
    return int(_lib.load().chaorec_adam_multi_max())


class _SpMMAdd(torch.autograd.Function):
    """y = A x + z in one launch (FREEDOM's `i_g_embeddings + h`, Model/FREEDOM.py:168,181)."""

    @staticmethod
    def forward(ctx, x, z, csr):
        ctx.csr = csr
        return _ops.spmm_raw(csr, x, z=_f32c(z), beta=1.0)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        return _ops.spmm_raw(ctx.csr.t(), gy), gy, None


def spmm_add(csr, x, z):
    return _SpMMAdd.apply(x, z, csr)


# --------------------------------------------------------------------------------------------
# per-step edge dropout (NGCF) and weighted edge sampling (FREEDOM)
# --------------------------------------------------------------------------------------------
def edge_dropout_norm(structure, p, seed, step=0, step_dev=None, salt=0, keep=None):
    """Values of the dropped-and-renormalised graph and of its transpose over graph.DropoutStructure (one call per
    NGCFConv.forward, Model/NGCF.py:38-58).  Returns (val, val_t), both fp32 [nnz] in the structure's entry order.
    `keep` (uint8 [nnz], optional) replaces the generator with an externally drawn mask."""
    _need_cuda(structure.col, step_dev, keep)
    nnz = structure.nnz
    dev = structure.col.device
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    val_t = torch.empty(nnz, dtype=torch.float32, device=dev)
    if keep is not None:
        keep = keep.to(torch.uint8).contiguous()
    rc = _lib.load().chaorec_edge_dropout_norm(_ptr(structure.entry_row), _ptr(structure.col),
                                               _ptr(structure.transpose_entry), nnz, structure.n_rows, float(p),
                                               int(seed) & (2**64 - 1), int(step), _ptr(step_dev), int(salt),
                                               _ptr(keep), _ptr(structure.deg_ws), _ptr(val), _ptr(val_t), _stream())
    _lib.check(rc, "chaorec_edge_dropout_norm")
    return val, val_t


def edge_dot_raw(entry_row, col, a, b, n_entries=None):
    """out[k] = <a[entry_row[k]], b[col[k]]> for the first n_entries stored entries (all of them by default): one pass that
    reads two rows per entry (chaorec_edge_dot_f32) instead of two [nnz, D] gathers, their product and its row sum."""
    _need_cuda(a, b, entry_row, col)
    for name, t in (("entry_row", entry_row), ("col", col)):      # (the kernel reads 4-byte indices: an int64 tensor would be
        if t.dtype != torch.int32 or not t.is_contiguous():       #  read as pairs of int32 -- garbage rows, out-of-bounds gathers)
            raise TypeError(f"edge_dot: {name} must be a contiguous int32 tensor, got {t.dtype}, contiguous={t.is_contiguous()}")
    a, b = _f32c(a), _f32c(b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError(f"edge_dot: a {tuple(a.shape)} and b {tuple(b.shape)} must be 2-D with the same width")
    n = int(col.numel()) if n_entries is None else int(n_entries)
    if n < 0 or n > int(col.numel()) or n > int(entry_row.numel()):
        raise ValueError(f"edge_dot: n_entries={n} beyond the {int(col.numel())} stored entries / {int(entry_row.numel())} entry rows")
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    rc = _lib.load().chaorec_edge_dot_f32(_ptr(entry_row), _ptr(col), _ptr(a), _ptr(b), _ptr(out), n, a.shape[1], _stream())
    _lib.check(rc, "chaorec_edge_dot_f32")
    return out


class _EdgeDot(torch.autograd.Function):
    """Edge scores over (the first n entries of) a symmetric-pattern structure, differentiable in both tables: with G the sparse
    matrix that holds the incoming gradient at those entries,  d a = G b,  d b = G^T a  -- two dynamic-values SpMMs."""

    @staticmethod
    def forward(ctx, a, b, structure, n):
        ctx.structure, ctx.n = structure, n
        ctx.save_for_backward(a, b)
        return edge_dot_raw(structure.entry_row, structure.col, a, b, n)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        st, n = ctx.structure, ctx.n
        g = g.contiguous()
        if n != st.nnz:
            g = torch.cat([g, g.new_zeros(st.nnz - n)])
        ga = _ops.spmm_raw(st.with_values(g), _f32c(b)) if ctx.needs_input_grad[0] else None
        gb = _ops.spmm_raw(st.with_values(g[st.transpose_entry.long()]), _f32c(a)) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def edge_dot(structure, a, b, n_entries=None):
    """<a[row_k], b[col_k]> per stored entry k of `structure` (a graph.DropoutStructure: symmetric pattern, entry -> row and
    entry -> reversed-entry maps), or of its first n_entries entries -- the (user, item) half of the bipartite structures
    the models build --, with autograd into both tables.  Needs a.shape[0] = b.shape[0] = the structure's node count."""
    return _EdgeDot.apply(a, b, structure, structure.nnz if n_entries is None else int(n_entries))


class _SpMMValues(torch.autograd.Function):
    """y = A x where A's values change every call (edge dropout, learned edge weights) while its structure is fixed:
    forward with `val`, backward with `val_t` (A^T in the same structure).  Values that require a gradient get one
    (what torch.sparse.mm gives a sparse operand's values): d val[k] = <gy[row_k], x[col_k]> per stored entry -- `val_t`
    is only the transposed COPY the backward SpMM reads, no gradient flows through it."""

    @staticmethod
    def forward(ctx, x, structure, val, val_t):
        ctx.structure = structure
        ctx.val_grad = ctx.needs_input_grad[2]
        ctx.save_for_backward(val_t, x if ctx.val_grad else None)
        return _ops.spmm_raw(structure.with_values(val.detach()), x)

    @staticmethod
    def backward(ctx, gy):
        val_t, x = ctx.saved_tensors
        st, gy = ctx.structure, gy.contiguous()
        gx = _ops.spmm_raw(st.with_values(val_t.detach()), gy) if ctx.needs_input_grad[0] else None
        gval = edge_dot_raw(st.entry_row, st.col, gy, x) if ctx.val_grad else None
        return gx, None, gval, None


def spmm_values(structure, val, val_t, x):
    return _SpMMValues.apply(x, structure, val, val_t)


class _NGCFLayer(torch.autograd.Function):
    """leaky_relu_0.2(s W1^T + (s * x) W2^T): the dense half of NGCFConv (Model/NGCF.py:68-84) as two MFMA GEMM launches
    (the second accumulating into the first's output with the activation in its epilogue) after one product launch;
    backward: the activation's mask in one launch (chaorec_leaky_bwd_f32), four GEMMs, and the product's backward plus
    the sum into s's other gradient in one launch (chaorec_mul_pair_bwd_f32)."""

    @staticmethod
    def forward(ctx, s, x, w1, w2):
        t = s * x
        y = gemm_raw(s, w1, transB=True)
        gemm_raw(t, w2, transB=True, out=y, accumulate=True, act=2)
        ctx.save_for_backward(s, x, w1, t, w2, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        s, x, w1, t, w2, y = ctx.saved_tensors
        lib = _lib.load()
        gy = gy.contiguous()
        g = torch.empty_like(gy)
        _lib.check(lib.chaorec_leaky_bwd_f32(_ptr(y), _ptr(gy), 0.2, _ptr(g), g.numel(), _stream()), "chaorec_leaky_bwd_f32")
        gs = gemm_raw(g, w1)
        gt = gemm_raw(g, w2)
        gx = torch.empty_like(x)
        _lib.check(lib.chaorec_mul_pair_bwd_f32(_ptr(gt), _ptr(s), _ptr(x), _ptr(gs), _ptr(gx), gs.numel(), _stream()),
                   "chaorec_mul_pair_bwd_f32")
        gw1 = gemm_raw(g, s, transA=True) if ctx.needs_input_grad[2] else None
        gw2 = gemm_raw(g, t, transA=True) if ctx.needs_input_grad[3] else None
        return gs, gx, gw1, gw2


def ngcf_layer(s, x, w1, w2):
    """leaky_relu_0.2(s W1^T + (s * x) W2^T) for [N, D] tables with D a multiple of 4."""
    _need_cuda(s, x, w1, w2)
    return _NGCFLayer.apply(_f32c(s), _f32c(x), w1, w2)


def weighted_sample_keep(weights, k, seed, step=0, step_dev=None, return_keys=False):
    """uint8 [n] keep mask of a weighted sample without replacement of k of the n entries (FREEDOM's
    torch.multinomial(edge_values, k), Model/FREEDOM.py:151, as a set; any n)."""
    _need_cuda(weights, step_dev)
    weights = _f32c(weights)
    n = weights.numel()
    lib = _lib.load()
    nbytes = lib.chaorec_weighted_sample_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=weights.device)
    keep = torch.empty(n, dtype=torch.uint8, device=weights.device)
    keys = torch.empty(n, dtype=torch.int64, device=weights.device) if return_keys else None
    rc = lib.chaorec_weighted_sample_keep(_ptr(weights), n, int(k), int(seed) & (2**64 - 1), int(step), _ptr(step_dev),
                                          _ptr(ws), nbytes, _ptr(keep), _ptr(keys), _stream())
    _lib.check(rc, "chaorec_weighted_sample_keep")
    return (keep, keys) if return_keys else keep


def weighted_sample_keys(weights, ids=None, seed=0, step=0, step_dev=None):
    """int64 [n] race keys of chaorec_weighted_sample_keep alone, entry j numbered ids[j] (its number in the whole
    edge list; None: j): one rank's share of a sharded pruning (dist.ShardedFREEDOM).  Non-negative as int64 for
    positive weights; entries with weight <= 0 get -1 (all 64 bits set: never among the k smallest of the unsigned order)."""
    _need_cuda(weights, ids, step_dev)
    weights = _f32c(weights)
    n = weights.numel()
    if ids is not None:
        ids = ids.to(torch.int64).contiguous()
        if ids.numel() != n:
            raise ValueError("weighted_sample_keys: ids and weights differ in length")
    keys = torch.empty(n, dtype=torch.int64, device=weights.device)
    rc = _lib.load().chaorec_weighted_sample_keys(_ptr(weights), _ptr(ids), n, int(seed) & (2**64 - 1), int(step),
                                                  _ptr(step_dev), _ptr(keys), _stream())
    _lib.check(rc, "chaorec_weighted_sample_keys")
    return keys


# --------------------------------------------------------------------------------------------
# row-wise cosine re-weighting (LayerGCN)
# --------------------------------------------------------------------------------------------
class _RowCosineScale(torch.autograd.Function):
    """out = cosine_similarity(y, e, dim=-1)[:, None] * y (Model/LayerGCN.py:125-127): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, y, e):
        _need_cuda(y, e)
        y, e = _f32c(y), _f32c(e)
        out = torch.empty_like(y)
        rc = _lib.load().chaorec_row_cosine_scale_fwd_f32(_ptr(y), _ptr(e), _ptr(out), None, y.shape[0], y.shape[1],
                                                          _stream())
        _lib.check(rc, "chaorec_row_cosine_scale_fwd_f32")
        ctx.save_for_backward(y, e)
        return out

    @staticmethod
    def backward(ctx, g):
        y, e = ctx.saved_tensors
        g = g.contiguous()
        gy, ge = torch.empty_like(y), torch.empty_like(e)
        rc = _lib.load().chaorec_row_cosine_scale_bwd_f32(_ptr(g), _ptr(y), _ptr(e), _ptr(gy), _ptr(ge), y.shape[0],
                                                          y.shape[1], _stream())
        _lib.check(rc, "chaorec_row_cosine_scale_bwd_f32")
        return gy, ge


def row_cosine_scale(y, e):
    return _RowCosineScale.apply(y, e)
