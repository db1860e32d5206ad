"""Dense products behind the model classes: the f32 / split-bf16 MFMA GEMM wrappers, `nn.Linear` as an autograd node
(ops.linear), MMGCN's fused layer and its row passes, F.normalize over a virtual concatenation.  Moved out of ops.py in round
5 with no behaviour change; `chaorec_amd.ops` re-exports every name here (and keeps LINEAR_FORWARD, which
tests switch on the facade: it is read through `_ops` at call time)."""
import ctypes
import os as _os

import torch

from . import _lib
from . import ops as _ops
from .ops import _f32c, _f32rows, _need_cuda, _ptr, _stream, col_sum, spmm_raw  # noqa: F401


# --------------------------------------------------------------------------------------------
# dense layers
# --------------------------------------------------------------------------------------------
def gemm_raw(A, B, transA=False, transB=False, bias=None, out=None, accumulate=False, act=0):
    _need_cuda(A, B, bias, out)
    A, B = _f32c(A), _f32c(B)
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    Kb, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != Kb:
        raise ValueError(f"gemm: inner dims {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("gemm: `out` must be fp32 [M, N] with unit column stride")
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device) if nbytes else None
    rc = lib.chaorec_gemm_f32(_ptr(A), _ptr(B), _ptr(out), _ptr(bias), M, N, K, A.shape[1], B.shape[1],
                              out.stride(0), int(transA), int(transB), int(accumulate), act, _ptr(ws), nbytes,
                              _stream())
    _lib.check(rc, "chaorec_gemm_f32")
    return out


def gemm_nt_bf16x3(x, weight, bias=None, act=0, out=None):
    """y = act(x W^T + b) on the bf16 MFMA pipe with every fp32 operand split into three bf16 planes (fp32-grade
    accuracy, chaorec_gemm_nt_bf16x3): the forward of nn.Linear.  x and `out` may be column slices of wider buffers."""
    _need_cuda(x, weight, bias, out)
    x, weight = _f32rows(x), _f32rows(weight)
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"gemm_nt_bf16x3: inner dims {K} vs {weight.shape[1]}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("gemm_nt_bf16x3: `out` must be fp32 [M, N] with unit column stride")
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_nt_bf16x3_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    rc = lib.chaorec_gemm_nt_bf16x3(_ptr(x), _ptr(weight), _ptr(out), _ptr(bias), M, N, K, x.stride(0), weight.stride(0),
                                    out.stride(0), act, _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_gemm_nt_bf16x3")
    return out


def gemm_tn_bf16x3(gy, x, out=None):
    """gy^T x  ([rows, M]^T [rows, N] -> [M, N]) on the bf16 MFMA pipe, three bf16 planes per fp32 operand
    (chaorec_gemm_tn_bf16x3): the weight gradient of nn.Linear."""
    _need_cuda(gy, x, out)
    gy, x = _f32rows(gy), _f32rows(x)
    K, M = gy.shape
    N = x.shape[1]
    if x.shape[0] != K:
        raise ValueError(f"gemm_tn_bf16x3: row counts {K} vs {x.shape[0]}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_tn_bf16x3_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    rc = lib.chaorec_gemm_tn_bf16x3(_ptr(gy), _ptr(x), _ptr(out), M, N, K, gy.stride(0), x.stride(0), out.stride(0),
                                    _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_gemm_tn_bf16x3")
    return out


def gemm_nn_bf16x3(gy, weight, out=None, accumulate=False):
    """gy W  ([M, K] [K, N] -> [M, N], W = an nn.Linear weight [out, in] as it lies in memory) on the bf16 MFMA pipe,
    three bf16 planes per fp32 operand (chaorec_gemm_nn_bf16x3): the input gradient of nn.Linear."""
    _need_cuda(gy, weight, out)
    gy, weight = _f32rows(gy), _f32rows(weight)
    M, K = gy.shape
    N = weight.shape[1]
    if weight.shape[0] != K:
        raise ValueError(f"gemm_nn_bf16x3: inner dims {K} vs {weight.shape[0]}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=gy.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("gemm_nn_bf16x3: `out` must be fp32 [M, N] with unit column stride")
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_nn_bf16x3_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gy.device) if nbytes else None
    if accumulate and out is None:
        raise ValueError("gemm_nn_bf16x3: accumulate needs `out`")
    rc = lib.chaorec_gemm_nn_bf16x3(_ptr(gy), _ptr(weight), _ptr(out), M, N, K, gy.stride(0), weight.stride(0),
                                    out.stride(0), int(bool(accumulate)), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_gemm_nn_bf16x3")
    return out


def _dual_ok(*ts):
    """Operands the dual (two-segment) GEMMs take: fp32, unit column stride, row strides multiples of 4, 16-byte aligned."""
    return all(t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and
               t.data_ptr() % 16 == 0 for t in ts)


def gemm_nt_bf16x3_dual(x, w1, w2, b1, b2, act1, act2):
    """(act1(x w1^T + b1), act2(x w2^T + b2)) in ONE launch (chaorec_gemm_nt_bf16x3_dual): two Linears over the same input."""
    _need_cuda(x, w1, w2, b1, b2)
    x, w1, w2 = _f32rows(x), _f32rows(w1), _f32rows(w2)
    M, K, N1, N2 = x.shape[0], x.shape[1], w1.shape[0], w2.shape[0]
    y1 = torch.empty((M, N1), dtype=torch.float32, device=x.device)
    y2 = torch.empty((M, N2), dtype=torch.float32, device=x.device)
    rc = _lib.load().chaorec_gemm_nt_bf16x3_dual(_ptr(x), _ptr(w1), _ptr(w2), _ptr(y1), _ptr(y2), _ptr(b1), _ptr(b2), M, N1, N2, K,
                                                 x.stride(0), w1.stride(0), w2.stride(0), N1, N2, act1, act2, _stream())
    _lib.check(rc, "chaorec_gemm_nt_bf16x3_dual")
    return y1, y2


def gemm_nn_bf16x3_dual(g1, g2, w1, w2):
    """g1 w1 + g2 w2 as ONE product [g1 | g2] [w1; w2] (chaorec_gemm_nn_bf16x3_dual): the input gradient of two Linears over the
    same input."""
    _need_cuda(g1, g2, w1, w2)
    g1, g2, w1, w2 = _f32rows(g1), _f32rows(g2), _f32rows(w1), _f32rows(w2)
    M, K1, K2, N = g1.shape[0], g1.shape[1], g2.shape[1], w1.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=g1.device)
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_nn_bf16x3_dual_workspace_bytes(M, N, K1 + K2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=g1.device) if nbytes else None
    rc = lib.chaorec_gemm_nn_bf16x3_dual(_ptr(g1), _ptr(g2), _ptr(w1), _ptr(w2), _ptr(out), M, N, K1, K2, g1.stride(0), g2.stride(0),
                                         w1.stride(0), w2.stride(0), N, _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_gemm_nn_bf16x3_dual")
    return out


def gemm_tn_bf16x3_dual(g1, g2, x):
    """(g1^T x, g2^T x) as ONE product [g1 | g2]^T x (chaorec_gemm_tn_bf16x3_dual): both weight gradients of two Linears over the
    same input."""
    _need_cuda(g1, g2, x)
    g1, g2, x = _f32rows(g1), _f32rows(g2), _f32rows(x)
    K, M1, M2, N = g1.shape[0], g1.shape[1], g2.shape[1], x.shape[1]
    o1 = torch.empty((M1, N), dtype=torch.float32, device=x.device)
    o2 = torch.empty((M2, N), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.chaorec_gemm_tn_bf16x3_dual_workspace_bytes(M1 + M2, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    rc = lib.chaorec_gemm_tn_bf16x3_dual(_ptr(g1), _ptr(g2), _ptr(x), _ptr(o1), _ptr(o2), M1, M2, N, K, g1.stride(0), g2.stride(0),
                                         x.stride(0), N, N, _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_gemm_tn_bf16x3_dual")
    return o1, o2


# which pipe nn.Linear's FORWARD runs on: "bf16x3" (split-bf16 MFMA, fp32-grade accuracy, 2.7x the f32 matrix rate) or
# "f32" (the exact k-ascending fmaf chain of chaorec_gemm_f32); the backward GEMMs follow it.


def _linear_fwd_raw(x, weight, bias, act, out=None):
    """act(x W^T + b): the bf16 MFMA pipe (three bf16 planes per fp32 operand) where the reduction is long enough to pay
    for the split, else the f32 MFMA pipe.  `out` may be a column slice of a wider buffer."""
    if _ops.LINEAR_FORWARD == "bf16x3" and x.shape[1] >= 64 and x.shape[0] >= 256:
        return gemm_nt_bf16x3(x, weight, bias=bias, act=act, out=out)
    return gemm_raw(x, weight, transB=True, bias=bias, act=act, out=out)


def _leaky_bwd_raw(y, gy, act):
    """gy * leaky_relu'(y) for act 1 (slope 0.01) / 2 (slope 0.2), one launch."""
    slope = 0.01 if act == 1 else 0.2
    if gy.numel() % 4 == 0:
        gy = gy.contiguous()
        g = torch.empty_like(gy)
        _lib.check(_lib.load().chaorec_leaky_bwd_f32(_ptr(y), _ptr(gy), slope, _ptr(g), g.numel(), _stream()),
                   "chaorec_leaky_bwd_f32")
        return g
    return torch.where(y > 0, gy, gy * slope)


def _linear_gx_raw(gy, weight, out=None, accumulate=False):
    """The input gradient gy W (W read as it lies: NN product) on the pipe the forward used; `accumulate`: out += gy W."""
    if _ops.LINEAR_FORWARD == "bf16x3" and weight.shape[0] >= 64 and gy.shape[0] >= 256:
        return gemm_nn_bf16x3(gy, weight, out=out, accumulate=accumulate)
    return gemm_raw(gy, weight, out=out, accumulate=accumulate)


def _linear_gw_raw(gy, x):
    """The weight gradient gy^T x: a reduction over all rows -- the split-bf16 pipe where its 128-row tile is not half
    padding (out >= 128: 768^2 over 60 k rows 994 -> 744 us, 256^2 128 -> 94 us; the 64-wide layers' gradients are faster
    on the f32 kernel's 64-row tile)."""
    if _ops.LINEAR_FORWARD == "bf16x3" and gy.shape[0] >= 4096 and gy.shape[1] >= 128 and x.shape[1] >= 64:
        return gemm_tn_bf16x3(gy, x)
    if _ops.LINEAR_FORWARD == "bf16x3" and gy.shape[0] >= 4096 and gy.shape[1] <= 64 and x.shape[1] >= 128:
        # a 64-row gradient of a wide layer: with the operands swapped the WIDE dimension fills the 128-row tiles
        # (x^T gy, then one small transpose): [64, 320] over 60 k rows 70 -> 51 us, [64, 832] 108 -> 86 us
        return gemm_tn_bf16x3(x, gy).t().contiguous()
    return gemm_raw(gy, x, transA=True)


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (nn.Linear [+ F.leaky_relu]): forward, input gradient and weight gradient (a TN product over all
    rows) on the bf16 MFMA pipe (three bf16 planes per fp32 operand) where the reduction is long enough to pay for the
    split, else on the f32 MFMA pipe."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        y = _linear_fwd_raw(x, weight, bias, act)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.has_bias, ctx.act = bias is not None, act
        # a trainable table an optimizer has claimed (optim.FusedAdam): its gradient gy W leaves as (gy, W), see backward
        ctx.x_param = x if getattr(x, "_chaorec_lowrank_sink", None) is not None else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = _leaky_bwd_raw(y, gy, ctx.act) if ctx.act else gy.contiguous()
        gx = None
        xp = ctx.x_param
        # Everything that READS x first: with FusedAdam.early_tables a claimed table's in-place update starts on a side
        # stream the moment submit() is called -- x IS that table here, and the weight gradient below reads all of it
        # (submit() before _linear_gw_raw raced the update against this node's own read).
        gw = _linear_gw_raw(gy, x) if ctx.needs_input_grad[1] else None
        gb = col_sum(gy) if ctx.has_bias and ctx.needs_input_grad[2] else None     # (not gy.sum(0): see col_sum)
        if ctx.needs_input_grad[0] and xp is not None and weight.shape[0] <= 64 and xp._chaorec_lowrank_sink.accepts(xp):
            # the input is a claimed feature table (Model/MGCN.py:80-83: trainable [I, 4096] features projected as a
            # whole): its dense gradient gy W is never formed, the optimizer applies it row by row (adam_lowrank)
            xp._chaorec_lowrank_sink.submit(xp, gy, weight, None, dense_reader=True)
        elif ctx.needs_input_grad[0]:
            gx = _linear_gx_raw(gy, weight)
        return gx, gw, gb, None


def linear(x, weight, bias=None, act=0):
    return _Linear.apply(x, weight, bias, act)


# --------------------------------------------------------------------------------------------
# MMGCN: one layer as one autograd node, F.normalize over a row concatenation
# --------------------------------------------------------------------------------------------
def leaky_cat_add(s, u, id_rows=None, out=None, slope=0.01):
    """[leaky_relu(s) | u + id_rows] in one pass (chaorec_leaky_cat_add_f32)."""
    _need_cuda(s, u, id_rows, out)
    s, u = _f32c(s), _f32c(u)
    id_rows = _f32c(id_rows) if id_rows is not None else None
    n, d1, d2 = s.shape[0], s.shape[1], u.shape[1]
    if out is None:
        out = torch.empty((n, d1 + d2), dtype=torch.float32, device=s.device)
    _lib.check(_lib.load().chaorec_leaky_cat_add_f32(_ptr(s), _ptr(u), _ptr(id_rows), _ptr(out), n, d1, d2, slope, _stream()),
               "chaorec_leaky_cat_add_f32")
    return out


def leaky_split_bwd(gcat, cat, uy, d1, want_gid=False, slope=0.01):
    """-> (gcat[:, :d1] * leaky'(cat[:, :d1]), gcat[:, d1:] * leaky'(uy), gcat[:, d1:] or None), all contiguous, one pass
    (chaorec_leaky_split_bwd_f32)."""
    _need_cuda(gcat, cat, uy)
    gcat, cat, uy = _f32c(gcat), _f32c(cat), _f32c(uy)
    n, d2 = gcat.shape[0], gcat.shape[1] - d1
    gs = torch.empty((n, d1), dtype=torch.float32, device=gcat.device)
    gu = torch.empty((n, d2), dtype=torch.float32, device=gcat.device)
    gid = torch.empty((n, d2), dtype=torch.float32, device=gcat.device) if want_gid else None
    _lib.check(_lib.load().chaorec_leaky_split_bwd_f32(_ptr(gcat), _ptr(cat), _ptr(uy), _ptr(gs), _ptr(gu), _ptr(gid), n, d1,
                                                       d2, slope, _stream()), "chaorec_leaky_split_bwd_f32")
    return gs, gu, gid


# "fused" (one autograd node per MMGCN layer, below) or "unfused" (the composition of linear / spmm / torch ops it
# replaces; the two are bit-identical -- tests/test_gpu_models.py)
MMGCN_LAYER = _os.environ.get("CHAOREC_MMGCN_LAYER", "fused")
# inside the fused layer: conv.lin and linear_layer (two Linears over the same x) as ONE product each way
# (chaorec_gemm_{nt,nn,tn}_bf16x3_dual); 0 = two products each way, bit-identical to the composition
MMGCN_DUAL = _os.environ.get("CHAOREC_MMGCN_DUAL", "1") == "1"


class _MMGCNLayer(torch.autograd.Function):
    """One MMGCN layer, concat branch (Model/MMGCN.py:102-131):
        h = leaky_relu(A (x Wc^T + bc));  u = leaky_relu(x Wl^T + bl) + id;  out = leaky_relu([h | u] Wg^T + bg)
    as one autograd node: 5 launches forward (GEMM, SpMM, GEMM, tail, GEMM), and backward the concatenation's gradient is
    split, masked and made contiguous by one launch, x's two gradient flows meet in a GEMM epilogue.  The composition of
    ops.linear / ops.spmm / F.leaky_relu / + / torch.cat it replaces spent 7 torch launches per layer and direction on
    the same data (the at::native share of the MMGCN step).  Same kernels, same arithmetic: bit-identical --
    except that at full size the two Linears over x (conv.lin, linear_layer) run as ONE product each way (MMGCN_DUAL:
    forward bit-identical, the two backward products with another association of the same sums).
    `csr`: a graph.CSR, or a sharded graph operator with propagate_raw / propagate_t_raw (dist.ShardedGraph, joined form:
    the item rows' exchange happens inside).  `ax_aug` = the cached [A x | A 1 | 0] of a constant input (GCN._constant_input): then h = leaky_relu(ax_aug [Wc | bc |
    0]^T) is written by the GEMM straight into the concatenation's left columns and there is no SpMM either way."""

    @staticmethod
    def forward(ctx, x, id_rows, Wc, bc, Wl, bl, Wg, bg, csr, ax_aug, pad):
        n, d1, d2 = x.shape[0], Wc.shape[0], Wl.shape[0]
        # conv.lin and linear_layer read the same x: ONE product each way where the split-bf16 pipe serves the shape
        dual = (MMGCN_DUAL and ax_aug is None and _ops.LINEAR_FORWARD == "bf16x3" and n >= 4096 and 64 <= x.shape[1] < 512
                and d1 % 4 == 0 and d2 % 4 == 0 and _dual_ok(x, Wc, Wl))
        ctx.dual = dual
        if dual:
            c, uy = gemm_nt_bf16x3_dual(x, Wc, Wl, bc, bl, 0, 1)
        else:
            uy = _linear_fwd_raw(x, Wl, bl, 1)
        if ax_aug is None:
            if not dual:
                c = _linear_fwd_raw(x, Wc, bc, 0)
            s = csr.propagate_raw(c) if hasattr(csr, "propagate_raw") else _ops.spmm_raw(csr, c)
            cat = leaky_cat_add(s, uy, id_rows)
        else:
            cat = torch.empty((n, d1 + d2), dtype=torch.float32, device=x.device)
            w_aug = torch.cat((Wc, bc[:, None], Wc.new_zeros(d1, pad)), 1)
            _linear_fwd_raw(ax_aug, w_aug, None, 1, out=cat[:, :d1])
            right = cat[:, d1:]
            if id_rows is not None:
                torch.add(uy, id_rows, out=right)
            else:
                right.copy_(uy)
        out = _linear_fwd_raw(cat, Wg, bg, 1)
        ctx.save_for_backward(x, Wc, Wl, Wg, cat, uy, out, ax_aug)
        ctx.csr, ctx.d1, ctx.has_id = csr, d1, id_rows is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        x, Wc, Wl, Wg, cat, uy, out, ax_aug = ctx.saved_tensors
        need = ctx.needs_input_grad
        d1 = ctx.d1
        g1 = _leaky_bwd_raw(out, gout, 1)
        gWg = _linear_gw_raw(g1, cat) if need[6] else None
        gbg = col_sum(g1) if need[7] else None
        gcat = _linear_gx_raw(g1, Wg)
        gs, gu, gid = leaky_split_bwd(gcat, cat, uy, d1, want_gid=ctx.has_id and need[1])
        gbl = col_sum(gu) if need[5] else None
        if ax_aug is None and ctx.dual:
            gc = ctx.csr.propagate_t_raw(gs) if hasattr(ctx.csr, "propagate_t_raw") else _ops.spmm_raw(ctx.csr.t(), gs)
            gbc = col_sum(gc) if need[3] else None
            gWc = gWl = None
            if need[2] or need[4]:
                gWc, gWl = gemm_tn_bf16x3_dual(gc, gu, x)            # [gc | gu]^T x
            gx = gemm_nn_bf16x3_dual(gc, gu, Wc, Wl) if need[0] else None      # [gc | gu] [Wc; Wl]
            return gx, gid, gWc, gbc, gWl, gbl, gWg, gbg, None, None, None
        gWl = _linear_gw_raw(gu, x) if need[4] else None
        gx = _linear_gx_raw(gu, Wl) if need[0] else None
        if ax_aug is None:
            gc = ctx.csr.propagate_t_raw(gs) if hasattr(ctx.csr, "propagate_t_raw") else _ops.spmm_raw(ctx.csr.t(), gs)
            gWc = _linear_gw_raw(gc, x) if need[2] else None
            gbc = col_sum(gc) if need[3] else None
            if need[0]:
                gx = _linear_gx_raw(gc, Wc, out=gx, accumulate=True)
        else:
            gWc = gbc = None
            if need[2] or need[3]:
                gw_aug = _linear_gw_raw(gs, ax_aug)
                k = Wc.shape[1]
                gWc, gbc = gw_aug[:, :k].contiguous(), gw_aug[:, k].contiguous()
        return gx, gid, gWc, gbc, gWl, gbl, gWg, gbg, None, None, None


def mmgcn_layer(x, id_rows, conv_lin, lin, g_lin, csr, ax_aug=None, pad=0):
    return _MMGCNLayer.apply(x, id_rows, conv_lin.weight, conv_lin.bias, lin.weight, lin.bias, g_lin.weight, g_lin.bias, csr,
                             ax_aug, pad)


class _NormalizeRows(torch.autograd.Function):
    """F.normalize(torch.cat((a, b), dim=0)) (Model/MMGCN.py:99-100) in one launch each way, the concatenation never
    materialised; the gradient of `a` is only computed when asked for (MMGCN's preference is no Parameter, Q2)."""

    @staticmethod
    def forward(ctx, a, b, eps):
        a = _f32c(a)
        b = _f32c(b) if b is not None else None
        na, nb, D = a.shape[0], (b.shape[0] if b is not None else 0), a.shape[1]
        y = torch.empty((na + nb, D), dtype=torch.float32, device=a.device)
        norm = torch.empty(na + nb, dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().chaorec_normalize_rows_fwd_f32(_ptr(a), _ptr(b), na, na + nb, D, eps, _ptr(y), _ptr(norm),
                                                              _stream()), "chaorec_normalize_rows_fwd_f32")
        ctx.save_for_backward(y, norm)
        ctx.na, ctx.eps = na, eps
        return y

    @staticmethod
    def backward(ctx, gy):
        y, norm = ctx.saved_tensors
        gy = gy.contiguous()
        skip = 0 if ctx.needs_input_grad[0] else ctx.na
        gx = torch.empty_like(y)
        _lib.check(_lib.load().chaorec_normalize_rows_bwd_f32(_ptr(gy), _ptr(y), _ptr(norm), skip, y.shape[0], y.shape[1],
                                                              ctx.eps, _ptr(gx), _stream()), "chaorec_normalize_rows_bwd_f32")
        return (gx[:ctx.na] if ctx.needs_input_grad[0] else None), (gx[ctx.na:] if ctx.needs_input_grad[1] else None), None


def normalize_rows(a, b=None, eps=1e-12):
    """== F.normalize(torch.cat((a, b), dim=0)) (or F.normalize(a)); D a multiple of 4."""
    return _NormalizeRows.apply(a, b, eps)
