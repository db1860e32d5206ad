"""torch-facing wrappers over the C-ABI kernels (include/chaorec_hip.h).

PyTorch here is plumbing: it owns HBM allocations, the current HIP stream and autograd's tape.
Every op takes CUDA (ROCm) tensors, passes raw device pointers + the current stream to
libchaorec_hip.so and returns immediately.  There is no CPU implementation: a CPU tensor or a
missing library raises.
"""
import ctypes
import os

import torch

from . import _lib
from .graph import CSR

VARIANT_LOG_SIGMOID_EPS = 0   # LightGCN  (Model/LightGCN.py:108)
VARIANT_LOGSIGMOID = 1        # FREEDOM   (Model/FREEDOM.py:189)
VARIANT_LOG_SIGMOID = 2       # MMGCN     (Model/MMGCN.py:196)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("chaorec_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _f32rows(t):
    """A 2-D fp32 operand the GEMM kernels can read or write in place: unit column stride, any row stride (a column
    slice of a wider buffer travels as (pointer, ld) -- no copy); anything else is made contiguous."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# SpMM
# --------------------------------------------------------------------------------------------
def spmm_raw(csr, x, y=None, alpha=1.0, z=None, beta=0.0, acc=None, acc_init=None, acc_w=0.0, want_y=True):
    """y = alpha * (A x) [+ beta z]; optional acc epilogue (see chaorec_spmm_csr_f32)."""
    _need_cuda(csr.rowptr, x, z, acc, acc_init)
    x = _f32c(x)
    D = x.shape[1]
    if x.shape[0] != csr.n_cols:
        raise ValueError(f"spmm: x has {x.shape[0]} rows, graph has {csr.n_cols} columns")
    if want_y and y is None:
        y = torch.empty((csr.n_rows, D), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    order = csr.schedule(D)
    mode = 1 if getattr(csr, "dynamic_values", False) else 0     # CHAOREC_SPMM_DYNAMIC_VALUES
    rc = lib.chaorec_spmm_csr_f32(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(x),
                                  _ptr(y if want_y else None), csr.n_rows, csr.n_cols, D, alpha,
                                  _ptr(z), beta, _ptr(acc), _ptr(acc_init), acc_w, _ptr(order), mode, _stream())
    _lib.check(rc, "chaorec_spmm_csr_f32")
    return y


def spmm_mean_raw(csr, x, terms, w, mean_out, y=None):
    """mean_out = w * terms[0] + w * terms[1] + ... + w * (A x), accumulated in that order (LightGCN's layer mean in the
    last forward propagate's epilogue, chaorec_spmm_csr_mean_f32); y (optional) receives A x."""
    _need_cuda(csr.rowptr, x, mean_out, y, *terms)
    x = _f32c(x)
    D = x.shape[1]
    if x.shape[0] != csr.n_cols:
        raise ValueError(f"spmm_mean: x has {x.shape[0]} rows, graph has {csr.n_cols} columns")
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    mode = 1 if getattr(csr, "dynamic_values", False) else 0
    rc = _lib.load().chaorec_spmm_csr_mean_f32(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(x), _ptr(y), csr.n_rows,
                                               csr.n_cols, D, _ptr(mean_out), ctypes.cast(ptrs, ctypes.c_void_p),
                                               len(terms), w, _ptr(csr.schedule(D)), mode, _stream())
    _lib.check(rc, "chaorec_spmm_csr_mean_f32")
    return mean_out


def rows_mean(terms, w, out):
    """out = w * terms[0] + w * terms[1] + ... in that order (chaorec_rows_mean_f32): the layer mean of rows whose
    propagated values arrive after the SpMM launches (the replicated item rows of a user shard)."""
    _need_cuda(out, *terms)
    n = out.numel()
    for t in terms:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise TypeError("rows_mean: contiguous float32 terms of out's size")
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    _lib.check(_lib.load().chaorec_rows_mean_f32(ctypes.cast(ptrs, ctypes.c_void_p), len(terms), float(w), _ptr(out), n,
                                                 _stream()), "chaorec_rows_mean_f32")
    return out


def mean_terms_limit(D):
    """How many earlier-layer tables the last forward propagate can fold into its epilogue for feature width D."""
    return 3 if D <= 64 else 2


def row_bitmap(n_rows, device):
    """An all-clear bitmap over n_rows rows (uint32 words as int32 storage) for the row-sparse backward propagates."""
    return torch.zeros((int(n_rows) + 31) // 32 + 1, dtype=torch.int32, device=device)


def expand_row_bits(csr, bits_in, bits_out, row_list=None, list_n=None, bits_self=None):
    """bits_out |= bits_self | {columns of the rows flagged in bits_in} (chaorec_expand_row_bits): the rows the next propagate
    can make non-zero.  bits_in: over the CSR's rows; bits_self / bits_out: over its columns -- a symmetric graph (bits_self
    defaults to bits_in), or one of a user shard's rectangular blocks with the other side's batch rows as bits_self.  Work ~
    the flagged rows' entries.  row_list (int32 [cap]) / list_n (int32 [1], zero on entry): the rows flagged by THIS launch
    are also appended -- spmm_rowlist_raw's work list."""
    _need_cuda(csr.rowptr, bits_in, bits_out, row_list, list_n, bits_self)
    if bits_self is None:
        if not csr.symmetric:
            raise ValueError("expand_row_bits: a graph that is not its own transpose needs bits_self (a bitmap over its columns)")
        bits_self = bits_in
    if bits_in.numel() * 32 < csr.n_rows or bits_out.numel() * 32 < csr.n_cols or bits_self.numel() * 32 < csr.n_cols:
        raise ValueError("expand_row_bits: bitmap shorter than its row range")
    _lib.check(_lib.load().chaorec_expand_row_bits(_ptr(csr.rowptr), _ptr(csr.col), csr.n_rows, _ptr(bits_in), _ptr(bits_self),
                                                   csr.n_cols, _ptr(bits_out), _ptr(row_list), _ptr(list_n),
                                                   row_list.numel() if row_list is not None else 0, _stream()),
               "chaorec_expand_row_bits")
    return bits_out


def zero_rows_by_bits(y, bits):
    """y[r] = 0 for every row flagged in `bits` (chaorec_zero_rows_by_bits_f32)."""
    _need_cuda(y, bits)
    if y.dtype != torch.float32 or not y.is_contiguous() or bits.numel() * 32 < y.shape[0]:
        raise ValueError("zero_rows_by_bits: contiguous float32 rows and a bitmap over all of them")
    _lib.check(_lib.load().chaorec_zero_rows_by_bits_f32(_ptr(y), y.shape[0], y.shape[1], _ptr(bits), _stream()),
               "chaorec_zero_rows_by_bits_f32")
    return y


def rows_copy_by_bits(dst, src, bits):
    """dst[r] = src[r] for every row flagged in `bits` (chaorec_rows_copy_by_bits_f32)."""
    _need_cuda(dst, src, bits)
    if dst.shape != src.shape or dst.dtype != torch.float32 or not dst.is_contiguous() or not src.is_contiguous() or \
            bits.numel() * 32 < dst.shape[0]:
        raise ValueError("rows_copy_by_bits: two contiguous float32 [n, D] buffers and a bitmap over their rows")
    _lib.check(_lib.load().chaorec_rows_copy_by_bits_f32(_ptr(dst), _ptr(src), dst.shape[0], dst.shape[1], _ptr(bits), _stream()),
               "chaorec_rows_copy_by_bits_f32")
    return dst


def rows_list_from_bits(bits, n_rows, row_list, list_n):
    """row_list[0 .. list_n) = the rows flagged in `bits` (chaorec_rows_list_from_bits; list_n zero on entry)."""
    _need_cuda(bits, row_list, list_n)
    _lib.check(_lib.load().chaorec_rows_list_from_bits(_ptr(bits), int(n_rows), _ptr(row_list), _ptr(list_n), row_list.numel(),
                                                       _stream()), "chaorec_rows_list_from_bits")
    return row_list


def rows_mean_by_bits(terms, w, out, bits):
    """out[r] = w * terms[0][r] + w * terms[1][r] + ... for the rows flagged in `bits` (chaorec_rows_mean_by_bits_f32)."""
    _need_cuda(out, bits, *terms)
    for t in terms:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != out.shape:
            raise TypeError("rows_mean_by_bits: contiguous float32 terms of out's shape")
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    _lib.check(_lib.load().chaorec_rows_mean_by_bits_f32(ctypes.cast(ptrs, ctypes.c_void_p), len(terms), float(w), _ptr(out),
                                                         out.shape[0], out.shape[1], _ptr(bits), _stream()),
               "chaorec_rows_mean_by_bits_f32")
    return out


def frontier_pack(src, bits, prefix, compact, overflow=None):
    """compact[k] = src[flagged row number k, bitmap order] (chaorec_frontier_pack_f32); prefix: int32 [n_words + 1] scratch
    of exactly that length (prefix[-1] = the number of flagged rows afterwards); rows of compact past them are zeroed.
    More flagged rows than compact holds: an error when called eagerly (costs a sync), or -- `overflow`: an int32 [1] device
    tensor -- recorded there (capturable: one small launch)."""
    _need_cuda(src, bits, prefix, compact)
    if src.dtype != torch.float32 or not src.is_contiguous() or not compact.is_contiguous() or compact.shape[1] != src.shape[1] \
            or prefix.numel() != (src.shape[0] + 31) // 32 + 1:
        raise ValueError("frontier_pack: contiguous float32 [n, D] / [cap, D] buffers, prefix of n_words + 1 ints")
    _lib.check(_lib.load().chaorec_frontier_pack_f32(_ptr(src), src.shape[0], src.shape[1], _ptr(bits), _ptr(prefix), _ptr(compact),
                                                     compact.shape[0], _stream()), "chaorec_frontier_pack_f32")
    if overflow is not None:
        # sticky, on the device: by how many rows a frontier ever exceeded the compact buffer (the kernel drops those rows;
        # a caller that sized `cap` by a static bound checks this after the fact -- FusedShardedLightGCNStep.check_frontier)
        torch.maximum(overflow, prefix[-1:] - compact.shape[0], out=overflow)
    elif not torch.cuda.is_current_stream_capturing():
        total = int(prefix[(src.shape[0] + 31) // 32])
        if total > compact.shape[0]:
            raise RuntimeError(f"frontier_pack: {total} flagged rows do not fit the compact buffer's {compact.shape[0]}")
    return compact


def frontier_unpack(dst, bits, prefix, compact):
    """dst[flagged row number k] = compact[k] (chaorec_frontier_unpack_f32), the inverse of frontier_pack."""
    _need_cuda(dst, bits, prefix, compact)
    _lib.check(_lib.load().chaorec_frontier_unpack_f32(_ptr(dst), dst.shape[0], dst.shape[1], _ptr(bits), _ptr(prefix),
                                                       _ptr(compact), compact.shape[0], _stream()), "chaorec_frontier_unpack_f32")
    return dst


def or_words(dst, src):
    """dst[w] = OR_k src[k, w] (chaorec_or_words_u32): the union of all-gathered row bitmaps."""
    _need_cuda(dst, src)
    if src.dim() != 2 or src.shape[1] != dst.numel() or not src.is_contiguous() or not dst.is_contiguous():
        raise ValueError("or_words: src [k, n_words] contiguous, dst [n_words]")
    _lib.check(_lib.load().chaorec_or_words_u32(_ptr(dst), _ptr(src), src.shape[0], dst.numel(), _stream()), "chaorec_or_words_u32")
    return dst


ROWLIST_LONG_T = int(os.environ.get("CHAOREC_ROWLIST_LONG_T", "256"))


def long_row_buffers(csr, threshold=None):
    """(list int32 [number of rows above the threshold], counters int32 [2] zero, threshold) for spmm_rowlist_raw's long_rows."""
    t = ROWLIST_LONG_T if threshold is None else int(threshold)
    n_long = int(((csr.rowptr[1:] - csr.rowptr[:-1]) > t).sum().item())
    dev = csr.rowptr.device
    return torch.zeros(max(n_long, 1), dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev), t


def spmm_rowlist_raw(csr, x, y, row_list, list_n, alpha=1.0, z=None, beta=0.0, src_bits=None, z_bits=None, mean_out=None,
                     mean_terms=(), mean_w=0.0, long_rows=None):
    """y[r] = alpha * (A x)[r] [+ beta z[r]] for the rows of a device-side list only (chaorec_spmm_csr_rowlist_f32); the other
    rows of y are not touched.  Same sums, bit for bit, as spmm_raw's for those rows.  mean_out / mean_terms / mean_w: the
    listed rows of the layer mean, spmm_mean_raw's arithmetic (y may then be None).  long_rows = long_row_buffers(csr):
    listed rows above the threshold are computed by a second launch, one workgroup per row."""
    ll, lc, lt = long_rows if long_rows is not None else (None, None, 0)
    _need_cuda(csr.rowptr, x, y, z, src_bits, z_bits, row_list, list_n, mean_out, ll, lc, *mean_terms)
    x = _f32c(x)
    if x.shape[0] != csr.n_cols or (y is not None and (y.shape[0] != csr.n_rows or not y.is_contiguous())):
        raise ValueError("spmm_rowlist: shape mismatch")
    terms = None
    if mean_out is not None:
        if not 1 <= len(mean_terms) <= 4 or any(t.shape != mean_out.shape or not t.is_contiguous() for t in mean_terms) or \
                mean_out.shape != (csr.n_rows, x.shape[1]) or not mean_out.is_contiguous():
            raise ValueError("spmm_rowlist: 1..4 contiguous mean terms of the output's shape")
        terms = (ctypes.c_void_p * len(mean_terms))(*[t.data_ptr() for t in mean_terms])
    _lib.check(_lib.load().chaorec_spmm_csr_rowlist_f32(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(x), _ptr(y), csr.n_rows,
                                                        x.shape[1], alpha, _ptr(z), beta, _ptr(src_bits), _ptr(z_bits),
                                                        _ptr(row_list), _ptr(list_n), row_list.numel(), _ptr(mean_out), terms,
                                                        len(mean_terms), float(mean_w), _ptr(ll), _ptr(lc),
                                                        ll.numel() if ll is not None else 0, int(lt), _stream()),
               "chaorec_spmm_csr_rowlist_f32")
    return y


def batch_rows(ids, row_bits, bits_item_offset, row_list=None, list_n=None, edges=None, hist=None, num_user=0, num_item=0, seed=0,
               step=0, step_dev=None, perm=None, perm_pos=None, pos_offset=0):
    """The batch BEFORE the forward (chaorec_batch_rows): ids = (users, pos, neg) int64 [B] -- written when `edges` is given
    (bpr_fwd_bwd's draw for the same seed / step / permutation position, LOCAL item ids), read otherwise -- and the three
    table rows of every sample flagged in row_bits (items from bit bits_item_offset on), the rows flagged first appended to
    row_list / list_n."""
    _need_cuda(edges, step_dev, perm, perm_pos, row_bits, row_list, list_n, *ids)
    rowptr, col = hist if hist is not None else (None, None)
    B = ids[0].numel()
    rc = _lib.load().chaorec_batch_rows(_ptr(edges), edges.shape[0] if edges is not None else 0, _ptr(rowptr), _ptr(col), B,
                                        int(num_user), int(num_item), int(seed) & (2**64 - 1), int(step), _ptr(step_dev),
                                        _ptr(perm), _ptr(perm_pos), int(pos_offset), _ptr(ids[0]), _ptr(ids[1]), _ptr(ids[2]),
                                        _ptr(row_bits), int(bits_item_offset), _ptr(row_list), _ptr(list_n),
                                        row_list.numel() if row_list is not None else 0, _stream())
    _lib.check(rc, "chaorec_batch_rows")


def spmm_rowsparse_raw(csr, x, y, alpha=1.0, z=None, beta=0.0, src_bits=None, z_bits=None, out_bits=None, row_bits=None,
                       write_zeros=True):
    """y = alpha * (A x) [+ beta z] for ROW-SPARSE operands (chaorec_spmm_csr_rowsparse_f32): rows of x whose bit in src_bits
    is clear are not gathered, rows of z whose bit in z_bits is clear are not read (they hold exact zeros: the result is the
    dense launch's, bit for bit); out_bits (all-clear on entry) receives a superset of y's non-zero rows.  row_bits: a
    superset of the rows of y that can be non-zero (expand_row_bits of src_bits): the others walk no entries and store zeros
    (write_zeros) or nothing (write_zeros=False: only when every later reader of y gathers flagged rows only)."""
    _need_cuda(csr.rowptr, x, y, z, src_bits, z_bits, out_bits, row_bits)
    x = _f32c(x)
    D = x.shape[1]
    if x.shape[0] != csr.n_cols or y.shape[0] != csr.n_rows:
        raise ValueError("spmm_rowsparse: shape mismatch")
    mode = 1 if getattr(csr, "dynamic_values", False) else 0
    rc = _lib.load().chaorec_spmm_csr_rowsparse_f32(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(x), _ptr(y),
                                                    csr.n_rows, csr.n_cols, D, alpha, _ptr(z), beta, _ptr(csr.schedule(D)),
                                                    mode, _ptr(src_bits), _ptr(z_bits), _ptr(out_bits), _ptr(row_bits),
                                                    int(bool(write_zeros)), _stream())
    _lib.check(rc, "chaorec_spmm_csr_rowsparse_f32")
    return y


def spmm_adam_raw(csr, x, param, exp_avg, exp_avg_sq, bias_corr, lr, betas, eps, weight_decay, alpha=1.0, z=None,
                  beta=0.0, clear_z=False, grad_out=None, clear_bits=()):
    """g = alpha * (A x) [+ beta z] and the Adam update of `param` with that gradient, row by row, in ONE launch
    (chaorec_spmm_csr_adam_f32): the last backward propagate of a LightGCN step with optimizer.step() in its epilogue.
    `bias_corr`: device float[2] written by bpr_finalize().  clear_z: zero the non-zero rows of z after use."""
    _need_cuda(csr.rowptr, x, z, param, exp_avg, exp_avg_sq, bias_corr, grad_out)
    x = _f32c(x)
    D = x.shape[1]
    if x.shape[0] != csr.n_cols or param.shape[0] != csr.n_rows or param.shape[1] != D:
        raise ValueError("spmm_adam: shape mismatch")
    for t in (param, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("spmm_adam: param / moments must be contiguous float32")
    mode = 1 if getattr(csr, "dynamic_values", False) else 0
    cb = (list(clear_bits) + [None, None])[:2]         # row bitmaps this (the step's last) launch zeroes as a side job
    _need_cuda(*cb)
    rc = _lib.load().chaorec_spmm_csr_adam_f32(_ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(x), _ptr(grad_out),
                                               csr.n_rows, csr.n_cols, D, alpha, _ptr(z), beta, _ptr(csr.schedule(D)),
                                               mode, _ptr(param), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(bias_corr),
                                               lr, betas[0], betas[1], eps, weight_decay, int(bool(clear_z)),
                                               _ptr(cb[0]), cb[0].numel() if cb[0] is not None else 0,
                                               _ptr(cb[1]), cb[1].numel() if cb[1] is not None else 0, _stream())
    _lib.check(rc, "chaorec_spmm_csr_adam_f32")


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, csr):
        ctx.csr = csr
        return spmm_raw(csr, x)

    @staticmethod
    def backward(ctx, gy):
        return spmm_raw(ctx.csr.t(), gy.contiguous()), None


def spmm(csr, x):
    """Differentiable y = A x (replaces propagate / torch.sparse.mm)."""
    return _SpMM.apply(x, csr)


class _LayerMeanPropagate(torch.autograd.Function):
    """LightGCN.forward (Model/LightGCN.py:76-95) as L fused launches:
    x_{l+1} = A x_l, final = sum_l w x_l with w = 1/(L+1), the mean folded into each SpMM's
    epilogue.  Backward: g_L = w G; g_l = A^T g_{l+1} + w G."""

    @staticmethod
    def forward(ctx, x0, csr, n_layers):
        x0 = _f32c(x0)
        w = 1.0 / (n_layers + 1)
        final = torch.empty_like(x0)
        if n_layers == 0:
            final.copy_(x0)
        forward_layers(csr, x0, n_layers, final, [torch.empty_like(x0) for _ in range(max(n_layers - 1, 0))])
        ctx.csr, ctx.n_layers, ctx.w = csr, n_layers, w
        return final

    @staticmethod
    def backward(ctx, G):
        G = _f32c(G)
        L, w, At = ctx.n_layers, ctx.w, ctx.csr.t()
        if L == 0:
            return G, None, None
        # g_{L-1} = w * (A^T G) + w * G, then g_l = A^T g_{l+1} + w * G
        g = spmm_raw(At, G, alpha=w, z=G, beta=w)
        for _ in range(L - 1):
            g = spmm_raw(At, g, z=G, beta=w)
        return g, None, None


def forward_layers(csr, x0, n_layers, final, bufs):
    """LightGCN.forward's L propagates into `final` (the layer mean, bit-identical to the reference's accumulation).
    Few layers: x_1 .. x_{L-1} are plain SpMMs into `bufs` and the whole mean is formed in the LAST propagate's epilogue
    (one read of every earlier layer, one write of the mean); more layers than the kernel has operand slots: the
    per-layer acc epilogue (read-modify-write of the mean in every layer).  x_L itself is never written."""
    L = n_layers
    if L == 0:
        return
    w = 1.0 / (L + 1)
    if L <= mean_terms_limit(x0.shape[1]):
        xs = [x0]
        for l in range(L - 1):
            xs.append(spmm_raw(csr, xs[-1], y=bufs[l]))
        spmm_mean_raw(csr, xs[-1], xs, w, final)
        return
    x = x0
    for l in range(L):
        last = l == L - 1
        y = None if last else bufs[l]
        spmm_raw(csr, x, y=y, acc=final, acc_init=x0 if l == 0 else None, acc_w=w, want_y=not last)
        x = y


def layer_mean_propagate(x0, csr, n_layers):
    return _LayerMeanPropagate.apply(x0, csr, n_layers)


# --------------------------------------------------------------------------------------------
# BPR
# --------------------------------------------------------------------------------------------
class _BPR(torch.autograd.Function):
    """-> (total loss, tensor[total, bpr, reg]); tab_i=None means "items live in tab_u from row item_offset on"
    (LightGCN/MMGCN keep users and items in one [N,D] table: one gradient buffer, no slicing)."""

    @staticmethod
    def forward(ctx, tab_u, tab_i, users, pos, neg, variant, reg_weight, item_offset):
        _need_cuda(tab_u, tab_i, users, pos, neg)
        tab_u = _f32c(tab_u)
        D = tab_u.shape[1]
        if tab_i is None:
            pi = ctypes.c_void_p(tab_u.data_ptr() + item_offset * D * 4)
        else:
            tab_i = _f32c(tab_i)
            pi = _ptr(tab_i)
        users, pos, neg = (t.to(torch.int64).contiguous() for t in (users, pos, neg))
        B = users.numel()
        dev = tab_u.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)   # own allocation: its gradient arrives 0-dim
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        ws = torch.empty(4 * B, dtype=torch.float32, device=dev)
        rc = _lib.load().chaorec_bpr_fwd_f32(_ptr(tab_u), pi, _ptr(users), _ptr(pos), _ptr(neg), B, D,
                                             variant, reg_weight, _ptr(out), _ptr(loss), _ptr(coef), _ptr(ws),
                                             _stream())
        _lib.check(rc, "chaorec_bpr_fwd_f32")
        ctx.save_for_backward(tab_u, tab_i, users, pos, neg, coef)
        ctx.reg_weight, ctx.item_offset = reg_weight, item_offset
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)      # no zero-filled gradient tensor for `out` on every backward
        return loss, out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        tab_u, tab_i, users, pos, neg, coef = ctx.saved_tensors
        B, D = users.numel(), tab_u.shape[1]
        g_u = torch.zeros_like(tab_u)
        if tab_i is None:
            g_i = None
            pi = ctypes.c_void_p(tab_u.data_ptr() + ctx.item_offset * D * 4)
            pgi = ctypes.c_void_p(g_u.data_ptr() + ctx.item_offset * D * 4)
        else:
            g_i = torch.zeros_like(tab_i)
            pi, pgi = _ptr(tab_i), _ptr(g_i)
        go = g_loss.contiguous()
        rc = _lib.load().chaorec_bpr_bwd_f32(_ptr(tab_u), pi, _ptr(users), _ptr(pos), _ptr(neg), B, D,
                                             _ptr(coef), ctx.reg_weight, _ptr(go), _ptr(g_u), pgi, _stream())
        _lib.check(rc, "chaorec_bpr_bwd_f32")
        return g_u, g_i, None, None, None, None, None, None


BPR_MULTI_MAX = 4 if os.environ.get("CHAOREC_BPR_MULTI", "1") == "1" else 0     # chaorec_bpr_multi_*_f32's term limit


class _BPRMulti(torch.autograd.Function):
    """sum_k w_k * BPR(tab_u[users], tab_i_k[pos_k], tab_i_k[neg_k]) for several item tables that share the user table and
    the batch's users (Model/FREEDOM.py:203-215: the id-embedding loss + reg_weight * (text loss + image loss)) as ONE
    autograd node: one gradient buffer for the user table that the T backward launches add into (instead of T zero-filled
    buffers and T - 1 additions by autograd); up to four terms run as ONE forward launch, one finalize that also forms the
    weighted sum, and one backward launch (chaorec_bpr_multi_*_f32: 4 launches per step instead of 13)."""

    @staticmethod
    def forward(ctx, tab_u, users, variant, wvec, gathered, tokens, *flat):
        T = len(flat) // 3
        ctx.tokens = list(tokens) if tokens is not None else [None] * T
        _need_cuda(tab_u, users, wvec, *flat)
        # gathered[k] = (rows, n_table_rows) or None: term k's table is a block of rows gathered from a longer table (the
        # projected batch rows of linear_rows): its backward also scatters the gradient into a [n_table_rows, D] buffer
        ctx.gathered = list(gathered) if gathered is not None else [None] * T
        tab_u = _f32c(tab_u)
        users = users.to(torch.int64).contiguous()
        B, D, dev = users.numel(), tab_u.shape[1], tab_u.device
        tabs = [_f32c(flat[3 * k]) for k in range(T)]
        ids = [(flat[3 * k + 1].to(torch.int64).contiguous(), flat[3 * k + 2].to(torch.int64).contiguous()) for k in range(T)]
        totals = torch.empty(T, dtype=torch.float32, device=dev)
        outs = torch.empty((T, 3), dtype=torch.float32, device=dev)
        coef = torch.empty((T, B), dtype=torch.float32, device=dev)
        ws = torch.empty(4 * B, dtype=torch.float32, device=dev)
        lib = _lib.load()
        ctx.T = T
        if T <= BPR_MULTI_MAX:       # all terms in one launch (+ one finalize that also forms the weighted sum)
            ws = torch.empty(4 * B * T, dtype=torch.float32, device=dev)
            total = torch.empty((), dtype=torch.float32, device=dev)
            arr = lambda ts: (ctypes.c_void_p * T)(*[t.data_ptr() for t in ts])
            rc = lib.chaorec_bpr_multi_fwd_f32(_ptr(tab_u), _ptr(users), T, arr(tabs), arr([p for p, _ in ids]),
                                               arr([n for _, n in ids]), B, D, variant, _ptr(wvec), _ptr(totals), _ptr(total),
                                               _ptr(coef), _ptr(ws), _stream())
            _lib.check(rc, "chaorec_bpr_multi_fwd_f32")
            ctx.save_for_backward(tab_u, users, coef, wvec, *tabs, *[t for pn in ids for t in pn])
            return total
        for k in range(T):
            rc = lib.chaorec_bpr_fwd_f32(_ptr(tab_u), _ptr(tabs[k]), _ptr(users), _ptr(ids[k][0]), _ptr(ids[k][1]), B, D,
                                         variant, 0.0, ctypes.c_void_p(outs.data_ptr() + 12 * k),
                                         ctypes.c_void_p(totals.data_ptr() + 4 * k),
                                         ctypes.c_void_p(coef.data_ptr() + 4 * B * k), _ptr(ws), _stream())
            _lib.check(rc, "chaorec_bpr_fwd_f32")
        ctx.save_for_backward(tab_u, users, coef, wvec, *tabs, *[t for pn in ids for t in pn])
        return (totals * wvec).sum()

    @staticmethod
    def backward(ctx, g):
        T = ctx.T
        tab_u, users, coef, wvec = ctx.saved_tensors[:4]
        tabs = ctx.saved_tensors[4:4 + T]
        ids = ctx.saved_tensors[4 + T:]
        B, D = users.numel(), tab_u.shape[1]
        # ONE zero fill for the T + 1 gradient buffers (views of it) -- and for the scattered row gradients of gathered
        # terms --, not one launch each
        gathered = ctx.gathered if T <= BPR_MULTI_MAX else [None] * T
        sizes = [tab_u.numel()] + [t.numel() for t in tabs] + [(gt[1] * D if gt is not None else 0) for gt in gathered]
        flat = torch.zeros(sum(sizes), dtype=tab_u.dtype, device=tab_u.device)
        offs = [0]
        for n_ in sizes:
            offs.append(offs[-1] + n_)
        g_u = flat[:sizes[0]].view_as(tab_u)
        lib = _lib.load()
        grads = []
        if T <= BPR_MULTI_MAX:
            g_is = [flat[offs[k + 1]:offs[k + 2]].view_as(tabs[k]) for k in range(T)]
            arr = lambda ts: (ctypes.c_void_p * T)(*[t.data_ptr() for t in ts])
            g = g.contiguous()
            srows, souts = None, None
            if any(gt is not None for gt in gathered):
                full = [flat[offs[T + 1 + k]:offs[T + 2 + k]].view(gathered[k][1], D) if gathered[k] is not None else None
                        for k in range(T)]
                parr = lambda ts: (ctypes.c_void_p * T)(*[(t.data_ptr() if t is not None else 0) for t in ts])
                srows, souts = parr([gt[0] if gt is not None else None for gt in gathered]), parr(full)
                for k in range(T):           # handed to the gathering node's backward (ops._LinearRows) through ITS token
                    if full[k] is not None and ctx.tokens[k] is not None:
                        ctx.tokens[k].put(full[k], gathered[k][0])
            rc = lib.chaorec_bpr_multi_bwd_f32(_ptr(tab_u), _ptr(users), T, arr(tabs), arr(ids[0::2]), arr(ids[1::2]), B, D,
                                               _ptr(coef), _ptr(wvec), _ptr(g), _ptr(g_u), arr(g_is), srows, souts, _stream())
            _lib.check(rc, "chaorec_bpr_multi_bwd_f32")
            for g_i in g_is:
                grads += [g_i, None, None]
            return (g_u, None, None, None, None, None, *grads)
        gvec = (g * wvec).contiguous()                   # d total / d loss_k, on the device
        for k in range(T):
            g_i = flat[offs[k + 1]:offs[k + 2]].view_as(tabs[k])
            rc = lib.chaorec_bpr_bwd_f32(_ptr(tab_u), _ptr(tabs[k]), _ptr(users), _ptr(ids[2 * k]), _ptr(ids[2 * k + 1]), B,
                                         D, ctypes.c_void_p(coef.data_ptr() + 4 * B * k), 0.0,
                                         ctypes.c_void_p(gvec.data_ptr() + 4 * k), _ptr(g_u), _ptr(g_i), _stream())
            _lib.check(rc, "chaorec_bpr_bwd_f32")
            grads += [g_i, None, None]
        return (g_u, None, None, None, None, None, *grads)


class RowScatterToken:
    """The hand-over between the two autograd nodes around a gathered BPR term: linear_rows() creates one per call, keeps it
    on its node and attaches it to its output; bpr_loss_multi() passes the tokens of its gathered terms to _BPRMulti, whose
    backward launch scatters each such term's gradient into a [n_table_rows, D] buffer itself and put()s it here; the
    gathering node's backward take()s it instead of scattering again.  One producer, one consumer, tied to ONE forward
    call: no process-wide state, nothing to match by address, and a buffer nobody takes dies with the graph."""
    __slots__ = ("_held", "puts", "hits")

    def __init__(self):
        self._held, self.puts, self.hits = None, 0, 0        # (puts / hits: what happened, for tests and debugging)

    def put(self, scattered, rows):
        self._held = (scattered, rows)
        self.puts += 1

    def take(self, n_rows, width, rows):
        held, self._held = self._held, None
        if held is not None and tuple(held[0].shape) == (n_rows, width) and held[1].data_ptr() == rows.data_ptr():
            self.hits += 1
            return held[0]
        return None


class _SplitRows(torch.autograd.Function):
    """(x[:n], x[n:]) of a [N, D] table whose two halves feed different branches (FREEDOM: the propagated user rows go
    to the loss, the item rows through the item-item graph first).  Plain slicing costs the backward two zero-filled
    [N, D] buffers, two slice copies and an add; here it is one concatenation of the two incoming gradients."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rows = n, x.shape[0]
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g_a, g_b):
        if g_a is None or g_b is None:               # (a half that does not reach the loss: its gradient is zero)
            other = g_a if g_a is not None else g_b
            rows = ctx.n if g_a is None else ctx.rows - ctx.n
            zero = other.new_zeros((rows, other.shape[1]))
            g_a, g_b = (zero, g_b) if g_a is None else (g_a, zero)
        if (g_a.is_contiguous() and g_b.is_contiguous() and g_a.dtype == g_b.dtype and g_a.shape[1] == g_b.shape[1]
                and g_a.untyped_storage().data_ptr() == g_b.untyped_storage().data_ptr()
                and g_b.data_ptr() == g_a.data_ptr() + g_a.numel() * g_a.element_size()):
            # the two gradients already lie back to back in one buffer (_BPRMulti.backward lays its gradient buffers out
            # as [g_u | g_i0 | ...] for exactly this): the concatenation is a view
            return torch.as_strided(g_a, (ctx.rows, g_a.shape[1]), (g_a.shape[1], 1)), None
        return torch.cat((g_a, g_b), 0), None


def split_rows(x, n):
    return _SplitRows.apply(x, n)


def bpr_loss_multi(tab_u, users, variant, terms, wvec, gathered=None):
    """sum_k wvec[k] * bpr_loss(tab_u, terms[k] = (tab_i, pos, neg), users)[0] with reg_weight 0 (see _BPRMulti); wvec: a
    float32 device tensor with one weight per term.  gathered (optional): per term None or (rows, n_table_rows) -- the
    term's table is linear_rows(table, rows, ...): the backward then scatters that block's gradient into the
    [n_table_rows, D] row gradient itself (one launch less per table and direction, one zero fill for everything)."""
    flat = [t for term in terms for t in term]
    tokens = None
    if gathered is not None:
        tokens = [getattr(term[0], "_chaorec_row_scatter", None) if gt is not None else None for term, gt in zip(terms, gathered)]
    return _BPRMulti.apply(tab_u, users, int(variant), wvec, gathered, tokens, *flat)


class _LossParts:
    """What bpr_loss returns: indexable like the old [total, bpr, reg] tensor; [0] is the differentiable total."""

    def __init__(self, loss, parts):
        self.loss, self.parts = loss, parts

    def __getitem__(self, i):
        return self.loss if i == 0 else self.parts[i]

    def detach(self):
        return self.parts.detach()


class _BPRDrawn(torch.autograd.Function):
    """_BPR with the batch drawn inside the forward launch (chaorec_bpr_fwd_drawn_f32): the (user, positive, negative)
    ids of ops.draw_batch(edges, hist, B, num_user, num_item, seed, step, step_dev) are produced by the kernel, used
    right away and kept for the backward.  -> (loss, parts, users, pos, neg)."""

    @staticmethod
    def forward(ctx, tab_u, tab_i, edges, hist_rowptr, hist_col, B, num_user, num_item, seed, step, step_dev, variant,
                reg_weight, item_offset, advance, perm, perm_pos):
        _need_cuda(tab_u, tab_i, edges, hist_rowptr, hist_col, step_dev, perm, perm_pos)
        tab_u = _f32c(tab_u)
        D = tab_u.shape[1]
        if tab_i is None:
            pi = ctypes.c_void_p(tab_u.data_ptr() + item_offset * D * 4)
        else:
            tab_i = _f32c(tab_i)
            pi = _ptr(tab_i)
        dev = tab_u.device
        users = torch.empty(B, dtype=torch.int64, device=dev)
        pos, neg = torch.empty_like(users), torch.empty_like(users)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        ws = torch.empty(4 * B, dtype=torch.float32, device=dev)
        rc = _lib.load().chaorec_bpr_fwd_drawn_f32(_ptr(tab_u), pi, _ptr(edges), edges.shape[0], _ptr(hist_rowptr),
                                                   _ptr(hist_col), num_user, num_item, int(seed) & (2**64 - 1),
                                                   int(step), _ptr(step_dev), B, D, variant, reg_weight, _ptr(users),
                                                   _ptr(pos), _ptr(neg), _ptr(out), _ptr(loss), _ptr(coef), _ptr(ws),
                                                   _ptr(step_dev if advance else None), _ptr(perm), _ptr(perm_pos),
                                                   _stream())
        _lib.check(rc, "chaorec_bpr_fwd_drawn_f32")
        ctx.save_for_backward(tab_u, tab_i, users, pos, neg, coef)
        ctx.reg_weight, ctx.item_offset = reg_weight, item_offset
        ctx.mark_non_differentiable(out, users, pos, neg)
        ctx.set_materialize_grads(False)
        return loss, out, users, pos, neg

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        return _BPR.backward(ctx, g_loss, None)[:2] + (None,) * 15


def bpr_loss_drawn(tab_u, tab_i, edges, hist, B, num_user, num_item, seed, step, variant, reg_weight=0.0,
                   item_offset=0, step_dev=None, advance=False, perm=None, perm_pos=None):
    """Fused batch draw + BPR(+L2): ([total, bpr, reg], users, pos, neg); differentiate [0][0].  advance=True: the
    launch also moves the device counter `step_dev` on by one (after the draw), for captured steps.  perm / perm_pos
    (int64 device tensors): take the edges from an epoch permutation at position *perm_pos (advanced by B too)."""
    loss, parts, users, pos, neg = _BPRDrawn.apply(tab_u, tab_i, edges, hist[0], hist[1], int(B), int(num_user),
                                                   int(num_item), seed, step, step_dev, int(variant),
                                                   float(reg_weight), int(item_offset),
                                                   bool(advance and step_dev is not None), perm, perm_pos)
    return _LossParts(loss, parts), users, pos, neg


def bpr_loss(tab_u, tab_i, users, pos, neg, variant, reg_weight=0.0, item_offset=0):
    """Fused BPR(+L2) over a batch of LOCAL row ids -> [total, bpr, reg]; differentiate [0]."""
    loss, parts = _BPR.apply(tab_u, tab_i, users, pos, neg, int(variant), float(reg_weight), int(item_offset))
    return _LossParts(loss, parts)


def bpr_fwd_bwd(tab, item_offset, grad, B, variant, reg_weight, coef, ws, ids, edges=None, hist=None, num_user=0,
                num_item=0, seed=0, step=0, step_dev=None, perm=None, perm_pos=None, adam_step=None, betas=(0.9, 0.999),
                adam_bc=None, pos_offset=0, row_bits=None, bits_item_offset=None):
    """BPR(+L2) forward terms and backward row adds in one launch (chaorec_bpr_fwd_bwd_f32) over ONE [N, D] table
    (items from row item_offset on) and its gradient buffer `grad` (same shape, zero where no sample lands).
    edges given: the batch is drawn in the launch and written to ids = (users, pos, neg); else ids are the batch
    (LOCAL item ids).  The loss comes from bpr_finalize(ws, ...).  adam_step / adam_bc: Adam's step counter is moved on
    and the new step's bias corrections are written by this launch.  pos_offset: added to *perm_pos (step j of a
    replay whose finalize runs once, after its last step: step = j, pos_offset = j * B).  row_bits (optional, ops.row_bitmap
    over the table's rows): the rows of `grad` the launch touched are flagged for the row-sparse backward propagates; item
    row i is bit bits_item_offset + i (default item_offset: one bitmap over the joined table; a word-aligned offset gives
    the item rows a bitmap of their own)."""
    _need_cuda(tab, grad, coef, ws, edges, step_dev, perm, perm_pos, adam_step, adam_bc, row_bits, *ids)
    D = tab.shape[1]
    off = item_offset * D * 4
    ti, gi = ctypes.c_void_p(tab.data_ptr() + off), ctypes.c_void_p(grad.data_ptr() + off)
    draw = edges is not None
    rowptr, col = hist if hist is not None else (None, None)
    rc = _lib.load().chaorec_bpr_fwd_bwd_at_f32(
        _ptr(tab), ti, _ptr(edges), edges.shape[0] if draw else 0, _ptr(rowptr), _ptr(col), int(num_user), int(num_item),
        int(seed) & (2**64 - 1), int(step), _ptr(step_dev), _ptr(None if draw else ids[0]), _ptr(None if draw else ids[1]),
        _ptr(None if draw else ids[2]), int(B), D, int(variant), float(reg_weight), _ptr(ids[0] if draw else None),
        _ptr(ids[1] if draw else None), _ptr(ids[2] if draw else None), _ptr(coef), _ptr(ws), _ptr(perm), _ptr(perm_pos),
        int(pos_offset), _ptr(grad), gi, _ptr(adam_step), betas[0], betas[1], _ptr(adam_bc), _ptr(row_bits),
        int(item_offset if bits_item_offset is None else bits_item_offset), _stream())
    _lib.check(rc, "chaorec_bpr_fwd_bwd_at_f32")


def bpr_finalize(ws, B, D, reg_weight, out_loss, out_total=None, loss_accum=None, advance=None, perm_pos=None,
                 adam_step=None, betas=(0.9, 0.999), adam_bc=None):
    """Reduce a BPR forward's workspace to out_loss = [total, bpr, reg] and do the step's scalar bookkeeping
    (chaorec_bpr_finalize_f32): loss_accum += total, advance += 1, perm_pos += B, adam_step += 1 with its bias
    corrections into adam_bc."""
    _need_cuda(ws, out_loss, out_total, loss_accum, advance, perm_pos, adam_step, adam_bc)
    rc = _lib.load().chaorec_bpr_finalize_f32(_ptr(ws), int(B), int(D), float(reg_weight), _ptr(out_loss), _ptr(out_total),
                                              _ptr(loss_accum), _ptr(advance), _ptr(perm_pos), _ptr(adam_step),
                                              betas[0], betas[1], _ptr(adam_bc), _stream())
    _lib.check(rc, "chaorec_bpr_finalize_f32")


def bpr_finalize_steps(ws, n_steps, B, D, reg_weight, out_loss, out_total=None, loss_accum=None, advance=None, perm_pos=None,
                       scratch=None):
    """bpr_finalize for the n_steps workspaces ws[j] (ws: [n_steps, >= 4 B]) of a replay in ONE launch
    (chaorec_bpr_finalize_steps_f32): the same sums and additions into loss_accum, in step order; out_loss / out_total of
    the last step; advance += n_steps, perm_pos += n_steps * B.  scratch: float32 [2 n_steps + 1], ZERO in its last element
    (a captured step passes its own; without one a fresh buffer is allocated)."""
    _need_cuda(ws, out_loss, out_total, loss_accum, advance, perm_pos, scratch)
    if scratch is None:
        scratch = torch.zeros(2 * int(n_steps) + 1, dtype=torch.float32, device=ws.device)
    elif scratch.numel() < 2 * int(n_steps) + 1 or scratch.dtype != torch.float32:
        raise ValueError("bpr_finalize_steps: scratch must hold 2 n_steps + 1 floats")
    rc = _lib.load().chaorec_bpr_finalize_steps_f32(_ptr(ws), int(ws.stride(0)), int(n_steps), int(B), int(D), float(reg_weight),
                                                    _ptr(out_loss), _ptr(out_total), _ptr(loss_accum), _ptr(advance),
                                                    _ptr(perm_pos), _ptr(scratch), _stream())
    _lib.check(rc, "chaorec_bpr_finalize_steps_f32")


SECOND_DRAW_SALT = 0x9E3779B97F4A7C15      # CHAOREC_SECOND_DRAW_SALT (include/chaorec_hip.h)


def sample_negatives(hist, users, num_item, seed, step, id_offset, step_dev=None, second=False):
    """One uniform negative per user, never in the user's history.  `step_dev` (int64 device scalar) is added to
    `step`: lets a captured hipGraph draw a fresh batch on every replay.  second: the sample's SECOND, independent draw
    (dataload.py:81-84's `int_items`, read by MCLN only): the same function under seed ^ SECOND_DRAW_SALT."""
    if second:
        seed = (int(seed) ^ SECOND_DRAW_SALT) & 0xFFFFFFFFFFFFFFFF
    rowptr, col = hist
    _need_cuda(rowptr, col, users, step_dev)
    users = users.to(torch.int64).contiguous()
    out = torch.empty_like(users)
    rc = _lib.load().chaorec_sample_negatives(_ptr(rowptr), _ptr(col), _ptr(users), users.numel(), num_item,
                                              seed, step, _ptr(step_dev), id_offset, _ptr(out), _stream())
    _lib.check(rc, "chaorec_sample_negatives")
    return out


def draw_batch(edges, hist, B, num_user, num_item, seed, step, step_dev=None, item_offset=0):
    """(users, pos_local + item_offset, neg_local + item_offset) for one batch in ONE launch: uniform edge pick + gather +
    negative draw (item_offset = num_user: the global ids Model.loss() takes)."""
    rowptr, col = hist
    _need_cuda(edges, rowptr, col, step_dev)
    if edges.dtype != torch.int64 or not edges.is_contiguous():
        raise TypeError("draw_batch: edges must be a contiguous int64 [E, 2] tensor")
    out = torch.empty((3, B), dtype=torch.int64, device=edges.device)
    rc = _lib.load().chaorec_draw_batch(_ptr(edges), edges.shape[0], _ptr(rowptr), _ptr(col), B, num_user, num_item,
                                        seed, step, _ptr(step_dev), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                        int(item_offset), _stream())
    _lib.check(rc, "chaorec_draw_batch")
    return out[0], out[1], out[2]


def shift_cat(pos, neg, offset):
    """(pos - offset, neg - offset, cat of the two) in one launch: the two halves are views of the row list."""
    _need_cuda(pos, neg)
    pos, neg = pos.to(torch.int64).contiguous(), neg.to(torch.int64).contiguous()
    B = pos.numel()
    rows = torch.empty(2 * B, dtype=torch.int64, device=pos.device)
    _lib.check(_lib.load().chaorec_shift_cat_i64(_ptr(pos), _ptr(neg), int(offset), B, _ptr(rows), _stream()),
               "chaorec_shift_cat_i64")
    return rows[:B], rows[B:], rows


# --------------------------------------------------------------------------------------------
# scoring + top-K
# --------------------------------------------------------------------------------------------
SCORE_LIGHT = 1      # CHAOREC_SCORE_LIGHT
SCORE_FRONT = 2      # CHAOREC_SCORE_FRONT: pack, sampling, the first sweep over all users
SCORE_BACK = 4       # CHAOREC_SCORE_BACK: selection, retry passes, exact routes
_SCORE_STREAMS = {}


def _score_streams(dev):
    """(front stream, back stream) of the user-range pipeline, per device."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SCORE_STREAMS:
        _SCORE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SCORE_STREAMS[key]


def _score_call(lib, user_emb, item_emb, hist, mask_value, K, id_offset, precision, hint, hint_valid, hint_rank, light,
                counters, idx, val, ws, nbytes, phase=0):
    """One chaorec_score_topk_*_f32 call on the current stream (phase: 0 = whole call, SCORE_FRONT / SCORE_BACK)."""
    U, D = user_emb.shape
    I = item_emb.shape[0]
    rowptr, col = hist if hist is not None else (None, None)
    if (hint is not None or phase or counters is not None) and precision == 0:
        rc = lib.chaorec_score_topk_hinted_f32(_ptr(user_emb), _ptr(item_emb), U, I, D, _ptr(rowptr), _ptr(col),
                                               mask_value, K, id_offset, _ptr(idx), _ptr(val), _ptr(ws), nbytes,
                                               _ptr(hint if hint_valid else None), _ptr(hint), int(hint_rank),
                                               (SCORE_LIGHT if (light and hint_valid and hint is not None) else 0) | phase,
                                               _ptr(counters), _stream())
        _lib.check(rc, "chaorec_score_topk_hinted_f32")
    else:
        rc = lib.chaorec_score_topk_f32(_ptr(user_emb), _ptr(item_emb), U, I, D, _ptr(rowptr), _ptr(col),
                                        mask_value, K, id_offset, _ptr(idx), _ptr(val), _ptr(ws), nbytes,
                                        precision, _stream())
        _lib.check(rc, "chaorec_score_topk_f32")


def _score_stats(lib, ws, U, I, K, D, dev):
    out10 = torch.zeros(10, dtype=torch.int64, device=dev)
    _lib.check(lib.chaorec_score_topk_stats(_ptr(ws), U, I, K, D, _ptr(out10), _stream()), "chaorec_score_topk_stats")
    return out10


def _stats_dict(v):
    return dict(fallback_users=v[0], candidates=v[1], longest_list=v[2], prefilter_users=v[3],
                fallback_reasons=dict(overflow=v[4], too_few=v[5], too_many=v[6], kth_not_above_threshold=v[7]),
                rethreshold_users=v[9])


def score_topk(user_emb, item_emb, hist, mask_value, K, id_offset=0, precision=0, stats=None, hint=None,
               hint_valid=False, hint_rank=80, light=False, counters=None, idx_out=None):
    """Top-K of user_emb @ item_emb.T with history masking, without the [U,I] matrix.
    Returns (idx int64 [U,K] = item + id_offset, val fp32 [U,K]).  `stats`: a dict to fill with the prefilter
    route's counters (chaorec_score_topk_stats; costs a device sync).
    hint (optional, float32 [U] on the device): per-user thresholds carried between calls
    (chaorec_score_topk_hinted_f32): written by every call, read when hint_valid.  Never changes the result.
    light: no retry pass (the caller saw a short retry queue last time); counters: int32 [4] tensor receiving this call's
    queue lengths -- on the device, or PINNED host memory (written by the call's last launch; read it after the stream has
    passed the call).  idx_out: an int64 [U, K] tensor to write the indices to -- a PINNED host tensor is allowed
    (page-locked memory is mapped into the device's address space: the selection then writes the rank list straight over
    PCIe while it runs, instead of a device buffer that is copied afterwards; sync the stream before reading it).

    A call whose workspace (per user: the candidate lists of every sweep split, ~20 KB) would exceed
    CHAOREC_SCORE_WS_LIMIT (24 GiB) is cut into user ranges of equal length -- the users are independent.  With
    CHAOREC_SCORE_PIPELINE=1 the ranges are PIPELINED: range k's back phase (selection, retry passes, exact routes) runs on a
    second stream beside range k + 1's front phase (sampling + the sweep), two workspaces of half the budget in flight.
    Measured at the config-5 shard (1.25 M x 2 M, D = 128): 634 ms against 630 ms for the ranges one after the
    other and 628 ms in one piece -- every kernel of the call fills the chip on its own, running them side by side conserves
    the work -- so the default is one range after the other on the caller's stream."""
    _need_cuda(user_emb, item_emb)
    user_emb, item_emb = _f32c(user_emb), _f32c(item_emb)
    U, D = user_emb.shape
    I = item_emb.shape[0]
    dev = user_emb.device
    lib = _lib.load()
    rowptr, col = hist if hist is not None else (None, None)
    _need_cuda(rowptr, col)
    if idx_out is not None and (idx_out.dtype != torch.int64 or tuple(idx_out.shape) != (U, K) or not idx_out.is_contiguous()
                                or not (idx_out.is_cuda or idx_out.is_pinned())):
        raise TypeError("score_topk: idx_out must be a contiguous int64 [n_users, K] device or pinned host tensor")
    if counters is not None and (counters.dtype != torch.int32 or counters.numel() < 4 or not counters.is_contiguous()
                                 or not (counters.is_cuda or counters.is_pinned())):
        raise TypeError("score_topk: counters must be a contiguous int32 [4] device or pinned host tensor")
    if hint is not None and precision == 0:
        _need_cuda(hint)
        if hint.dtype != torch.float32 or hint.numel() != U or not hint.is_contiguous():
            raise TypeError("score_topk: hint must be a contiguous float32 [n_users] tensor")
    elif precision != 0:
        hint = None
    idx = idx_out if idx_out is not None else torch.empty((U, K), dtype=torch.int64, device=dev)
    val = torch.empty((U, K), dtype=torch.float32, device=dev)
    limit = int(os.environ.get("CHAOREC_SCORE_WS_LIMIT", str(24 << 30)))
    nbytes = lib.chaorec_score_topk_workspace_bytes(U, I, K, D)
    if not (U > 4096 and nbytes > limit):
        ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
        _score_call(lib, user_emb, item_emb, hist, mask_value, K, id_offset, precision, hint, hint_valid, hint_rank, light,
                    counters if precision == 0 and hint is not None else None, idx, val, ws, nbytes)
        if stats is not None:
            stats.update(_stats_dict(_score_stats(lib, ws, U, I, K, D, dev).tolist()))
        return idx, val

    # ---- user ranges (BASELINE configs[4] at 1e7 users would ask for > 200 GB of workspace) -------------------------
    pipelined = (os.environ.get("CHAOREC_SCORE_PIPELINE", "0") == "1" and precision == 0
                 and not torch.cuda.is_current_stream_capturing())
    budget = limit // 2 if pipelined else limit           # (two workspaces in flight when the ranges are pipelined)
    per = max(4096, (U * budget // nbytes) // 4096 * 4096)
    while per > 4096 and lib.chaorec_score_topk_workspace_bytes(per, I, K, D) > budget:
        per -= 4096
    n_ranges = (U + per - 1) // per
    per = min(per, ((U + n_ranges - 1) // n_ranges + 4095) // 4096 * 4096)     # ranges of equal length, not a full one + a rest
    ranges = [(u0, min(U, u0 + per)) for u0 in range(0, U, per)]
    ws_bytes = lib.chaorec_score_topk_workspace_bytes(per, I, K, D)
    pipelined = pipelined and len(ranges) > 1
    want_counters = counters is not None and precision == 0
    tot = torch.zeros(4, dtype=torch.int32, device=dev) if want_counters else None
    stat_sum = torch.zeros(10, dtype=torch.int64, device=dev) if stats is not None else None
    stat_max = torch.zeros(1, dtype=torch.int64, device=dev) if stats is not None else None

    def args_of(u0, u1):
        return (user_emb[u0:u1], item_emb, None if hist is None else (rowptr[u0:u1 + 1], col), mask_value, K, id_offset,
                precision, None if hint is None else hint[u0:u1], hint_valid, hint_rank, light)

    def after(ws, cnt, u0, u1):          # per-range bookkeeping, on the stream that ran the range's back phase
        if tot is not None:
            tot.add_(cnt)
        if stats is not None:
            o = _score_stats(lib, ws, u1 - u0, I, K, D, dev)
            stat_max.copy_(torch.maximum(stat_max, o[2:3]))
            stat_sum.add_(o)

    if not pipelined:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev) if want_counters else None
        for u0, u1 in ranges:
            nb = lib.chaorec_score_topk_workspace_bytes(u1 - u0, I, K, D)
            _score_call(lib, *args_of(u0, u1), cnt, idx[u0:u1], val[u0:u1], ws, nb)
            after(ws, cnt, u0, u1)
    else:
        cur = torch.cuda.current_stream(dev)
        s_front, s_back = _score_streams(dev)
        wss = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        cnts = [torch.zeros(4, dtype=torch.int32, device=dev) if want_counters else None for _ in range(2)]
        start = torch.cuda.Event()
        start.record(cur)
        s_front.wait_event(start)
        s_back.wait_event(start)
        back_done = [None, None]
        for k, (u0, u1) in enumerate(ranges):
            b = k & 1
            nb = lib.chaorec_score_topk_workspace_bytes(u1 - u0, I, K, D)
            a = args_of(u0, u1)
            with torch.cuda.stream(s_front):
                if back_done[b] is not None:
                    s_front.wait_event(back_done[b])          # (the workspace's previous range has left it)
                _score_call(lib, *a, cnts[b], idx[u0:u1], val[u0:u1], wss[b], nb, phase=SCORE_FRONT)
                front_done = torch.cuda.Event()
                front_done.record(s_front)
            with torch.cuda.stream(s_back):
                s_back.wait_event(front_done)
                _score_call(lib, *a, cnts[b], idx[u0:u1], val[u0:u1], wss[b], nb, phase=SCORE_BACK)
                after(wss[b], cnts[b], u0, u1)
                back_done[b] = torch.cuda.Event()
                back_done[b].record(s_back)
        end = torch.cuda.Event()
        end.record(s_back)
        cur.wait_event(end)              # (the back stream is in order: its last event covers every range)
        for t in wss + [c for c in cnts if c is not None]:
            t.record_stream(s_front)
            t.record_stream(s_back)
    if want_counters:
        counters.copy_(tot, non_blocking=True)
    if stats is not None:
        v = stat_sum.tolist()
        v[2] = int(stat_max.item())
        stats.update(_stats_dict(v), user_chunks=len(ranges), pipelined=bool(pipelined))
    return idx, val


def rank_metrics(rank_idx, row_user, pos_rowptr, pos_items, k_list):
    """[len(k_list), 5] fp64 on the device: precision, recall, ndcg, hit_rate, map averaged over the evaluation rows
    (chaorec_rank_metrics_f64).  rank_idx [U, K] int64 device; the rows' positives as a device CSR."""
    import ctypes
    import numpy as np
    _need_cuda(rank_idx, row_user, pos_rowptr, pos_items)
    if rank_idx.dtype != torch.int64 or not rank_idx.is_contiguous():
        raise TypeError("rank_metrics: rank_idx must be a contiguous int64 tensor")
    k = np.asarray([int(x) for x in k_list], dtype=np.int32)
    disc = 1.0 / np.log(np.arange(int(k.max())) + 2.0)          # the reference's own discount table (metrics.py:31,34)
    n_rows = int(row_user.shape[0])
    lib = _lib.load()
    nbytes = lib.chaorec_rank_metrics_workspace_bytes(n_rows, len(k))
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=rank_idx.device)
    out = torch.empty((len(k), 5), dtype=torch.float64, device=rank_idx.device)
    rc = lib.chaorec_rank_metrics_f64(_ptr(rank_idx), rank_idx.shape[0], rank_idx.shape[1], _ptr(row_user), _ptr(pos_rowptr),
                                      _ptr(pos_items), n_rows, k.ctypes.data_as(ctypes.c_void_p), len(k),
                                      disc.ctypes.data_as(ctypes.c_void_p), _ptr(out), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_rank_metrics_f64")
    return out


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
import os as _os
LINEAR_FORWARD = _os.environ.get("CHAOREC_LINEAR_FORWARD", "bf16x3")


def _linear_fwd_raw(x, weight, bias, act, out=None):
    """act(x W^T + b): the bf16 MFMA pipe (three bf16 planes per fp32 operand) where the reduction is long enough to pay
    for the split, else the f32 MFMA pipe.  `out` may be a column slice of a wider buffer."""
    if LINEAR_FORWARD == "bf16x3" and x.shape[1] >= 64 and x.shape[0] >= 256:
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
    if LINEAR_FORWARD == "bf16x3" and weight.shape[0] >= 64 and gy.shape[0] >= 256:
        return gemm_nn_bf16x3(gy, weight, out=out, accumulate=accumulate)
    return gemm_raw(gy, weight, out=out, accumulate=accumulate)


def _linear_gw_raw(gy, x):
    """The weight gradient gy^T x: a reduction over all rows -- the split-bf16 pipe where its 128-row tile is not half
    padding (out >= 128: 768^2 over 60 k rows 994 -> 744 us, 256^2 128 -> 94 us; the 64-wide layers' gradients are faster
    on the f32 kernel's 64-row tile)."""
    if LINEAR_FORWARD == "bf16x3" and gy.shape[0] >= 4096 and gy.shape[1] >= 128 and x.shape[1] >= 64:
        return gemm_tn_bf16x3(gy, x)
    if LINEAR_FORWARD == "bf16x3" and gy.shape[0] >= 4096 and gy.shape[1] <= 64 and x.shape[1] >= 128:
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
        dual = (MMGCN_DUAL and ax_aug is None and LINEAR_FORWARD == "bf16x3" and n >= 4096 and 64 <= x.shape[1] < 512
                and d1 % 4 == 0 and d2 % 4 == 0 and _dual_ok(x, Wc, Wl))
        ctx.dual = dual
        if dual:
            c, uy = gemm_nt_bf16x3_dual(x, Wc, Wl, bc, bl, 0, 1)
        else:
            uy = _linear_fwd_raw(x, Wl, bl, 1)
        if ax_aug is None:
            if not dual:
                c = _linear_fwd_raw(x, Wc, bc, 0)
            s = csr.propagate_raw(c) if hasattr(csr, "propagate_raw") else spmm_raw(csr, c)
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
            gc = ctx.csr.propagate_t_raw(gs) if hasattr(ctx.csr, "propagate_t_raw") else spmm_raw(ctx.csr.t(), gs)
            gbc = col_sum(gc) if need[3] else None
            gWc = gWl = None
            if need[2] or need[4]:
                gWc, gWl = gemm_tn_bf16x3_dual(gc, gu, x)            # [gc | gu]^T x
            gx = gemm_nn_bf16x3_dual(gc, gu, Wc, Wl) if need[0] else None      # [gc | gu] [Wc; Wl]
            return gx, gid, gWc, gbc, gWl, gbl, gWg, gbg, None, None, None
        gWl = _linear_gw_raw(gu, x) if need[4] else None
        gx = _linear_gx_raw(gu, Wl) if need[0] else None
        if ax_aug is None:
            gc = ctx.csr.propagate_t_raw(gs) if hasattr(ctx.csr, "propagate_t_raw") else spmm_raw(ctx.csr.t(), gs)
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
        if LINEAR_FORWARD == "bf16x3" and xg.shape[1] >= 64 and xg.shape[0] >= 256:
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
            elif LINEAR_FORWARD == "bf16x3" and weight.shape[0] >= 64 and gy_full.shape[0] >= 256:
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
        return spmm_raw(csr, x, z=_f32c(z), beta=1.0)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        return spmm_raw(ctx.csr.t(), gy), gy, None


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


class _SpMMValues(torch.autograd.Function):
    """y = A x where A's values change every call (edge dropout) while its structure is fixed: forward with `val`,
    backward with `val_t` (A^T in the same structure)."""

    @staticmethod
    def forward(ctx, x, structure, val, val_t):
        ctx.structure, ctx.val_t = structure, val_t
        return spmm_raw(structure.with_values(val), x)

    @staticmethod
    def backward(ctx, gy):
        return spmm_raw(ctx.structure.with_values(ctx.val_t), gy.contiguous()), None, None, None


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


# --------------------------------------------------------------------------------------------
# reductions that survive hipGraph replay
# --------------------------------------------------------------------------------------------
def col_sum(x):
    """x.sum(0) for fp32 [M, N], deterministic, in two launches.  torch's own multi-block reductions clear a semaphore
    buffer with a memset, and a memset node inside a captured hipGraph does not replay on this stack:
    inside captured training steps every large reduction goes through here or mean_all()."""
    _need_cuda(x)
    x = _f32c(x)
    M, N = x.shape
    lib = _lib.load()
    nbytes = lib.chaorec_reduce_workspace_bytes(M, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    rc = lib.chaorec_colsum_f32(_ptr(x), M, N, N, _ptr(out), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "chaorec_colsum_f32")
    return out


class _MeanAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        xc = _f32c(x)
        n = xc.numel()
        lib = _lib.load()
        nbytes = lib.chaorec_reduce_workspace_bytes(n, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        rc = lib.chaorec_sum_f32(_ptr(xc), n, 1.0 / max(n, 1), _ptr(out), _ptr(ws), nbytes, _stream())
        _lib.check(rc, "chaorec_sum_f32")
        ctx.shape, ctx.n = x.shape, n
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).expand(ctx.shape)


def mean_all(x):
    """x.mean() over all elements (differentiable), safe inside captured steps (see col_sum)."""
    return _MeanAll.apply(x)
