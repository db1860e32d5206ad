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
    """(list int32 [number of rows above the threshold], counters int32 [4] zero, threshold) for spmm_rowlist_raw's long_rows."""
    t = ROWLIST_LONG_T if threshold is None else int(threshold)
    n_long = int(((csr.rowptr[1:] - csr.rowptr[:-1]) > t).sum().item())
    dev = csr.rowptr.device
    return torch.zeros(max(n_long, 1), dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev), t


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
# The BPR backward launches of the autograd nodes below: "ordered" (default) = one owner wave per gradient row, contributions
# added in a fixed order, no atomics -- a training step is then reproducible bit for bit from run to run; CHAOREC_BPR_ORDERED=0
# = fp32 atomic row adds (order-dependent from three addends per element on: tools/stream_stress.py).
# (2: the fused LightGCN steps' in-launch row adds go through the ordered launch as well -- read by the library, csrc/bpr.hip)
BPR_ORDERED = os.environ.get("CHAOREC_BPR_ORDERED", "1") != "0"


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
        lib = _lib.load()
        bwd = lib.chaorec_bpr_bwd_ordered_f32 if BPR_ORDERED else lib.chaorec_bpr_bwd_f32
        rc = bwd(_ptr(tab_u), pi, _ptr(users), _ptr(pos), _ptr(neg), B, D, _ptr(coef), ctx.reg_weight, _ptr(go), _ptr(g_u), pgi,
                 _stream())
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
            multi_bwd = lib.chaorec_bpr_multi_bwd_ordered_f32 if BPR_ORDERED else lib.chaorec_bpr_multi_bwd_f32
            rc = multi_bwd(_ptr(tab_u), _ptr(users), T, arr(tabs), arr(ids[0::2]), arr(ids[1::2]), B, D,
                           _ptr(coef), _ptr(wvec), _ptr(g), _ptr(g_u), arr(g_is), srows, souts, _stream())
            _lib.check(rc, "chaorec_bpr_multi_bwd_f32")
            for g_i in g_is:
                grads += [g_i, None, None]
            return (g_u, None, None, None, None, None, *grads)
        gvec = (g * wvec).contiguous()                   # d total / d loss_k, on the device
        for k in range(T):
            g_i = flat[offs[k + 1]:offs[k + 2]].view_as(tabs[k])
            bwd1 = lib.chaorec_bpr_bwd_ordered_f32 if BPR_ORDERED else lib.chaorec_bpr_bwd_f32
            rc = bwd1(_ptr(tab_u), _ptr(tabs[k]), _ptr(users), _ptr(ids[2 * k]), _ptr(ids[2 * k + 1]), B,
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
    CHAOREC_SCORE_WS_LIMIT (24 GiB) is cut into user ranges of equal length -- the users are independent --, one range after
    the other on the caller's stream.  (Round 5 also ran them PIPELINED -- range k's back phase on a second stream beside range
    k + 1's front phase, two workspaces in flight --: 634 ms against 630 ms serial and 628 ms in one piece at the config-5 shard,
    every kernel of the call fills the chip on its own; that code is profiles/r05_exp_score_pipeline.patch)"""
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
    budget = limit
    per = max(4096, (U * budget // nbytes) // 4096 * 4096)
    while per > 4096 and lib.chaorec_score_topk_workspace_bytes(per, I, K, D) > budget:
        per -= 4096
    n_ranges = (U + per - 1) // per
    per = min(per, ((U + n_ranges - 1) // n_ranges + 4095) // 4096 * 4096)     # ranges of equal length, not a full one + a rest
    ranges = [(u0, min(U, u0 + per)) for u0 in range(0, U, per)]
    ws_bytes = lib.chaorec_score_topk_workspace_bytes(per, I, K, D)
    want_counters = counters is not None and precision == 0
    tot = torch.zeros(4, dtype=torch.int32, device=dev) if want_counters else None
    stat_sum = torch.zeros(10, dtype=torch.int64, device=dev) if stats is not None else None
    stat_max = torch.zeros(1, dtype=torch.int64, device=dev) if stats is not None else None

    def args_of(u0, u1):
        return (user_emb[u0:u1], item_emb, None if hist is None else (rowptr[u0:u1 + 1], col), mask_value, K, id_offset,
                precision, None if hint is None else hint[u0:u1], hint_valid, hint_rank, light)

    def after(ws, cnt, u0, u1):          # per-range bookkeeping
        if tot is not None:
            tot.add_(cnt)
        if stats is not None:
            o = _score_stats(lib, ws, u1 - u0, I, K, D, dev)
            stat_max.copy_(torch.maximum(stat_max, o[2:3]))
            stat_sum.add_(o)

    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev) if want_counters else None
    for u0, u1 in ranges:
        nb = lib.chaorec_score_topk_workspace_bytes(u1 - u0, I, K, D)
        _score_call(lib, *args_of(u0, u1), cnt, idx[u0:u1], val[u0:u1], ws, nb)
        after(ws, cnt, u0, u1)
    if want_counters:
        counters.copy_(tot, non_blocking=True)
    if stats is not None:
        v = stat_sum.tolist()
        v[2] = int(stat_max.item())
        stats.update(_stats_dict(v), user_chunks=len(ranges), pipelined=False)
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


# (ops.linear's pipe: "bf16x3" = split-bf16 MFMA, "f32" = the f32 MFMA chain.  Lives on this facade because callers and tests
#  switch it here; ops_dense / ops_extra read it through the facade at call time.)
LINEAR_FORWARD = os.environ.get("CHAOREC_LINEAR_FORWARD", "bf16x3")


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


# ---- the dense products (GEMM, linear, MMGCN's layer) and the optimizer / model-family wrappers live in their own modules since
# ---- round 5 (ops.py had grown to 1 700 lines); everything is still reachable as chaorec_amd.ops.<name>, resolved
# ---- on first use (PEP 562) because those modules import the helpers of this one.
_MOVED = {
    "ops_dense": (
        "gemm_raw", "gemm_nt_bf16x3", "gemm_tn_bf16x3", "gemm_nn_bf16x3", "_dual_ok", "gemm_nt_bf16x3_dual",
        "gemm_nn_bf16x3_dual", "gemm_tn_bf16x3_dual", "_linear_fwd_raw", "_leaky_bwd_raw", "_linear_gx_raw",
        "_linear_gw_raw", "_Linear", "linear", "leaky_cat_add", "leaky_split_bwd", "MMGCN_LAYER", "MMGCN_DUAL",
        "_MMGCNLayer", "mmgcn_layer", "_NormalizeRows", "normalize_rows",
    ),
    "ops_extra": (
        "adam_step", "adam_bias_table", "unique_rows", "adam_lowrank_strips", "adam_lowrank", "_LinearRows",
        "linear_rows", "adam_multi", "adam_multi_max", "_SpMMAdd", "spmm_add", "edge_dropout_norm", "_SpMMValues",
        "spmm_values", "edge_dot_raw", "_EdgeDot", "edge_dot", "_NGCFLayer", "ngcf_layer", "weighted_sample_keep", "weighted_sample_keys",
        "_RowCosineScale", "row_cosine_scale",
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
