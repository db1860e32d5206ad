"""CPU oracle for the ChaoRec GCN-propagate + BPR + full-rank hot path (numpy + oracle C).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/chaorec_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package chaorec_amd/ never does.

Each function restates one reference call site; `file:line` is relative to the reference root.
Pinned against reference outputs by tests/test_oracle_golden.py (fixtures in tests/golden/).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libchaorec_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile oracle/chaorec_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "chaorec_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_score_dot.restype = ctypes.c_float
    return _lib


def _p(a, ty):
    if a is None:
        return ctypes.cast(None, ty)
    return a.ctypes.data_as(ty)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------------------------
# graph construction
# ----------------------------------------------------------------------------------------------
def bidirectional_edges(train_edges):
    """Model/LightGCN.py:63-64: edge_index = cat(E^T, E^T[[1,0]]) -> (src, dst) of length 2E.

    Flow is source_to_target: messages x[src] are summed at dst (Model/LightGCN.py:40-43)."""
    e = np.asarray(train_edges, dtype=np.int64)
    src = np.concatenate([e[:, 0], e[:, 1]])
    dst = np.concatenate([e[:, 1], e[:, 0]])
    return src, dst


def sym_norm_weights(src, dst, n_nodes):
    """Model/LightGCN.py:36-38: deg = degree(row); norm = deg^-1/2[row] * deg^-1/2[col], fp32.

    deg^-1/2 of an isolated node is inf but is never indexed by an edge."""
    deg = np.bincount(src, minlength=n_nodes).astype(np.float32)
    with np.errstate(divide="ignore"):
        dinv = (np.float32(1.0) / np.sqrt(deg)).astype(np.float32)
    return (dinv[src] * dinv[dst]).astype(np.float32)


def add_self_loops(src, dst, n_nodes):
    """BasicGCN.py:37: torch_geometric.utils.add_self_loops appends (i,i) for all i AFTER the edges."""
    loop = np.arange(n_nodes, dtype=np.int64)
    return np.concatenate([src, loop]), np.concatenate([dst, loop])


def csr_from_edges(src, dst, w, n_rows):
    """Destination-major CSR whose per-row entry order is the reference's edge order
    (stable sort), so a sequential row sum equals scatter_add_ over the edge list."""
    order = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, _c(src[order], np.int32), _c(w[order], np.float32)


def lightgcn_csr(train_edges, n_nodes):
    src, dst = bidirectional_edges(train_edges)
    w = sym_norm_weights(src, dst, n_nodes)
    return csr_from_edges(src, dst, w, n_nodes)


def basicgcn_csr(train_edges, n_nodes):
    """BasicGCN.py:37-46: self-loops appended, degree counted on the looped list."""
    src, dst = bidirectional_edges(train_edges)
    src, dst = add_self_loops(src, dst, n_nodes)
    w = sym_norm_weights(src, dst, n_nodes)
    return csr_from_edges(src, dst, w, n_nodes)


def user_hist_csr(train_edges, num_user):
    """user -> ascending LOCAL item ids; the reference's user_item_dict (dataload.py:30) holds
    the same sets as python lists of global ids in file order."""
    e = np.asarray(train_edges, dtype=np.int64)
    u, i = e[:, 0], e[:, 1] - num_user
    order = np.lexsort((i, u))
    counts = np.bincount(u, minlength=num_user)
    rowptr = np.zeros(num_user + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, _c(i[order], np.int32)


def user_item_dict_from_edges(train_edges):
    """user_item_dict = train edges grouped by user, file order, keys ascending."""
    d = {}
    for u, i in np.asarray(train_edges).tolist():
        d.setdefault(u, []).append(i)
    return d


# ----------------------------------------------------------------------------------------------
# C-backed primitives
# ----------------------------------------------------------------------------------------------
def spmm(csr, x, alpha=1.0, z=None, beta=0.0, acc=None, acc_init=None, acc_w=0.0, want_y=True):
    rowptr, col, val = csr
    x = _c(x, np.float32)
    n_rows, D = len(rowptr) - 1, x.shape[1]
    y = np.empty((n_rows, D), np.float32) if want_y else None
    z = None if z is None else _c(z, np.float32)
    acc_init = None if acc_init is None else _c(acc_init, np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr, _i64p), _p(col, _i32p), _p(val, _f32p), _p(x, _f32p),
                              _p(y, _f32p), ctypes.c_int64(n_rows), ctypes.c_int32(D),
                              ctypes.c_float(alpha), _p(z, _f32p), ctypes.c_float(beta),
                              _p(acc, _f32p), _p(acc_init, _f32p), ctypes.c_float(acc_w))
    return y


def scatter_edges(src, dst, w, x, n_rows):
    x = _c(x, np.float32)
    out = np.zeros((n_rows, x.shape[1]), np.float32)
    src, dst, w = _c(src, np.int64), _c(dst, np.int64), _c(w, np.float32)
    lib().oracle_scatter_edges_f32(_p(src, _i64p), _p(dst, _i64p), _p(w, _f32p), _p(x, _f32p),
                                   _p(out, _f32p), ctypes.c_int64(len(src)), ctypes.c_int32(x.shape[1]))
    return out


def bpr_fwd(tab_u, tab_i, users, pos, neg, variant, reg_weight):
    tab_u, tab_i = _c(tab_u, np.float32), _c(tab_i, np.float32)
    users, pos, neg = _c(users, np.int64), _c(pos, np.int64), _c(neg, np.int64)
    B, D = len(users), tab_u.shape[1]
    out = np.zeros(3, np.float64)
    coef = np.zeros(B, np.float64)
    lib().oracle_bpr_fwd_f32(_p(tab_u, _f32p), _p(tab_i, _f32p), _p(users, _i64p), _p(pos, _i64p),
                             _p(neg, _i64p), ctypes.c_int32(B), ctypes.c_int32(D),
                             ctypes.c_int32(variant), ctypes.c_float(reg_weight), _p(out, _f64p),
                             _p(coef, _f64p))
    return out, coef


def bpr_bwd(tab_u, tab_i, users, pos, neg, coef, reg_weight, grad_out=1.0):
    tab_u, tab_i = _c(tab_u, np.float32), _c(tab_i, np.float32)
    users, pos, neg = _c(users, np.int64), _c(pos, np.int64), _c(neg, np.int64)
    B, D = len(users), tab_u.shape[1]
    g_u = np.zeros(tab_u.shape, np.float64)
    g_i = np.zeros(tab_i.shape, np.float64)
    coef = _c(coef, np.float64)
    lib().oracle_bpr_bwd_f32(_p(tab_u, _f32p), _p(tab_i, _f32p), _p(users, _i64p), _p(pos, _i64p),
                             _p(neg, _i64p), ctypes.c_int32(B), ctypes.c_int32(D), _p(coef, _f64p),
                             ctypes.c_float(reg_weight), ctypes.c_double(grad_out), _p(g_u, _f64p),
                             _p(g_i, _f64p))
    return g_u, g_i


def bpr_bwd_ordered(tab_u, tab_i, users, pos, neg, coef, reg_weight, grad_out=1.0, item_offset=None):
    """The float, role-major, batch-ordered index_add (oracle_bpr_bwd_ordered_f32): bit-exact twin of the product's ordered
    backward launch.  item_offset (tab_i None): items are rows item_offset.. of the ONE table tab_u -> one gradient array."""
    tab_u = _c(tab_u, np.float32)
    users, pos, neg = _c(users, np.int64), _c(pos, np.int64), _c(neg, np.int64)
    B, D = len(users), tab_u.shape[1]
    coef = _c(coef, np.float32)
    g_u = np.zeros(tab_u.shape, np.float32)
    if tab_i is None:
        off = int(item_offset) * D * 4
        ti = ctypes.cast(ctypes.c_void_p(tab_u.ctypes.data + off), _f32p)
        gi = ctypes.cast(ctypes.c_void_p(g_u.ctypes.data + off), _f32p)
        g_i = None
    else:
        tab_i = _c(tab_i, np.float32)
        g_i = np.zeros(tab_i.shape, np.float32)
        ti, gi = _p(tab_i, _f32p), _p(g_i, _f32p)
    lib().oracle_bpr_bwd_ordered_f32(_p(tab_u, _f32p), ti, _p(users, _i64p), _p(pos, _i64p), _p(neg, _i64p), ctypes.c_int32(B),
                                     ctypes.c_int32(D), _p(coef, _f32p), ctypes.c_float(reg_weight), ctypes.c_float(grad_out),
                                     _p(g_u, _f32p), gi)
    return g_u, g_i


SECOND_DRAW_SALT = 0x9E3779B97F4A7C15      # the stream of the sample's second draw (dataload.py:81-84), chaorec_hip.h


def sample_negatives(hist, users, num_item, seed, step, id_offset, second=False):
    if second:
        seed = (int(seed) ^ SECOND_DRAW_SALT) & 0xFFFFFFFFFFFFFFFF
    rowptr, col = hist
    users = _c(users, np.int64)
    out = np.empty(len(users), np.int64)
    lib().oracle_sample_negatives(_p(rowptr, _i64p), _p(col, _i32p), _p(users, _i64p),
                                  ctypes.c_int32(len(users)), ctypes.c_int32(num_item),
                                  ctypes.c_uint64(seed), ctypes.c_uint64(step),
                                  ctypes.c_int64(id_offset), _p(out, _i64p))
    return out


def score_topk(user_emb, item_emb, hist, mask_value, K, id_offset):
    user_emb, item_emb = _c(user_emb, np.float32), _c(item_emb, np.float32)
    U, D = user_emb.shape
    idx = np.empty((U, K), np.int64)
    val = np.empty((U, K), np.float32)
    rowptr, col = (None, None) if hist is None else hist
    lib().oracle_score_topk_f32(_p(user_emb, _f32p), _p(item_emb, _f32p), ctypes.c_int64(U),
                                ctypes.c_int64(item_emb.shape[0]), ctypes.c_int32(D),
                                _p(rowptr, _i64p), _p(col, _i32p), ctypes.c_float(mask_value),
                                ctypes.c_int32(K), ctypes.c_int64(id_offset), _p(idx, _i64p),
                                _p(val, _f32p))
    return idx, val


def gemm(A, B, bias=None, transA=False, transB=False, C=None, act=0):
    A, B = _c(A, np.float32), _c(B, np.float32)
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[0] if transB else B.shape[1]
    accumulate = C is not None
    if C is None:
        C = np.empty((M, N), np.float32)
    bias = None if bias is None else _c(bias, np.float32)
    lib().oracle_gemm_f32(_p(A, _f32p), _p(B, _f32p), _p(C, _f32p), _p(bias, _f32p),
                          ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K),
                          ctypes.c_int64(A.shape[1]), ctypes.c_int64(B.shape[1]), ctypes.c_int64(N),
                          ctypes.c_int32(int(transA)), ctypes.c_int32(int(transB)),
                          ctypes.c_int32(int(accumulate)), ctypes.c_int32(act))
    return C


def adam_step(p, g, m, v, lr, b1, b2, eps, wd, step):
    lib().oracle_adam_step_f32(_p(p, _f32p), _p(g, _f32p), _p(m, _f32p), _p(v, _f32p),
                               ctypes.c_int64(p.size), ctypes.c_float(lr), ctypes.c_float(b1),
                               ctypes.c_float(b2), ctypes.c_float(eps), ctypes.c_float(wd),
                               ctypes.c_int32(step))


# ----------------------------------------------------------------------------------------------
# model-level restatements
# ----------------------------------------------------------------------------------------------
def lightgcn_forward(x0, csr, n_layers):
    """Model/LightGCN.py:76-95: x_{l+1} = conv(x_l); result = sum_l (1/(L+1)) x_l accumulated in
    layer order into zeros.  Returns (result, [x_0..x_L])."""
    w = np.float32(1.0 / (n_layers + 1))
    x0 = _c(x0, np.float32)
    layers = [x0]
    final = np.zeros_like(x0)
    final = final + w * x0
    x = x0
    for _ in range(n_layers):
        x = spmm(csr, x)
        layers.append(x)
        final = final + w * x
    return final.astype(np.float32), layers


def lightgcn_loss(x0, csr, n_layers, num_user, users, pos_local, neg_local, reg_weight):
    """Model/LightGCN.py:123-135 (items already local).  Returns (out[3], dL/dx0 float64)."""
    final, _ = lightgcn_forward(x0, csr, n_layers)
    out, coef = bpr_fwd(final[:num_user], final[num_user:], users, pos_local, neg_local, 0, reg_weight)
    g_u, g_i = bpr_bwd(final[:num_user], final[num_user:], users, pos_local, neg_local, coef, reg_weight)
    G = np.concatenate([g_u, g_i], 0)
    # backward of the layer mean + symmetric propagate, in float64 via scipy (accurate side)
    import scipy.sparse as sp
    rowptr, col, val = csr
    A = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(len(rowptr) - 1,) * 2)
    w = 1.0 / (n_layers + 1)
    g = w * G
    for _ in range(n_layers):
        g = A.T @ g + w * G
    return out, g


def ngcf_forward(x0, w1s, w2s, train_edges, n_nodes, keep_masks=None):
    """Model/NGCF.py:38-84 + :114-127, EDGE-WISE as the reference evaluates it: per conv call drop edges
    (keep_masks[l], bool over the 2E bidirectional edges; None = no dropout), append self loops, recount degree(row),
    m_e = norm_e * (W1 x[row_e] + W2 (x[row_e] * x[col_e])), scatter-add at col in edge order, leaky_relu(0.2);
    result = x_0 + x_1 + ... (torch.sum over the stack).  fp32 throughout."""
    src0, dst0 = bidirectional_edges(train_edges)
    x = _c(x0, np.float32)
    out = x.copy()
    for l, (w1, w2) in enumerate(zip(w1s, w2s)):
        src, dst = src0, dst0
        if keep_masks is not None and len(keep_masks):
            k = np.asarray(keep_masks[l], dtype=bool)
            src, dst = src0[k], dst0[k]
        src, dst = add_self_loops(src, dst, n_nodes)
        norm = sym_norm_weights(src, dst, n_nodes)
        xj, xi = x[src], x[dst]
        msg = (norm[:, None] * (xj @ w1.T + (xj * xi) @ w2.T)).astype(np.float32)
        agg = np.zeros_like(x)
        np.add.at(agg, dst, msg)
        x = np.where(agg > 0, agg, np.float32(0.2) * agg).astype(np.float32)
        out = (out + x).astype(np.float32)
    return out


def row_cosine_scale(y, e, eps=1e-8):
    """Model/LayerGCN.py:125-127 in fp64: w = cosine_similarity(y, e, dim=-1) (each norm clamped at eps, torch's
    definition), out = w[:, None] * y.  Returns (out, w)."""
    y, e = np.asarray(y, np.float64), np.asarray(e, np.float64)
    a = np.maximum(np.linalg.norm(y, axis=1), eps)
    b = np.maximum(np.linalg.norm(e, axis=1), eps)
    w = (y * e).sum(1) / (a * b)
    return w[:, None] * y, w


def mix64(z):
    """splitmix64 finaliser on uint64 arrays (chaorec_amd/csrc/common.h:mix64)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def edge_dropout_keep(nnz, p, seed, step, salt):
    """The generator of chaorec_edge_dropout_norm (include/chaorec_hip.h): u_k = 2^-24 * (h_k >> 40), keep = u_k >= p.
    Self loops are forced to 1 by the caller."""
    k = np.arange(nnz, dtype=np.uint64)
    h = mix64(np.uint64(seed) ^ mix64(np.uint64(step) ^ mix64((np.uint64(salt) << np.uint64(48)) ^ k)))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return u >= np.float32(p)


def edge_dropout_norm(entry_row, col, transpose_entry, n_nodes, keep):
    """Values of the dropped + renormalised graph and of its transpose (chaorec_edge_dropout_norm), keep already
    holding 1 at the self loops.  deg counts kept entries by SOURCE (= col of the destination-major CSR)."""
    keep = np.asarray(keep, dtype=bool)
    deg = np.bincount(np.asarray(col)[keep], minlength=n_nodes).astype(np.float32)
    with np.errstate(divide="ignore"):
        dinv = (np.float32(1.0) / np.sqrt(deg)).astype(np.float32)
    w = (dinv[col] * dinv[entry_row]).astype(np.float32)
    zero = np.float32(0.0)
    return np.where(keep, w, zero).astype(np.float32), np.where(keep[transpose_entry], w, zero).astype(np.float32)


def race_uniform(n, seed, step):
    """u_e and the low key bits of chaorec_weighted_sample_keep's exponential race."""
    e = np.arange(n, dtype=np.uint64)
    h = mix64(np.uint64(seed) ^ mix64(np.uint64(step) ^ mix64((np.uint64(0x5A3B) << np.uint64(48)) ^ e)))
    u = ((h >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return u, (h & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def race_keys(weights, seed, step, ids=None):
    """uint64 keys of chaorec_weighted_sample_keep / _keys (chaorec_amd/csrc/graph_dropout.hip:race_key_of): high word
    = fp32 bits of |log u| / w, low word = hash bits; entry j numbered ids[j] (None: j); w <= 0 -> all ones."""
    w = np.asarray(weights, dtype=np.float32)
    e = (np.arange(len(w)) if ids is None else np.asarray(ids)).astype(np.uint64)
    h = mix64(np.uint64(seed) ^ mix64(np.uint64(step) ^ mix64((np.uint64(0x5A3B) << np.uint64(48)) ^ e)))
    u = ((h >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    with np.errstate(divide="ignore", invalid="ignore"):
        key = (np.abs(np.log(u)).astype(np.float32) / w).astype(np.float32)
    out = (key.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (h & np.uint64(0xFFFFFFFF))
    return np.where(w > 0, out, np.uint64(0xFFFFFFFFFFFFFFFF))


def gene_ranklist(result, num_user, num_item, hist, mask_value=1e-6, topk=50):
    """Model/LightGCN.py:137-162 -> int64 [U, topk] of GLOBAL item ids."""
    idx, val = score_topk(result[:num_user], result[num_user:num_user + num_item], hist,
                          mask_value, topk, num_user)
    return idx, val


# metrics.py:13-57 + utils.py:112-139, restated as the same per-user python loops
def gene_metrics(val_data, rank_list, k_list):
    names = ("precision", "recall", "ndcg", "hit_rate", "map")
    m = {k: {n: 0.0 for n in names} for k in k_list}
    for data in val_data:
        user, pos = data[0], list(data[1:])
        ranked = [int(v) for v in rank_list[user]]
        pos_set = set(int(p) for p in pos)
        for k in k_list:
            top = ranked[:k]
            inter = len(set(top) & pos_set)
            m[k]["precision"] += inter / k                                   # metrics.py:13-16
            m[k]["recall"] += 0 if len(pos) == 0 else inter / len(pos)       # metrics.py:19-23
            if pos:                                                          # metrics.py:26-40
                idcg = sum(1.0 / np.log(i + 2) for i in range(min(len(pos), k)))
                dcg = sum(1.0 / np.log(i + 2) for i, it in enumerate(top) if it in pos_set)
                m[k]["ndcg"] += dcg / idcg
            m[k]["hit_rate"] += int(inter > 0)                               # metrics.py:43-45
            if pos:                                                          # metrics.py:48-57
                hits, sc = 0, 0.0
                for i, it in enumerate(top):
                    if it in pos_set:
                        hits += 1
                        sc += hits / (i + 1)
                m[k]["map"] += sc / len(pos)
    n = len(val_data)
    for k in k_list:
        for nme in names:
            m[k][nme] /= n
    return m
