"""Host-side graph setup: edge lists -> destination-major CSR resident in HBM.

Done once per model (or once per epoch for FREEDOM's pruned graph); the per-batch work is all
in the HIP kernels.  The normalisation values are computed with the same torch CPU ops the
reference uses, so they are bit-identical to the reference's (torch's pow(-0.5) etc.).

HBM layout of a graph: rowptr int64 [n_rows+1], col int32 [nnz], val fp32 [nnz]; row = the
DESTINATION node, entries in the reference's edge order (stable sort by destination), which is
what makes the ordered SpMM reproduce scatter_add_'s accumulation order.
"""
import numpy as np
import torch


import os as _os

SCHEDULE_ON_DEVICE = _os.environ.get("CHAOREC_SCHEDULE_ON_DEVICE", "1") == "1"
K_INLINE, DESC_DWORDS = 6, 16            # csrc/spmm.hip: (col, val) pairs inside a row descriptor, dwords per descriptor


def schedule_tensors(rowptr, col, val, n_rows, g, long_t):
    """chaorec_spmm_build_schedule (csrc/spmm.hip) as tensor operations on the device the CSR lives on -- the same int32
    words (tests/test_host_logic.py compares them with the host builder's): g rows per wave, 4 waves per workgroup.
      * rows sorted by degree (stable, longest first); a row above `long_t` entries leads a group of its own, with the g - 1
        SHORTEST rows left as company; all other rows fill the remaining groups g at a time;
      * groups sorted by their heaviest row; a group with a long row leads a workgroup of its own, with the three LIGHTEST
        groups left; the others fill the remaining workgroups four at a time;
      * per slot a 64-byte descriptor: row (-1: empty), degree | long-workgroup flag, first entry (lo, hi), the first six
        (col, val) pairs."""
    dev = rowptr.device
    i64 = dict(dtype=torch.int64, device=dev)
    n_rows, g, long_t = int(n_rows), int(g), int(long_t)
    groups = (n_rows + g - 1) // g
    nb = (groups + 3) // 4
    deg = (rowptr[1:n_rows + 1] - rowptr[:n_rows]).to(torch.int64)
    rows_sorted = torch.sort(deg, descending=True, stable=True).indices
    n_long = min(int((deg > long_t).sum().item()), groups) if g > 1 else 0
    slot_row = torch.full((groups * g,), -1, **i64)
    # long groups: slot 0 = the long row, slots 1.. = rows from the short end, while any are left
    if n_long:
        slot_row[torch.arange(n_long, **i64) * g] = rows_sorted[:n_long]
    taken = min(n_long * (g - 1), n_rows - n_long) if g > 1 else 0
    if taken:
        k = torch.arange(taken, **i64)
        slot_row[(k // (g - 1)) * g + 1 + k % (g - 1)] = rows_sorted[n_rows - 1 - k]
    rest = n_rows - n_long - taken
    if rest:
        p_ = torch.arange(rest, **i64)
        slot_row[(n_long + p_ // g) * g + p_ % g] = rows_sorted[n_long + p_]
    deg_pad = torch.cat((deg, deg.new_zeros(1)))                       # (index -1 -> degree 0)
    heavy = deg_pad[slot_row].view(groups, g).max(dim=1).values
    order = torch.sort(heavy, descending=True, stable=True).indices
    n_long_groups = min(int((heavy > long_t).sum().item()), nb) if g > 1 else 0
    block_group = torch.full((nb * 4,), -1, **i64)
    if n_long_groups:
        block_group[torch.arange(n_long_groups, **i64) * 4] = order[:n_long_groups]
    taken2 = min(n_long_groups * 3, groups - n_long_groups)
    if taken2:
        k = torch.arange(taken2, **i64)
        block_group[(k // 3) * 4 + 1 + k % 3] = order[groups - 1 - k]
    rest2 = groups - n_long_groups - taken2
    if rest2:
        p_ = torch.arange(rest2, **i64)
        block_group[(n_long_groups + p_ // 4) * 4 + p_ % 4] = order[n_long_groups + p_]
    heavy_pad = torch.cat((heavy, heavy.new_zeros(1)))
    any_long = (heavy_pad[block_group].view(nb, 4) > long_t).any(dim=1) if g > 1 else torch.zeros(nb, dtype=torch.bool, device=dev)
    # descriptor of slot (workgroup b, wave j, lane group s)
    slot_row_pad = torch.cat((slot_row.view(groups, g), slot_row.new_full((1, g), -1)))      # (group -1 -> all empty)
    r = slot_row_pad[block_group].reshape(-1)                                                # [nb * 4 * g]
    flag = any_long.repeat_interleave(4 * g)
    ok = r >= 0
    rr = r.clamp(min=0)
    e0 = torch.where(ok, rowptr[rr].to(torch.int64), torch.zeros_like(rr))
    dg = torch.where(ok, deg[rr], torch.zeros_like(rr))
    if int(dg.max().item()) > 0x7fffffff:
        raise ValueError("schedule: a row has more than 2^31 - 1 entries")
    out = torch.zeros((r.numel(), DESC_DWORDS), dtype=torch.int32, device=dev)
    sign = torch.tensor(-0x80000000, **i64)
    out[:, 0] = torch.where(ok, r, torch.full_like(r, -1)).to(torch.int32)
    d1 = dg + torch.where(flag, sign, torch.zeros_like(dg))           # deg | 0x80000000 as a signed 32-bit value
    out[:, 1] = d1.to(torch.int32)
    lo = e0 & 0xffffffff
    out[:, 2] = torch.where(lo >= 0x80000000, lo - 0x100000000, lo).to(torch.int32)
    out[:, 3] = (e0 >> 32).to(torch.int32)
    nnz = int(col.numel())
    vbits = val.view(torch.int32)
    for q in range(K_INLINE):
        has = ok & (dg > q)
        at = (e0 + q).clamp(max=max(nnz - 1, 0))
        if nnz:
            out[:, 4 + q] = torch.where(has, col[at].to(torch.int32), torch.zeros_like(out[:, 0]))
            out[:, 4 + K_INLINE + q] = torch.where(has, vbits[at], torch.zeros_like(out[:, 0]))
    return out.reshape(-1)


class CSR:
    """Destination-major CSR (+ optional transpose for the backward pass)."""

    def __init__(self, rowptr, col, val, n_rows, n_cols, symmetric=False, transpose=None):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.symmetric = bool(symmetric)
        self._t = transpose
        self._orders = {}
        self._order_dims = {}

    @property
    def nnz(self):
        return int(self.col.numel())

    def to(self, device):
        out = CSR(self.rowptr.to(device), self.col.to(device), self.val.to(device), self.n_rows,
                  self.n_cols, self.symmetric)
        t = self._t
        if t is not None:        # t() links a matrix and its transpose to each other: move the pair, keep the link
            out._t = CSR(t.rowptr.to(device), t.col.to(device), t.val.to(device), t.n_rows, t.n_cols, t.symmetric)
            out._t._t = out
        return out

    def schedule(self, D):
        """The SpMM kernel's schedule for feature width D (int32 tensor on the graph's device, cached): row
        descriptors in workgroup order, built on the host by chaorec_spmm_build_schedule -- longest rows first,
        one heavy group per workgroup, the first (col,val) pairs of every row inline."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        g = lib.chaorec_spmm_rows_per_wave(int(D))
        if g <= 0 or self.n_rows == 0:
            return None
        if g not in self._orders:
            if self.rowptr.is_cuda and SCHEDULE_ON_DEVICE:
                # the same schedule, word for word, from tensor operations ON THE DEVICE (schedule_tensors): the host builder
                # walks a host copy of the CSR -- 5.8 s of the 15.5 s a rank spent building BASELINE configs[4] whole
                self._orders[g] = schedule_tensors(self.rowptr, self.col, self.val, self.n_rows, g,
                                                   int(lib.chaorec_spmm_long_threshold()))
            else:
                rowptr = self.rowptr.cpu().contiguous()
                col = self.col.cpu().contiguous()
                val = self.val.cpu().contiguous()
                n = lib.chaorec_spmm_schedule_len(self.n_rows, int(D))
                out = torch.empty(n, dtype=torch.int32)
                rc = lib.chaorec_spmm_build_schedule(ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                                     ctypes.c_void_p(val.data_ptr()), self.n_rows, int(D),
                                                     ctypes.c_void_p(out.data_ptr()), n)
                _lib.check(rc, "chaorec_spmm_build_schedule")
                self._orders[g] = out.to(self.rowptr.device)
            self._order_dims[g] = int(D)
        return self._orders[g]

    def update_from(self, other):
        """Overwrite this graph IN PLACE with `other` (same shape and entry count) and refresh the SpMM schedules
        already built for it: the device addresses stay the same, so a captured hipGraph that reads this CSR sees
        the new graph on its next replay (FREEDOM re-prunes its adjacency every epoch).  False if the sizes differ."""
        if (self.n_rows, self.n_cols, self.nnz) != (other.n_rows, other.n_cols, other.nnz) or self._t is not None:
            return False
        self.rowptr.copy_(other.rowptr)
        self.col.copy_(other.col)
        self.val.copy_(other.val)
        self.symmetric = other.symmetric
        for g, D in self._order_dims.items():
            self._orders[g].copy_(other.schedule(D))
        return True

    def t(self):
        """CSR of A^T (the backward operator).  A symmetric graph is its own transpose."""
        if self.symmetric:
            return self
        if self._t is None:
            self._t = transpose_csr(self)
            self._t._t = self
        return self._t


def coo_to_csr(dst, src, val, n_rows, n_cols, symmetric=False, device=None):
    """Stable counting sort by destination: per-row entry order == edge-list order.  Runs on the CPU unless `device`
    names where to do it (the inputs then live there already: config 5's 4e8 entries are sorted on the GPU)."""
    device = torch.device("cpu") if device is None else torch.device(device)
    dst = torch.as_tensor(dst, dtype=torch.int64).to(device)
    src = torch.as_tensor(src, dtype=torch.int64).to(device)
    val = torch.as_tensor(val, dtype=torch.float32).to(device)
    if device.type != "cpu":
        # (a stable sort of (dst, position) pairs: sort() carries the permutation, no 8-byte argsort + two gathers of
        #  64-bit operands)
        counts = torch.bincount(dst, minlength=n_rows)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        del counts
        _, order = torch.sort(dst, stable=True)
        del dst
        col = src.to(torch.int32)[order].contiguous()
        del src
        return CSR(rowptr, col, val[order].contiguous(), n_rows, n_cols, symmetric)
    order = torch.argsort(dst, stable=True)
    counts = torch.bincount(dst, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return CSR(rowptr, src[order].to(torch.int32).contiguous(), val[order].contiguous(), n_rows, n_cols,
               symmetric)


def transpose_csr(csr):
    rowptr = csr.rowptr.cpu()
    rows = torch.repeat_interleave(torch.arange(csr.n_rows, dtype=torch.int64), rowptr[1:] - rowptr[:-1])
    out = coo_to_csr(csr.col.cpu().to(torch.int64), rows, csr.val.cpu(), csr.n_cols, csr.n_rows)
    return out.to(csr.col.device)


def bidirectional_edge_index(edge_index):
    """Model/LightGCN.py:63-64: [E,2] int32 ndarray -> LongTensor [2, 2E] = cat(E^T, E^T[[1,0]])."""
    e = torch.as_tensor(np.asarray(edge_index)).t().contiguous()
    return torch.cat((e, e[[1, 0]]), dim=1).long()


def _lightgcn_csr_device(edges, n_nodes):
    """lightgcn_csr() for an edge list that lives on the GPU (int32/int64 [E, 2] tensor): the same entries in the same
    order with the same values -- the degree count is an integer bincount, d^-1/2 is taken by the same CPU pow() as below
    (n_nodes values), the per-entry product is one IEEE multiply wherever it runs -- laid out by a device sort instead
    of a host argsort over 2 E keys (minutes at 4e8)."""
    dev = edges.device
    u, i = edges[:, 0].to(torch.int64), edges[:, 1].to(torch.int64)
    deg = torch.bincount(torch.cat([u, i]), minlength=n_nodes)        # degree(row) over the bidirectional list
    dis = deg.to(torch.float32).cpu().pow(-0.5).to(dev)
    del deg
    row = torch.cat([u, i])            # sources: forward edges u -> i in file order, then the reversed edges
    col = torch.cat([i, u])            # destinations
    del u, i
    norm = dis[row] * dis[col]
    return coo_to_csr(col, row, norm, n_nodes, n_nodes, symmetric=True, device=dev)


def lightgcn_csr(edge_index, n_nodes):
    """LightGCNConv.forward (Model/LightGCN.py:28-40): deg = degree(row); norm = d^-1/2[row] d^-1/2[col];
    out[col] += norm * x[row].  No self loops.  Symmetric by construction."""
    if torch.is_tensor(edge_index) and edge_index.is_cuda:
        return _lightgcn_csr_device(edge_index, n_nodes)
    ei = bidirectional_edge_index(edge_index)
    row, col = ei[0], ei[1]
    deg = torch.zeros(n_nodes, dtype=torch.float32).scatter_add_(0, row, torch.ones(row.numel()))
    dis = deg.pow(-0.5)
    norm = dis[row] * dis[col]
    return coo_to_csr(col, row, norm, n_nodes, n_nodes, symmetric=True)


def basicgcn_csr(edge_index, n_nodes):
    """BasicGCN.forward (BasicGCN.py:33-48): add_self_loops appended AFTER the edges, degree counted on
    the looped list, same symmetric normalisation."""
    ei = bidirectional_edge_index(edge_index)
    loop = torch.arange(n_nodes, dtype=torch.int64)
    row = torch.cat([ei[0], loop])
    col = torch.cat([ei[1], loop])
    deg = torch.zeros(n_nodes, dtype=torch.float32).scatter_add_(0, row, torch.ones(row.numel()))
    dis = deg.pow(-0.5)
    norm = dis[row] * dis[col]
    return coo_to_csr(col, row, norm, n_nodes, n_nodes, symmetric=True)


def user_hist_csr(user_item_dict, num_user):
    """user_item_dict {user: [global item ids]} -> (rowptr int64 [U+1], col int32 ascending LOCAL ids).

    This is the mask of gene_ranklist (Model/LightGCN.py:150-152) and the rejection set of the
    sampler (dataload.py:76-79)."""
    counts = np.zeros(num_user, dtype=np.int64)
    for u, items in user_item_dict.items():
        counts[int(u)] = len(set(int(i) for i in items))
    rowptr = np.zeros(num_user + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    for u, items in user_item_dict.items():
        u = int(u)
        loc = np.unique(np.asarray([int(i) for i in items], dtype=np.int64)) - num_user
        col[rowptr[u]:rowptr[u + 1]] = loc
    return torch.from_numpy(rowptr), torch.from_numpy(col)


def user_hist_csr_from_edges(edge_index, num_user):
    """Vectorised form for large graphs (config 5): edges [E,2] with global item ids.  A CUDA tensor is laid out on
    the device (and the CSR stays there)."""
    if torch.is_tensor(edge_index):
        u = edge_index[:, 0].to(torch.int64)
        key = torch.unique((u << 32) + (edge_index[:, 1].to(torch.int64) - num_user))
        u = key >> 32
        rowptr = torch.zeros(num_user + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(torch.bincount(u, minlength=num_user), 0, out=rowptr[1:])
        return rowptr, (key & 0xFFFFFFFF).to(torch.int32)
    e = np.asarray(edge_index, dtype=np.int64)
    u, i = e[:, 0], e[:, 1] - num_user
    key = np.unique(u * (1 << 32) + i)
    u, i = key >> 32, key & 0xFFFFFFFF
    rowptr = np.zeros(num_user + 1, dtype=np.int64)
    np.cumsum(np.bincount(u, minlength=num_user), out=rowptr[1:])
    return torch.from_numpy(rowptr), torch.from_numpy(i.astype(np.int32))


def user_item_dict_from_edges(edge_index):
    """the missing user_item_dict.npy blobs are the train edges grouped by user in
    file order (verified against the shipped dicts by tests/golden/gen_golden.py)."""
    d = {}
    for u, i in np.asarray(edge_index).tolist():
        d.setdefault(u, []).append(i)
    return d


def coo_to_csr_coalesced(row, col, val, n_rows, n_cols, symmetric=False):
    """COO (destination=row, source=col) -> CSR with COALESCED rows: entries sorted by column, duplicates
    summed -- the form torch.sparse.mm works on (Model/FREEDOM.py:168,174 coalesce their operand first).
    Runs on whatever device the inputs live on (FREEDOM re-prunes its graph on the GPU every epoch)."""
    row = torch.as_tensor(row).long()
    col = torch.as_tensor(col).long()
    val = torch.as_tensor(val, dtype=torch.float32, device=row.device)
    key = row * n_cols + col
    order = torch.argsort(key, stable=True)
    key, val = key[order], val[order]
    uniq, inverse = torch.unique_consecutive(key, return_inverse=True)
    if uniq.numel() != key.numel():
        val = torch.zeros(uniq.numel(), dtype=torch.float32, device=val.device).index_add_(0, inverse, val)
    r = torch.div(uniq, n_cols, rounding_mode="floor")
    c = uniq - r * n_cols
    counts = torch.bincount(r, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return CSR(rowptr, c.to(torch.int32).contiguous(), val.contiguous(), n_rows, n_cols, symmetric)


def add_scaled_coo(a, wa, b, wb, n):
    """wa * A + wb * B for two (indices [2,nnz], values [nnz]) COO pairs over an n x n matrix -> coalesced CSR
    (Model/FREEDOM.py:69: mm_adj = w * image_adj + (1 - w) * text_adj)."""
    (ia, va), (ib, vb) = a, b
    idx = torch.cat([ia, ib], dim=1)
    val = torch.cat([wa * va, wb * vb])
    return coo_to_csr_coalesced(idx[0], idx[1], val, n, n, symmetric=False)


def inv_sqrt_degree_edge_weights(src, dst, n_src, n_dst, eps=1e-7):
    """Weight of every edge (src_e, dst_e) of a bipartite list: (deg(src_e) + eps)^-1/2 * (deg(dst_e) + eps)^-1/2, degrees
    counted on THIS list (FREEDOM normalises its full interaction list once and every epoch's pruned list again:
    Model/FREEDOM.py:85-99).  Integer counts, then + eps (which makes them fp32), then the power, then one product per
    edge -- the order in which the reference's sparse row / column sums round, so the values are bit-identical."""
    inv_s = (torch.bincount(src, minlength=n_src) + eps).pow(-0.5)
    inv_d = (torch.bincount(dst, minlength=n_dst) + eps).pow(-0.5)
    return inv_s[src] * inv_d[dst]


def out_degree_normalised_weights(rows, cols, n, eps=1e-7):
    """Weight of every entry (r, c) of an n x n adjacency normalised by its ROW sums on both sides:
    (rowsum(r) + eps)^-1/2 * (rowsum(c) + eps)^-1/2 (the kNN item graph, Model/FREEDOM.py:128-138: every row has exactly
    k entries, so this is 1/k up to eps)."""
    inv = (torch.bincount(rows, minlength=n) + eps).pow(-0.5)
    return inv[rows] * inv[cols]


def symmetric_bipartite_csr(users, items, weights, num_user, num_item):
    """CSR of [[0, W], [W^T, 0]] over users then items for the weighted interactions (users_e, items_e, weights_e):
    entries coalesced per row in ascending column order (what torch.sparse.mm works on).  On the inputs' device."""
    n = num_user + num_item
    shifted = items + num_user
    return coo_to_csr_coalesced(torch.cat((users, shifted)), torch.cat((shifted, users)), torch.cat((weights, weights)),
                                n, n, symmetric=True)


class DropoutStructure(CSR):
    """The CSR of D^-1/2 (A + I) D^-1/2 plus what the per-step edge dropout needs (ops.edge_dropout_norm): the
    destination of every entry, the entry of the reversed edge, and a degree workspace.  `val` holds the
    no-dropout normalisation; with_values() gives a view of the same structure over another value array that shares
    this graph's SpMM schedule (built once; the kernel is told to take every value from the array,
    CHAOREC_SPMM_DYNAMIC_VALUES, because the schedule's inlined values are the static ones)."""

    def __init__(self, csr, entry_row, transpose_entry, deg_ws=None):
        super().__init__(csr.rowptr, csr.col, csr.val, csr.n_rows, csr.n_cols, symmetric=True)
        self.entry_row, self.transpose_entry = entry_row, transpose_entry
        self.deg_ws = deg_ws if deg_ws is not None else torch.zeros(csr.n_rows, dtype=torch.int32,
                                                                    device=csr.rowptr.device)

    def to(self, device):
        return DropoutStructure(CSR.to(self, device), self.entry_row.to(device), self.transpose_entry.to(device),
                                self.deg_ws.to(device))

    def with_values(self, val):
        v = CSR(self.rowptr, self.col, val, self.n_rows, self.n_cols)
        v.schedule = self.schedule          # same structure, same descriptors
        v.dynamic_values = True
        return v


def ngcf_structure(edge_index, n_nodes):
    """NGCFConv.forward's graph (Model/NGCF.py:49-58) for p = 0 -- the same D^-1/2 (A+I) D^-1/2 as BasicGCN -- and
    the entry maps the dropout kernels need.  transpose_entry pairs the j-th copy of (r, c) with the j-th copy of
    (c, r), so it is a bijection even when the edge list repeats an interaction."""
    csr = basicgcn_csr(edge_index, n_nodes)
    counts = csr.rowptr[1:] - csr.rowptr[:-1]
    entry_row = torch.repeat_interleave(torch.arange(n_nodes, dtype=torch.int64), counts)
    col = csr.col.to(torch.int64)
    fwd = torch.argsort(entry_row * n_nodes + col, stable=True)
    rev = torch.argsort(col * n_nodes + entry_row, stable=True)
    tentry = torch.empty(col.numel(), dtype=torch.int64)
    tentry[rev] = fwd
    return DropoutStructure(csr, entry_row.to(torch.int32), tentry.to(torch.int32))


def binary_sym_norm_csr(users, items, num_user, num_item):
    """The adjacency every `torch.sparse.mm` model of the reference builds with scipy (e.g. Model/SimGCL.py:64-112,
    Model/NCL.py:97-137, Model/SelfCF.py:63-99): binary A over users then items (a repeated interaction counts once),
    degree = distinct neighbours + 1e-7, D^-1/2 A D^-1/2 evaluated in fp64 and stored as fp32 -- vectorised (the
    reference fills a dok matrix entry by entry).  users / items: int64 tensors, LOCAL item ids.  -> coalesced CSR."""
    import numpy as np
    U, I, N = num_user, num_item, num_user + num_item
    key = torch.unique(users.long() * I + items.long())
    u, i = torch.div(key, I, rounding_mode="floor"), key % I
    deg = torch.zeros(N, dtype=torch.float64)
    deg.index_add_(0, u, torch.ones(u.numel(), dtype=torch.float64))
    deg.index_add_(0, U + i, torch.ones(i.numel(), dtype=torch.float64))
    d = torch.from_numpy(np.power(deg.numpy() + 1e-7, -0.5))
    val = ((d[u] * 1.0) * d[U + i]).to(torch.float32)
    return coo_to_csr_coalesced(torch.cat([u, U + i]), torch.cat([U + i, u]), torch.cat([val, val]), N, N, symmetric=True)
