"""Drop-in for `torch.sparse.mm(norm_adj, x)` -- the propagate of the 34 `sparse.mm` models of the reference
(e.g. Model/SimGCL.py:64-112: a scipy-built D^-1/2 A D^-1/2 as a torch COO tensor, multiplied
into the embedding table every layer).

    from chaorec_amd import sparse
    ego_embeddings = sparse.mm(self.sparse_norm_adj, ego_embeddings)      # was: torch.sparse.mm(...)

The COO tensor is converted ONCE into the kernel's CSR (coalesced, cached on the tensor object) and every call
is one chaorec_spmm_csr_f32 launch with autograd (backward = the transposed CSR, or the same one when the
matrix is symmetric).

`sparse_dropout(adj, rate)` is the family's per-step edge dropout helper (Model/SelfCF.py:101-112 and its copies:
every stored entry kept with probability 1 - rate, the kept ones scaled by 1 / (1 - rate)).  The reference builds a
new COO tensor per step; here the structure stays (one CSR, one SpMM schedule) and only the VALUE array changes --
a dropped entry is a zero value, which adds +0 * x to its row's sum -- so `mm` of the result is the same kernel in
its dynamic-values mode, forward over the masked values and backward over the masked values of the transpose.
"""
import torch

from . import graph, ops

_CACHE_ATTR = "_chaorec_csr"


def from_torch_sparse(adj, assume_symmetric=None):
    """torch sparse COO/CSR [n, m] -> graph.CSR on the same device (row = output row, as torch.sparse.mm)."""
    if adj.layout == torch.sparse_csr:
        adj = adj.to_sparse_coo()
    a = adj.coalesce()
    idx, val = a.indices(), a.values().to(torch.float32)
    n, m = a.shape
    csr = graph.coo_to_csr_coalesced(idx[0], idx[1], val, n, m)
    if assume_symmetric is None and n == m:
        # symmetric iff A^T has the same coalesced entries
        t = torch.sparse_coo_tensor(torch.stack([idx[1], idx[0]]), val, (m, n)).coalesce()
        assume_symmetric = bool(t.indices().shape == idx.shape and torch.equal(t.indices(), idx)
                                and torch.equal(t.values(), val))
    csr.symmetric = bool(assume_symmetric)
    return csr


def _cached_csr(adj):
    csr = getattr(adj, _CACHE_ATTR, None)
    if csr is None:
        csr = from_torch_sparse(adj)
        try:
            setattr(adj, _CACHE_ATTR, csr)
        except AttributeError:
            pass
    return csr


def mm(adj, dense):
    """torch.sparse.mm(adj, dense) on the HIP SpMM kernel; `adj` may be a torch sparse tensor, a graph.CSR or the
    result of sparse_dropout()."""
    if isinstance(adj, DroppedAdj):
        return ops.spmm_values(adj.structure, adj.val, adj.val_t, dense)
    if isinstance(adj, LearnedAdj):
        return _SpMMLearned.apply(dense, adj, adj.val)
    if isinstance(adj, graph.CSR):
        return ops.spmm(adj, dense)
    return ops.spmm(_cached_csr(adj), dense)


class DroppedAdj:
    """`sparse_dropout`'s result: the adjacency's fixed structure with this step's values (and its transpose's)."""

    def __init__(self, structure, val, val_t):
        self.structure, self.val, self.val_t = structure, val, val_t


class PairStructure:
    """The fixed skeleton of the models whose bipartite graph changes its VALUES per step (MMGCL, SGL, DDRec, DCCF, GRCN, MGAT): the
    distinct (user, item) interactions of an edge list in row-major order (`eu`, `ei`, multiplicity `ew`; `pair_of_edge` maps a
    LISTED edge to its pair), and ONE symmetric [N, N] structure over them (`structure`, a graph.DropoutStructure on `csr`) whose
    first n entries are the pairs (u, U + i) in that order and whose other n entries are the same pairs ordered by (item, user):
    `lower[j]` = the pair of the j-th of those.  both(up, low) lays two per-pair value arrays out over the structure: `up` at
    the entries (u, U + i) -- an edge item -> user --, `low` at (U + i, u)."""

    def __init__(self, edge_index, num_user, num_item, device):
        import numpy as np
        U, I = num_user, num_item
        e = torch.as_tensor(np.asarray(edge_index)).long()
        self.n_listed = int(e.shape[0])
        key, pair_of_edge, cnt = torch.unique(e[:, 0] * I + (e[:, 1] - U), return_inverse=True, return_counts=True)
        self.pair_of_edge = pair_of_edge.to(device)
        self.eu, self.ei = torch.div(key, I, rounding_mode="floor").to(device), (key % I).to(device)
        self.ew = cnt.to(torch.float32).to(device)
        self.n = int(key.numel())
        self.csr = graph.coo_to_csr_coalesced(torch.cat([self.eu, U + self.ei]), torch.cat([U + self.ei, self.eu]),
                                              torch.ones(2 * self.n, device=device), U + I, U + I, symmetric=True)
        self.lower = torch.argsort(self.ei * U + self.eu, stable=True)
        self.structure = _dropout_structure(self.csr)

    def both(self, up, low=None):
        """-> [2 n] values in the structure's entry order (low = up: a symmetric matrix)."""
        return torch.cat([up, (up if low is None else low)[self.lower]])

    def kept_copies(self, keep):
        """per-pair weight = how many LISTED copies of the pair a bool [n_listed] mask keeps"""
        return torch.zeros(self.n, dtype=torch.float32, device=self.ew.device).index_add_(0, self.pair_of_edge, keep.to(torch.float32))


class LearnedAdj:
    """A sparse operand built from embeddings THIS step (Model/MICRO.py:176-187: a kNN graph of the projected features), its
    values carrying gradient, over its own structure -- any pattern; row-major entries (rowptr, col), `val` [nnz].  `mm` of it
    is the HIP SpMM on a CSR laid down for this step, backward = the SpMM over the transposed layout (built on first use)
    for the dense operand and d val[k] = <gy[row_k], x[col_k]> for the values.  detach() -> the constant graph.CSR the
    following steps multiply with (the reference detaches the tensor, :188-190)."""

    def __init__(self, rowptr, col, val, n_rows, n_cols):
        self.rowptr, self.col, self.val = rowptr, col.to(torch.int32).contiguous(), val
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self._t = None

    def entry_rows(self):
        if getattr(self, "_rows", None) is None:
            self._rows = torch.repeat_interleave(torch.arange(self.n_rows, dtype=torch.int64, device=self.col.device),
                                                 self.rowptr[1:] - self.rowptr[:-1])
            self._rows32 = self._rows.to(torch.int32)
        return self._rows

    def transposed(self):
        """-> (rowptr_t, col_t, perm): entry k of A^T's row-major layout is entry perm[k] of A's."""
        if self._t is None:
            rows, col = self.entry_rows(), self.col.to(torch.int64)
            perm = torch.argsort(col * self.n_rows + rows, stable=True)
            rowptr_t = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=col.device)
            torch.cumsum(torch.bincount(col, minlength=self.n_cols), 0, out=rowptr_t[1:])
            self._t = (rowptr_t, rows[perm].to(torch.int32).contiguous(), perm)
        return self._t

    def detach(self):
        return graph.CSR(self.rowptr, self.col, self.val.detach().contiguous(), self.n_rows, self.n_cols)


class _SpMMLearned(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, val):
        ctx.adj = adj
        ctx.save_for_backward(x, val)
        return ops.spmm_raw(graph.CSR(adj.rowptr, adj.col, val.detach().contiguous(), adj.n_rows, adj.n_cols), x)

    @staticmethod
    def backward(ctx, gy):
        x, val = ctx.saved_tensors
        adj, gy = ctx.adj, gy.contiguous()
        gx = gval = None
        if ctx.needs_input_grad[0]:
            rowptr_t, col_t, perm = adj.transposed()
            gx = ops.spmm_raw(graph.CSR(rowptr_t, col_t, val.detach()[perm].contiguous(), adj.n_cols, adj.n_rows), gy)
        if ctx.needs_input_grad[2]:
            adj.entry_rows()
            gval = ops.edge_dot_raw(adj._rows32, adj.col, gy, x)
        return gx, None, gval


def _dropout_structure(csr):
    """graph.DropoutStructure over an arbitrary square CSR (entry -> its row, entry -> the entry of the reversed pair;
    entries without a stored reverse cannot be transposed in place and raise)."""
    n = csr.n_rows
    dev = csr.col.device
    counts = csr.rowptr[1:] - csr.rowptr[:-1]
    entry_row = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev), counts)
    col = csr.col.to(torch.int64)
    fwd = torch.argsort(entry_row * n + col, stable=True)
    rev = torch.argsort(col * n + entry_row, stable=True)
    if not torch.equal((entry_row * n + col)[fwd], (col * n + entry_row)[rev]):
        raise ValueError("sparse_dropout: the adjacency's pattern is not symmetric")
    tentry = torch.empty(col.numel(), dtype=torch.int64, device=dev)
    tentry[rev] = fwd
    return graph.DropoutStructure(csr, entry_row.to(torch.int32), tentry.to(torch.int32))


def sparse_dropout(adj, rate, keep=None, generator=None):
    """Model/SelfCF.py:101-112: keep each stored entry with probability 1 - rate (mask = floor(1 - rate + U[0,1))), scale
    the kept ones by 1 / (1 - rate).  `adj`: a torch sparse tensor or a graph.CSR with a symmetric pattern; `keep`
    (bool [nnz] in the CSR's entry order, optional) replaces the draw.  -> DroppedAdj for `mm`."""
    csr = adj if isinstance(adj, graph.CSR) else _cached_csr(adj)
    st = getattr(csr, "_dropout_structure", None)
    if st is None:
        st = csr._dropout_structure = _dropout_structure(csr)
    if keep is None:
        keep = torch.floor(1.0 - rate + torch.rand(csr.nnz, device=csr.val.device, generator=generator)).to(torch.bool)
    val = torch.where(keep, csr.val * (1.0 / (1.0 - rate)), torch.zeros((), dtype=csr.val.dtype, device=csr.val.device))
    return DroppedAdj(st, val, val[st.transpose_entry.long()])
