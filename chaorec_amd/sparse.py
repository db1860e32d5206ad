"""Drop-in for `torch.sparse.mm(norm_adj, x)` -- the propagate of the 34 `sparse.mm` models of the reference
(e.g. Model/SimGCL.py:64-112: a scipy-built D^-1/2 A D^-1/2 as a torch COO tensor, multiplied
into the embedding table every layer).

    from chaorec_amd import sparse
    ego_embeddings = sparse.mm(self.sparse_norm_adj, ego_embeddings)      # was: torch.sparse.mm(...)

The COO tensor is converted ONCE into the kernel's CSR (coalesced, cached on the tensor object) and every call
is one chaorec_spmm_csr_f32 launch with autograd (backward = the transposed CSR, or the same one when the
matrix is symmetric).
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


def mm(adj, dense):
    """torch.sparse.mm(adj, dense) on the HIP SpMM kernel; `adj` may be a torch sparse tensor or a graph.CSR."""
    if isinstance(adj, graph.CSR):
        return ops.spmm(adj, dense)
    csr = getattr(adj, _CACHE_ATTR, None)
    if csr is None:
        csr = from_torch_sparse(adj)
        try:
            setattr(adj, _CACHE_ATTR, csr)
        except AttributeError:
            pass
    return ops.spmm(csr, dense)
