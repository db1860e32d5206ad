"""Graph-conv primitives with the reference's class names (BasicGCN.py:21-87), on the CSR SpMM kernel.

The reference classes are torch_geometric MessagePassing modules that recompute degree + norm and
re-gather the edge list on every call.  Here `forward(x, edge_index)` accepts either the same
[2, E_dir] LongTensor (converted to a normalised CSR in HBM once and cached) or a ready
chaorec_amd.graph.CSR, and runs ONE SpMM launch.
"""
import torch

from . import graph, ops


def _as_csr(edge_index, n_nodes, builder, cache):
    if isinstance(edge_index, graph.CSR):
        return edge_index
    key = (edge_index.data_ptr(), tuple(edge_index.shape), n_nodes)
    if key not in cache:
        ei = edge_index.detach().cpu().long()
        cache.clear()
        cache[key] = builder(ei, n_nodes).to(edge_index.device)
    return cache[key]


def _sym_norm_csr(ei, n_nodes, self_loops):
    row, col = ei[0], ei[1]
    if self_loops:  # BasicGCN.py:37: appended after the edges
        loop = torch.arange(n_nodes, dtype=torch.int64)
        row, col = torch.cat([row, loop]), torch.cat([col, loop])
    deg = torch.zeros(n_nodes, dtype=torch.float32).scatter_add_(0, row, torch.ones(row.numel()))
    dis = deg.pow(-0.5)
    return graph.coo_to_csr(col, row, dis[row] * dis[col], n_nodes, n_nodes, symmetric=False)


class BasicGCN(torch.nn.Module):
    """BasicGCN.py:21-59: self loops -> Linear (with bias) -> D^-1/2 (A+I) D^-1/2 add-aggregate."""

    def __init__(self, in_channels, out_channels, aggr='add', **kwargs):
        super(BasicGCN, self).__init__()
        assert aggr == 'add', "only add-aggregation is on the hot path (main.py passes args.aggr_mode='add')"
        self.aggr = aggr
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = torch.nn.Linear(in_channels, out_channels)
        self._cache = {}

    def forward(self, x, edge_index):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if hasattr(edge_index, "propagate"):      # a graph operator (chaorec_amd.dist.ShardedGraph): rows are sharded
            return edge_index.propagate(ops.linear(x, self.lin.weight, self.lin.bias))
        csr = _as_csr(edge_index, x.size(0), lambda ei, n: _sym_norm_csr(ei, n, True), self._cache)
        x = ops.linear(x, self.lin.weight, self.lin.bias)
        return ops.spmm(csr, x)

    def __repr__(self):
        return '{}({},{})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class GCNConv(torch.nn.Module):
    """BasicGCN.py:63-87: LightGCN-style conv, symmetric normalisation only."""

    def __init__(self, in_channels, out_channels, aggr='add', **kwargs):
        super(GCNConv, self).__init__()
        self.aggr = aggr
        self._cache = {}

    def forward(self, x, edge_index):
        csr = _as_csr(edge_index, x.size(0), lambda ei, n: _sym_norm_csr(ei, n, False), self._cache)
        return ops.spmm(csr, x)
