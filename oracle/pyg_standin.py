"""Restatement of the torch-geometric 2.1.0 / torch-scatter 2.0.9 call surface that the
reference's LightGCN / MMGCN / BasicGCN import (Model/LightGCN.py:14,16; BasicGCN.py:13-15).

TEST INFRASTRUCTURE ONLY.  The third-party packages are pinned in the reference's
requirements.txt:49-50 but are neither vendored nor installed in the build container, so
tests/golden/gen_golden.py installs this stand-in into sys.modules before importing the
reference classes.  Goldens produced that way pin "reference model code + THIS restatement of
propagate", and say so.  Published semantics restated here:

  * MessagePassing(aggr='add', flow='source_to_target').propagate(edge_index, x=..., norm=...):
      x_j = x.index_select(0, edge_index[0]); out = scatter(message(x_j, ...), edge_index[1],
      dim=0, dim_size=N, reduce='add'); return update(out)
  * torch_scatter.scatter(reduce='sum') = zeros.scatter_add_(0, broadcast(index), src)
  * utils.degree(index, num_nodes, dtype) = zeros(N).scatter_add_(0, index, ones)
  * utils.add_self_loops(edge_index, num_nodes) appends arange(N) x2 AFTER the edges
  * utils.dropout_adj(edge_index, p): mask = torch.rand(E) >= p; edge_index[:, mask]   (Model/NGCF.py:40)
  * utils.remove_self_loops(edge_index): the columns with row != col; a message() that names `edge_index` / `size` receives
    propagate's own two arguments (Model/MENTOR.py:82-99, Model/DDRec.py / Model/MICRO.py through their GCNConv); one that names
    `edge_index_i` / `size_i` receives the target index row and the node count (Model/GRCN.py:31)
  * utils.softmax(src, index, num_nodes): exp(src - max of the group) / (sum over the group + 1e-16)
  * nn.inits.uniform(size, tensor): tensor ~ U(-1/sqrt(size), 1/sqrt(size))
"""
import inspect
import sys
import types

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        assert flow == "source_to_target" and node_dim == 0
        self.aggr = aggr
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs["x"]
        n = size[1] if size is not None else x.size(0)
        src, dst = edge_index[0], edge_index[1]
        msg_kwargs = {}
        for name in self._msg_params:
            if name in ("edge_index_i", "edge_index_j", "size_i", "size_j"):     # (Model/GRCN.py:31: the target / source index rows, the node counts)
                msg_kwargs[name] = {"edge_index_i": dst, "edge_index_j": src, "size_i": n,
                                    "size_j": size[0] if size is not None else x.size(0)}[name]
            elif name.endswith("_j"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(0, src)
            elif name.endswith("_i"):
                msg_kwargs[name] = kwargs[name[:-2]].index_select(0, dst)
            elif name == "edge_index":          # (PyG hands these two to a message() that names them: Model/MENTOR.py:92)
                msg_kwargs[name] = edge_index
            elif name == "size":
                msg_kwargs[name] = list(size) if size is not None else [x.size(0), x.size(0)]
            else:
                msg_kwargs[name] = kwargs[name]
        msg = self.message(**msg_kwargs)
        assert self.aggr == "add", "only add-aggregation is on the hot path"
        index = dst.view(-1, 1).expand_as(msg)
        out = torch.zeros((n, msg.size(1)), dtype=msg.dtype, device=msg.device).scatter_add_(0, index, msg)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype, device=index.device)
    one = torch.ones((index.size(0),), dtype=out.dtype, device=out.device)
    return out.scatter_add_(0, index, one)


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(0, n, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index, loop], dim=1), edge_attr


DROPOUT_LOG = []   # every keep mask dropout_adj drew, in call order (the golden generator stores them as inputs)


def dropout_adj(edge_index, edge_attr=None, p=0.5, force_undirected=False, num_nodes=None, training=True):
    """torch_geometric.utils.dropout_adj (2.1.0): keep each edge independently with probability 1 - p."""
    assert not force_undirected and edge_attr is None
    if not training or p == 0.0:
        return edge_index, edge_attr
    mask = torch.rand(edge_index.size(1), device=edge_index.device) >= p
    DROPOUT_LOG.append(mask.clone())
    return edge_index[:, mask], edge_attr


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_add (2.0.9) for 1-D src along dim 0 (Model/MGCN.py:21-24)."""
    assert dim == 0 and out is None and src.dim() == 1
    n = int(index.max()) + 1 if dim_size is None else dim_size
    return torch.zeros((n,), dtype=src.dtype, device=src.device).scatter_add_(0, index, src)


def softmax(src, index, ptr=None, num_nodes=None):
    """torch_geometric.utils.softmax 2.1: per `index` group, exp(src - group max) / (group sum + 1e-16)."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    mx = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device).scatter_reduce(0, idx, src, reduce="amax")
    out = (src - mx.index_select(0, index)).exp()
    den = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(0, idx, out)
    return out / (den.index_select(0, index) + 1e-16)


def uniform(size, tensor):
    """torch_geometric.nn.inits.uniform: U(-1/sqrt(size), 1/sqrt(size)) in place (Model/MGAT.py:33-35)."""
    if tensor is not None:
        bound = 1.0 / (size ** 0.5)
        tensor.data.uniform_(-bound, bound)


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (edge_attr[keep] if edge_attr is not None else None)


def _unused(*a, **k):
    raise NotImplementedError("not on the hot path")


def install():
    """Register stand-in modules under the torch_geometric names the reference imports."""
    if "torch_geometric" in sys.modules and not getattr(sys.modules["torch_geometric"], "_standin", False):
        return  # a real install wins
    tg = types.ModuleType("torch_geometric")
    tg._standin = True
    nn = types.ModuleType("torch_geometric.nn")
    conv = types.ModuleType("torch_geometric.nn.conv")
    inits = types.ModuleType("torch_geometric.nn.inits")
    utils = types.ModuleType("torch_geometric.utils")
    conv.MessagePassing = MessagePassing
    nn.MessagePassing = MessagePassing
    nn.conv = conv
    nn.inits = inits
    inits.uniform = uniform
    utils.degree = degree
    utils.add_self_loops = add_self_loops
    utils.remove_self_loops = remove_self_loops
    utils.softmax = softmax
    utils.dropout_adj = dropout_adj
    tg.nn = nn
    tg.utils = utils
    ts = types.ModuleType("torch_scatter")
    ts._standin = True
    ts.scatter_add = scatter_add
    if "torch_scatter" not in sys.modules:
        sys.modules["torch_scatter"] = ts
    for name, mod in (("torch_geometric", tg), ("torch_geometric.nn", nn),
                      ("torch_geometric.nn.conv", conv), ("torch_geometric.nn.inits", inits),
                      ("torch_geometric.utils", utils)):
        sys.modules[name] = mod
