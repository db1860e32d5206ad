"""MGAT with the reference's surface (Model/MGAT.py:17-251) -- MMGCN's layout (per modality: user preferences | projected item
features, three graph layers with id-embedding skips, the modalities averaged) with a GATED ATTENTION propagate: the weight
of an edge i <- j is the softmax, over the edges arriving at i, of  s * sigmoid(deg(j)^-1/2 * s),  s = <x_i W, leaky_relu(x_j W)>.

PyG's per-edge gather / softmax / scatter is here one symmetric CSR over the distinct interactions and, per layer, a value
array over it -- two segment softmaxes in pair space (edges arriving at users, edges arriving at items; a repeated interaction
counts as often as it is listed) -- on the dynamic-values HIP SpMM whose values receive their gradient (`sparse.DroppedAdj`,
the same treatment as GRCN).  The dense projections (x W, the linear / gate layers, the [I, F] feature MLP) are `ops.linear` on
the MFMA GEMM, the ranking `ranking.gene_ranklist` over the [N, 3 dim_E] table of the last forward.

Same constructor, parameters in the reference's creation order and drawn the same way (torch_geometric's `uniform` for the
attention layers' weight and bias, the weight then re-drawn by xavier_normal_, :79-95).  The reference keeps the two feature
matrices as leaf tensors that require a gradient nobody reads (:130-131); they are buffers here."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter

from .. import graph, ops, ranking, sparse
from .GRCN import _segment_softmax


class GraphGAT(nn.Module):
    """:17-61."""

    def __init__(self, in_channels, out_channels, normalize=True, bias=True, aggr='add'):
        super(GraphGAT, self).__init__()
        self.in_channels, self.out_channels, self.normalize, self.dropout = in_channels, out_channels, normalize, 0.1
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        self.bias = Parameter(torch.Tensor(out_channels)) if bias else None
        bound = 1.0 / math.sqrt(in_channels)                    # torch_geometric.nn.inits.uniform(size, tensor)
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.uniform_(-bound, bound)

    def forward(self, owner, x):
        x = ops.linear(x, self.weight.t().contiguous())
        U = owner.num_user
        d, lx = owner._deg_inv_sqrt, F.leaky_relu(x)

        def attention(inner, d_src, seg, n_seg):
            tmp = torch.mul(inner, torch.sigmoid(torch.mul(d_src, inner)))
            return _segment_softmax(tmp, seg, n_seg, owner._ew)

        # <x_i, leaky_relu(x_j)> per pair, both directions: edge scores over the structure's (user, item) half
        to_user = attention(ops.edge_dot(owner._structure, x, lx, owner.n_edges), d[U + owner._ei], owner._eu, owner.num_user)   # item -> user
        to_item = attention(ops.edge_dot(owner._structure, lx, x, owner.n_edges), d[owner._eu], owner._ei, owner.num_item)
        up, low = owner._ew * to_user, owner._ew * to_item
        adj = sparse.DroppedAdj(owner._structure, torch.cat([up, low[owner._lower]]), torch.cat([low, up[owner._lower]]))
        out = sparse.mm(adj, x)
        if self.bias is not None:
            out = out + self.bias
        return F.normalize(out, p=2, dim=-1) if self.normalize else out


class GNN(torch.nn.Module):
    """:63-111."""

    def __init__(self, features, num_user, num_item, dim_E, dim_latent=None):
        super(GNN, self).__init__()
        self.num_user, self.num_item, self.dim_E, self.dim_feat, self.dim_latent = num_user, num_item, dim_E, features.size(1), dim_latent
        self.register_buffer("features", features.clone().detach(), persistent=False)
        self.preference = nn.Embedding(num_user, dim_latent)
        nn.init.xavier_normal_(self.preference.weight)
        self.MLP = nn.Linear(self.dim_feat, dim_latent)
        self.conv_embed_1 = GraphGAT(dim_latent, dim_latent, aggr='add')
        self.linear_layer1 = nn.Linear(dim_latent, dim_E)
        self.g_layer1 = nn.Linear(dim_latent, dim_E)
        nn.init.xavier_normal_(self.conv_embed_1.weight)
        nn.init.xavier_normal_(self.linear_layer1.weight)
        nn.init.xavier_normal_(self.g_layer1.weight)
        for k in (2, 3):
            conv, lin, gl = GraphGAT(dim_E, dim_E, aggr='add'), nn.Linear(dim_E, dim_E), nn.Linear(dim_E, dim_E)
            setattr(self, f"conv_embed_{k}", conv)
            setattr(self, f"linear_layer{k}", lin)
            setattr(self, f"g_layer{k}", gl)
            nn.init.xavier_normal_(conv.weight)
            nn.init.xavier_normal_(lin.weight)
            nn.init.xavier_normal_(gl.weight)

    def forward(self, owner, id_embedding):
        lin = lambda layer, t, act=0: ops.linear(t, layer.weight, layer.bias, act=act)
        temp_features = torch.tanh(lin(self.MLP, self.features))
        x = F.normalize(torch.cat((self.preference.weight, temp_features), dim=0))
        outs = []
        for k in (1, 2, 3):
            h = F.leaky_relu(getattr(self, f"conv_embed_{k}")(owner, x))
            x_hat = lin(getattr(self, f"linear_layer{k}"), x, act=1) + id_embedding.weight
            x = F.leaky_relu(lin(getattr(self, f"g_layer{k}"), h) + x_hat)
            outs.append(x)
        return torch.cat(outs, dim=1)


class MGAT(torch.nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, reg_weight, device):
        super(MGAT, self).__init__()
        self.num_user, self.num_item, self.dim_E, self.device = num_user, num_item, dim_E, device
        self.user_item_dict, self.reg_weight = user_item_dict, reg_weight
        U, I = num_user, num_item
        self._pairs = sparse.PairStructure(edge_index, U, I, device)      # distinct interactions + one symmetric [N, N] structure
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self._lower, self._structure = self._pairs.lower, self._pairs.structure
        deg = torch.zeros(U + I, dtype=torch.float32, device=device).index_add_(0, self._eu, self._ew).index_add_(0, U + self._ei, self._ew)
        self._deg_inv_sqrt = deg.pow(-0.5)                       # (degree(row) over the bidirectional list, :45-46; only read at edge endpoints)
        self.v_gnn = GNN(v_feat, num_user, num_item, dim_E, dim_latent=256)
        self.t_gnn = GNN(t_feat, num_user, num_item, dim_E, dim_latent=100)
        self.id_embedding = nn.Embedding(num_user + num_item, dim_E)
        nn.init.xavier_normal_(self.id_embedding.weight)
        self.result = nn.init.xavier_normal_(torch.rand((num_user + num_item, dim_E))).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def forward(self):
        self.result = (self.v_gnn(self, self.id_embedding) + self.t_gnn(self, self.id_embedding)) / 2
        return self.result

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        u, p, n = embeddings[users], embeddings[self.num_user + pos_items], embeddings[self.num_user + neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items, embeddings):
        u, p, n = embeddings[users], embeddings[self.num_user + pos_items], embeddings[self.num_user + neg_items]
        return self.reg_weight * (torch.mean(u ** 2) + torch.mean(p ** 2) + torch.mean(n ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        embeddings = self.forward()
        return self.bpr_loss(users, pos_items, neg_items, embeddings) + self.regularization_loss(users, pos_items, neg_items, embeddings)

    def gene_ranklist(self, topk=50, to_cpu=True):
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
