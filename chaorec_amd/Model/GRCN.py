"""GRCN with the reference's surface (Model/GRCN.py:18-311) -- graph-refined convolution: per modality a content GCN whose one
attention layer (GATConv: softmax over a node's incoming edges of the endpoint rows' dot product) yields a weight per DIRECTED
edge, the modalities' weights are combined with learned per-node confidences and pruned (relu), and an id GCN propagates the
normalised id table twice over the graph weighted that way.  The edge weights carry gradient into the content GCNs.

PyG's per-edge gather / scatter (MessagePassing, `softmax`, `dropout_adj`) is here ONE symmetric CSR over the distinct
interactions, built once, and per step value arrays over it (`sparse.DroppedAdj` on the dynamic-values HIP SpMM, whose values
receive their gradient, d value = <gy[dst], x[src]>): an edge i <- j of the bidirectional list is the entry (row i, column j); the
softmax over a node's incoming edges is a segment softmax over its row; the per-step edge dropout (:202) is a 0 / multiplicity
weight per pair.  The MLPs are `ops.linear` on the MFMA GEMM, the ranking `ranking.gene_ranklist` with this model's 1e-5 mask
(:299) over the [N, dim_E + 2 dim_C] table of the last forward.

Kept quirks: the routing iterations of the content GCN run the attention layer on the ONE-directional list (user -> item, :161-166),
whose messages only reach item rows -- the user rows they add to the preferences are zero, a routing step is a re-normalisation
(no product is launched for it); `weight_mode` / `fusion_mode` / `pruning` keep their defaults' branches (confid / concat / on).
Batches are MMGCN's ([B, 2] user / item tensors with global item ids)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


def _segment_softmax(logit, seg, n_seg, w):
    """torch_geometric.utils.softmax(src, index, num_nodes) over the pairs with w > 0, a pair of multiplicity w counting w times:
    exp(src - max of its segment) / (sum over the segment + 1e-16)."""
    live = w > 0
    neg = torch.full((n_seg,), float("-inf"), dtype=logit.dtype, device=logit.device)
    m = neg.scatter_reduce(0, seg, torch.where(live, logit, torch.full_like(logit, float("-inf"))), reduce="amax")
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m).detach()
    e = torch.where(live, torch.exp(logit - m[seg]), torch.zeros_like(logit))
    denom = torch.zeros(n_seg, dtype=logit.dtype, device=logit.device).index_add(0, seg, w * e)
    return e / (denom[seg] + 1e-16)


class EGCN(torch.nn.Module):
    """:58-77."""

    def __init__(self, num_user, num_item, dim_E, aggr_mode):
        super(EGCN, self).__init__()
        self.num_user, self.num_item, self.dim_E, self.aggr_mode = num_user, num_item, dim_E, aggr_mode
        self.id_embedding = nn.Parameter(nn.init.xavier_normal_(torch.rand((num_user + num_item, dim_E))))

    def forward(self, adj):
        x = F.normalize(self.id_embedding)
        x_hat_1 = F.leaky_relu(sparse.mm(adj, x))
        x_hat_2 = F.leaky_relu(sparse.mm(adj, x_hat_1))
        return x + x_hat_1 + x_hat_2


class CGCN(torch.nn.Module):
    """:79-107."""

    def __init__(self, features, num_user, num_item, dim_C, aggr_mode, num_routing):
        super(CGCN, self).__init__()
        self.num_user, self.num_item, self.aggr_mode, self.num_routing, self.dim_C = num_user, num_item, aggr_mode, num_routing, dim_C
        self.preference = nn.Parameter(nn.init.xavier_normal_(torch.rand((num_user, dim_C))))
        self.dim_feat = features.size(1)
        self.register_buffer("features", features.clone(), persistent=False)
        self.MLP = nn.Linear(self.dim_feat, dim_C)

    def forward(self, owner, w):
        """-> (x + leaky_relu(A_alpha x), alpha of the user -> item edges, alpha of the item -> user edges), per distinct pair"""
        features = F.normalize(ops.linear(self.features, self.MLP.weight, self.MLP.bias, act=1))
        preference = F.normalize(self.preference)
        for _ in range(self.num_routing):
            preference = F.normalize(preference)           # (+ the attention layer's user rows: zero, see the module docstring)
        x = torch.cat((preference, features), dim=0)
        alpha_to_item, alpha_to_user, adj = owner.attention(x, w)
        return x + F.leaky_relu(sparse.mm(adj, x)), alpha_to_item, alpha_to_user


class GRCN(torch.nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, dim_C, reg_weight,
                 dropout, num_routing, aggr_mode, device, weight_mode='confid', fusion_mode='concat', pruning='True'):
        super(GRCN, self).__init__()
        if weight_mode != 'confid' or fusion_mode != 'concat' or aggr_mode != 'add':
            raise NotImplementedError("GRCN on the MI355X path: weight_mode='confid', fusion_mode='concat', aggr_mode='add' (the reference's defaults)")
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.weight_mode, self.fusion_mode, self.pruning = weight_mode, fusion_mode, pruning
        self.reg_weight, self.dropout, self.device = reg_weight, dropout, device
        self.register_buffer("weight", torch.tensor([[1.0], [-1.0]]), persistent=False)
        self._pairs = sparse.PairStructure(edge_index, num_user, num_item, device)      # distinct interactions + one symmetric [N, N] structure
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self.n_listed, self._lower, self._structure = self._pairs.n_listed, self._pairs.lower, self._pairs.structure

        self.id_gcn = EGCN(num_user, num_item, dim_E, aggr_mode)
        self.v_gcn = CGCN(v_feat, num_user, num_item, dim_C, aggr_mode, num_routing)
        self.t_gcn = CGCN(t_feat, num_user, num_item, dim_C, aggr_mode, num_routing)
        self.model_specific_conf = nn.Parameter(nn.init.xavier_normal_(torch.rand((num_user + num_item, 2))))
        self.result = nn.init.xavier_normal_(torch.rand((num_user + num_item, dim_E))).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.edge_keep_fn = None

    # ---- edges ----------------------------------------------------------------------------------------------------------------
    def _adj(self, to_user, to_item, w):
        """The [N, N] operand whose entry (u, U + i) is w * to_user (the edge item -> user) and (U + i, u) is w * to_item."""
        up, low = w * to_user, w * to_item
        return sparse.DroppedAdj(self._structure, torch.cat([up, low[self._lower]]), torch.cat([low, up[self._lower]]))

    def attention(self, x, w):
        """GATConv (:18-37) over the bidirectional list: -> (alpha of user -> item edges, alpha of item -> user edges, operand)."""
        U = self.num_user
        logit = ops.edge_dot(self._structure, x, x, self.n_edges)         # <x[u], x[U + i]> per pair: the structure's first half
        alpha_to_item = _segment_softmax(logit, self._ei, self.num_item, w)         # softmax over the edges arriving at an item
        alpha_to_user = _segment_softmax(logit, self._eu, self.num_user, w)
        return alpha_to_item, alpha_to_user, self._adj(alpha_to_user, alpha_to_item, w)

    def _kept_weights(self):
        """dropout_adj (:202): every LISTED edge kept with probability 1 - p; a pair's weight = its kept copies."""
        if self.edge_keep_fn is not None:
            keep = self.edge_keep_fn(self.n_listed, self.dropout).to(self._ew.device)
        else:
            keep = torch.rand(self.n_listed, device=self._ew.device) >= self.dropout
        return self._pairs.kept_copies(keep)

    # ---- :196-249 -------------------------------------------------------------------------------------------------------------
    def forward(self):
        U = self.num_user
        w = self._kept_weights()
        v_rep, v_to_item, v_to_user = self.v_gcn(self, w)
        t_rep, t_to_item, t_to_user = self.t_gcn(self, w)
        content_rep = torch.cat((v_rep, t_rep), dim=1)
        conf_u, conf_i = self.model_specific_conf[self._eu], self.model_specific_conf[U + self._ei]
        # the weight of an edge = max over the modalities of (its attention x the confidence of its SOURCE node), pruned
        to_item = torch.relu(torch.max(torch.stack((v_to_item, t_to_item), dim=1) * conf_u, dim=1)[0])
        to_user = torch.relu(torch.max(torch.stack((v_to_user, t_to_user), dim=1) * conf_i, dim=1)[0])
        id_rep = self.id_gcn(self._adj(to_user, to_item, w))
        self.result = torch.cat((id_rep, content_rep), dim=1)
        return self.result

    def loss(self, user_tensor, item_tensor):
        """:251-272."""
        user_tensor, item_tensor = user_tensor.view(-1).to(self.device), item_tensor.view(-1).to(self.device)
        out = self.forward()
        score = torch.sum(out[user_tensor] * out[item_tensor], dim=1).view(-1, 2)
        loss = -torch.mean(torch.log(torch.sigmoid(torch.matmul(score, self.weight))))
        reg_embedding_loss = (self.id_gcn.id_embedding[user_tensor] ** 2 + self.id_gcn.id_embedding[item_tensor] ** 2).mean()
        reg_content_loss = (self.v_gcn.preference[user_tensor] ** 2).mean() + (self.t_gcn.preference[user_tensor] ** 2).mean()
        return loss + self.reg_weight * (reg_embedding_loss + reg_content_loss)

    def gene_ranklist(self, step=200, topk=50, to_cpu=True):
        """:274-311: the table of the last forward, history at 1e-5 (this model's mask value)."""
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-5, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
