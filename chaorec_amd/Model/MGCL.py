"""MGCL with the reference's surface (Model/MGCL.py:16-194) -- three LightGCN encoders over the same user-item graph (ids,
projected visual features, projected textual features; BasicGCN.GCNConv, the hot path's own propagate), a BPR + L2 term per
encoder and a cross-entropy contrast of the id view with the two modality views.  Each encoder is one
`ops.layer_mean_propagate` over `graph.lightgcn_csr`, the projections are `ops.linear` on the MFMA GEMM, the three BPR terms
the fused BPR kernel, the ranking `ranking.gene_ranklist` over the id encoder's table (:170-194).

Same constructor, parameters in the reference's creation order (the unused 0-dim `lambda_m` included)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking


class MGCL(torch.nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, reg_weight,
                 n_layers, aggr_mode, ssl_temp, ssl_alpha, device):
        super(MGCL, self).__init__()
        self.result = None
        self.num_user, self.num_item, self.user_item_dict, self.dim_E = num_user, num_item, user_item_dict, dim_E
        self.n_layers, self.ssl_temp, self.ssl_alpha, self.device = n_layers, ssl_temp, ssl_alpha, device
        self.reg_weight, self.aggr_mode = reg_weight, aggr_mode
        self.register_buffer("v_feat", v_feat.clone(), persistent=False)
        self.register_buffer("t_feat", t_feat.clone(), persistent=False)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.user_embedding_v = nn.Embedding(num_user, dim_E)
        nn.init.xavier_uniform_(self.user_embedding_v.weight)
        self.user_embedding_t = nn.Embedding(num_user, dim_E)
        nn.init.xavier_uniform_(self.user_embedding_t.weight)
        self.image_trs = nn.Linear(v_feat.shape[1], dim_E)
        self.text_trs = nn.Linear(t_feat.shape[1], dim_E)
        nn.init.xavier_uniform_(self.image_trs.weight)
        nn.init.xavier_uniform_(self.text_trs.weight)
        self.lambda_m = nn.Parameter(torch.tensor(0.1))
        self.graph = graph.lightgcn_csr(edge_index, num_user + num_item).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def _encode(self, users, items):
        mean = ops.layer_mean_propagate(torch.cat((users, items), dim=0), self.graph, self.n_layers)
        return mean, torch.split(mean, [self.num_user, self.num_item], dim=0)

    def forward(self):
        """:52-92."""
        v_embedding = ops.linear(self.v_feat, self.image_trs.weight, self.image_trs.bias)
        t_embedding = ops.linear(self.t_feat, self.text_trs.weight, self.text_trs.bias)
        self.result, (u_g, i_g) = self._encode(self.user_embedding.weight, self.item_embedding.weight)
        _, (u_v, i_v) = self._encode(self.user_embedding_v.weight, v_embedding)
        _, (u_t, i_t) = self._encode(self.user_embedding_t.weight, t_embedding)
        return u_g, i_g, u_v, i_v, u_t, i_t

    def bpr_loss(self, users, pos_items, neg_items, u_g, i_g):
        return ops.bpr_loss(u_g.contiguous(), i_g.contiguous(), users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items, u_g, i_g):
        return self.reg_weight * (torch.mean(u_g[users] ** 2) + torch.mean(i_g[pos_items] ** 2) + torch.mean(i_g[neg_items] ** 2))

    def cl_loss(self, id, emb, visual, textual):
        emb = F.normalize(emb[id], p=2, dim=1)
        visual, text = F.normalize(visual[id], p=2, dim=1), F.normalize(textual[id], p=2, dim=1)
        labels = torch.arange(emb.shape[0], device=emb.device)
        v_cl_loss = nn.CrossEntropyLoss()(torch.mm(emb, visual.T) / self.ssl_temp, labels)
        t_cl_loss = nn.CrossEntropyLoss()(torch.mm(emb, text.T) / self.ssl_temp, labels)
        return self.ssl_alpha * (v_cl_loss + t_cl_loss)

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        u_g, i_g, u_v, i_v, u_t, i_t = self.forward()
        views = ((u_g, i_g), (u_v, i_v), (u_t, i_t))
        bpr_loss = sum(self.bpr_loss(users, pos_items, neg_items, u, i) for u, i in views)
        reg_loss = sum(self.regularization_loss(users, pos_items, neg_items, u, i) for u, i in views)
        cl_loss = self.cl_loss(users, u_g, u_v, u_t) + self.cl_loss(pos_items, i_g, i_v, i_t)
        return bpr_loss + reg_loss + cl_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
