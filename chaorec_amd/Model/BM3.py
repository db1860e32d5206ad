"""BM3 with the reference's surface (Model/BM3.py:15-166) -- a bootstrap (BYOL-style) recommender: LightGCN over the user-item
graph (BasicGCN.GCNConv, the hot path's own propagate: `ops.layer_mean_propagate` over `graph.lightgcn_csr`), a shared
predictor Linear, dropout targets under no_grad, six cosine terms; no negative sample is read (the batch's third column is
ignored, :58).  The feature projections ([I, 4096] x [4096, D]) and the predictor are `ops.linear` on the MFMA GEMM; the
ranking scores PREDICTED tables (:142-145) through `ranking.gene_ranklist`.

Same constructor, parameters in the reference's creation order.  The four dropout targets draw on the device; `dropout_fn`
replays stored masks in the golden test."""
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.functional import cosine_similarity

from .. import graph, ops, ranking


class BM3(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, feat_E,
                 reg_weight, dropout, n_layers, cl_weight, aggr_mode, device):
        super(BM3, self).__init__()
        self.result, self.device = None, device
        self.num_user, self.num_item, self.aggr_mode, self.user_item_dict = num_user, num_item, aggr_mode, user_item_dict
        self.reg_weight, self.dim_E, self.feat_E, self.cl_weight = reg_weight, dim_E, feat_E, cl_weight
        self.dropout, self.n_layers = dropout, n_layers
        self.graph = graph.lightgcn_csr(edge_index, num_user + num_item).to(device)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.predictor = nn.Linear(dim_E, dim_E)
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        self.image_trs = nn.Linear(v_feat.shape[1], feat_E)
        nn.init.xavier_normal_(self.image_trs.weight)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        self.text_trs = nn.Linear(t_feat.shape[1], feat_E)
        nn.init.xavier_normal_(self.text_trs.weight)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.dropout_fn = None

    def forward(self):
        """:43-56."""
        h = self.item_embedding.weight
        ego = torch.cat((self.user_embedding.weight, h), dim=0)
        u_g, i_g = torch.split(ops.layer_mean_propagate(ego, self.graph, self.n_layers), [self.num_user, self.num_item], dim=0)
        i_g = i_g + h
        self.result = torch.cat((u_g, i_g), dim=0)
        return u_g, i_g

    def _predict(self, x):
        return ops.linear(x, self.predictor.weight, self.predictor.bias)

    def loss(self, users, items, _):
        """:58-108."""
        users, items = users.to(self.device), (items - self.num_user).to(self.device)
        u_online_ori, i_online_ori = self.forward()
        t_feat_online = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        v_feat_online = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        drop = self.dropout_fn if self.dropout_fn is not None else (lambda x, p: F.dropout(x, p))
        with torch.no_grad():
            u_target, i_target = drop(u_online_ori, self.dropout), drop(i_online_ori, self.dropout)
            t_feat_target, v_feat_target = drop(t_feat_online, self.dropout), drop(v_feat_online, self.dropout)
        u_online, i_online = self._predict(u_online_ori)[users, :], self._predict(i_online_ori)[items, :]
        u_target, i_target = u_target[users, :], i_target[items, :]
        t_feat_online, t_feat_target = self._predict(t_feat_online)[items, :], t_feat_target[items, :]
        loss_t = 1 - cosine_similarity(t_feat_online, i_target, dim=-1).mean()
        loss_tv = 1 - cosine_similarity(t_feat_online, t_feat_target, dim=-1).mean()
        v_feat_online, v_feat_target = self._predict(v_feat_online)[items, :], v_feat_target[items, :]
        loss_v = 1 - cosine_similarity(v_feat_online, i_target, dim=-1).mean()
        loss_vt = 1 - cosine_similarity(v_feat_online, v_feat_target, dim=-1).mean()
        loss_ui = 1 - cosine_similarity(u_online, i_target, dim=-1).mean()
        loss_iu = 1 - cosine_similarity(i_online, u_target, dim=-1).mean()
        reg_loss = self.reg_weight * (torch.mean(u_online_ori ** 2) + torch.mean(i_online_ori ** 2))
        return (loss_ui + loss_iu).mean() + reg_loss + self.cl_weight * (loss_t + loss_v + loss_tv + loss_vt).mean()

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:140-166: the PREDICTED tables of the last forward, history at 1e-6."""
        with torch.no_grad():
            res = self._predict(self.result.detach())
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
