"""SelfCF with the reference's surface (Model/SelfCF.py:14-238) -- `torch.sparse.mm` family, no per-model kernel work
(SURVEY 8(f).1): a LightGCN encoder whose propagate is `chaorec_amd.sparse.mm` over a per-step edge-dropped adjacency
(`sparse.sparse_dropout`: the family's helper, the structure and its SpMM schedule stay, only the value array changes), a
Linear predictor, negative cosine similarity between predicted and (stop-gradient, dropped-out) target views -- no
negative samples.  The ranking sums two score matrices, u_online i_target^T + u_target i_online^T (:213-238): ONE inner
product over the concatenated 2 D-wide rows, so it is the shared ranking kernel over [u_online | u_target] and
[i_target | i_online].

Same constructor, module tree and parameter names (`online_encoder.embedding_dict.{user_emb,item_emb}`,
`predictor.{weight,bias}`: same seed, same weights), `forward`, `get_embedding`, `loss_fn`, `loss`, `gene_ranklist`."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse


class L2Loss(nn.Module):
    def forward(self, *embeddings):
        total = torch.zeros(1, device=embeddings[-1].device)
        for e in embeddings:
            total = total + torch.sum(e ** 2) * 0.5
        return total


class LightGCN_Encoder(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, device):
        super(LightGCN_Encoder, self).__init__()
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers, self.device = dim_E, reg_weight, n_layers, device
        self.layers = [dim_E] * n_layers
        self.drop_ratio, self.drop_flag = 1.0, True                  # (:46-47)
        self.embedding_dict = nn.ParameterDict({
            "user_emb": nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_user, dim_E))),
            "item_emb": nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_item, dim_E)))})
        e = torch.as_tensor(edge_index).long()
        self.sparse_norm_adj = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)

    def sparse_dropout(self, x, rate, noise_shape=None):
        return sparse.sparse_dropout(x, rate)

    def _propagate(self, A_hat):
        ego = torch.cat([self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]], 0)
        layers = [ego]
        for _ in range(len(self.layers)):
            ego = sparse.mm(A_hat, ego)
            layers.append(ego)
        out = torch.mean(torch.stack(layers, dim=1), dim=1)
        return out[:self.num_user, :], out[self.num_user:, :]

    def forward(self, users, items):
        """:114-135: a fresh random dropout RATE per step (uniform in [0, drop_ratio)) on the adjacency's entries."""
        A_hat = (self.sparse_dropout(self.sparse_norm_adj, np.random.random() * self.drop_ratio) if self.drop_flag
                 else self.sparse_norm_adj)
        u, i = self._propagate(A_hat)
        return u[users, :], i[items, :]

    @torch.no_grad()
    def get_embedding(self):
        return self._propagate(self.sparse_norm_adj)


class SelfCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, dropout, device):
        super(SelfCF, self).__init__()
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers, self.device, self.dropout = dim_E, reg_weight, n_layers, device, dropout
        self.reg_loss = L2Loss()
        self.online_encoder = LightGCN_Encoder(num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers,
                                               device)
        self.predictor = nn.Linear(dim_E, dim_E)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def forward(self, users, items):
        u_online, i_online = self.online_encoder(users, items)
        with torch.no_grad():
            u_target = F.dropout(u_online.clone(), self.dropout)
            i_target = F.dropout(i_online.clone(), self.dropout)
        return u_online, u_target, i_online, i_target

    @torch.no_grad()
    def get_embedding(self):
        u_online, i_online = self.online_encoder.get_embedding()
        return self.predictor(u_online), u_online, self.predictor(i_online), i_online

    def loss_fn(self, p, z):
        return -F.cosine_similarity(p, z.detach(), dim=-1).mean()

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        u_online, u_target, i_online, i_target = self.forward(users, pos_items)
        reg_loss = self.reg_weight * self.reg_loss(u_online, i_online)
        u_online, i_online = self.predictor(u_online), self.predictor(i_online)
        return self.loss_fn(u_online, i_target) / 2 + self.loss_fn(i_online, u_target) / 2 + reg_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        u_online, u_target, i_online, i_target = self.get_embedding()
        res = torch.cat((torch.cat((u_online, u_target), 1), torch.cat((i_target, i_online), 1)), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
