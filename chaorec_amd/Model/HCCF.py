"""HCCF with the reference's surface (Model/HCCF.py:17-227) -- per layer a LightGCN propagate over the (edge-dropped)
D^-1/2 A D^-1/2 plus a hypergraph channel  H (H^T x)  whose incidence H = mult * embeddings is learned, and a contrast of the two
channels' tables -- `torch.spmm` family: the propagate is `chaorec_amd.sparse.mm`, its per-step edge dropout
`sparse.sparse_dropout` (:57-70 is SelfCF's mask: floor(rand + keepRate), kept values / keepRate -- the structure stays, the
value array changes), the ranking `ranking.gene_ranklist` over the layer-summed table of the last forward (:203-227).  The
hypergraph channel is two [n, D] x [D, D] products (:52-55): dense library GEMMs.

Same constructor, parameters in the reference's creation order; `leaky`, `aggr_mode` and `hyperNum` are stored and unused, as
there."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse


class HCCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, aggr_mode,
                 ssl_alpha, ssl_temp, keepRate, leaky, mult, device):
        super(HCCF, self).__init__()
        self.result = None
        self.num_user, self.num_item, self.user_item_dict, self.dim_E = num_user, num_item, user_item_dict, dim_E
        self.edge_index, self.gnn_layer, self.aggr_mode, self.device = edge_index, n_layers, aggr_mode, device
        self.reg_weight, self.ssl_alpha, self.ssl_temp = reg_weight, ssl_alpha, ssl_temp
        self.hyperNum, self.leaky, self.keepRate, self.mult = 128, leaky, keepRate, mult
        self.uEmbeds = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_user, dim_E)))
        self.iEmbeds = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_item, dim_E)))
        e = torch.as_tensor(edge_index).long()
        self.adj = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.edge_keep_fn = None

    def gcn_layer(self, adj, embeds):
        return sparse.mm(adj, embeds)

    def hgnn_layer(self, adj, embeds):
        return adj @ (adj.T @ embeds)

    def sp_adj_drop_edge(self):
        """:57-70."""
        if self.keepRate == 1.0:
            return self.adj
        keep = self.edge_keep_fn(self.adj.nnz) if self.edge_keep_fn is not None else None
        return sparse.sparse_dropout(self.adj, 1.0 - self.keepRate, keep=keep)

    def forward(self):
        """:95-125."""
        embeds = torch.concat([self.uEmbeds, self.iEmbeds], dim=0)
        lats, gnnLats, hyperLats = [embeds], [embeds], [embeds]
        uuHyper, iiHyper = self.uEmbeds * self.mult, self.iEmbeds * self.mult
        for _ in range(self.gnn_layer):
            temEmbeds = self.gcn_layer(self.sp_adj_drop_edge(), lats[-1])
            hyperULat = self.hgnn_layer(F.dropout(uuHyper, p=1 - self.keepRate), lats[-1][:self.num_user])
            hyperILat = self.hgnn_layer(F.dropout(iiHyper, p=1 - self.keepRate), lats[-1][self.num_user:])
            gnnLats.append(temEmbeds)
            hyperLats.append(torch.concat([hyperULat, hyperILat], dim=0))
            lats.append(temEmbeds + hyperLats[-1])
        embeds = sum(lats)
        self.result = embeds
        return embeds, gnnLats, hyperLats

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        u, p, n = embeddings[users], embeddings[self.num_user + pos_items], embeddings[self.num_user + neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def ssl_loss(self, embeds1, embeds2, nodes):
        pck1, pck2 = F.normalize(embeds1 + 1e-8, p=2)[nodes], F.normalize(embeds2 + 1e-8, p=2)[nodes]
        nume = torch.exp(torch.sum(pck1 * pck2, dim=-1) / self.ssl_temp)
        deno = torch.exp(pck1 @ pck2.T / self.ssl_temp).sum(-1) + 1e-8
        return -torch.log(nume / deno).mean()

    def regularization_loss(self, users, pos_items, neg_items):
        u, p, n = self.result[users], self.result[self.num_user + pos_items], self.result[self.num_user + neg_items]
        return self.reg_weight * (torch.mean(u ** 2) + torch.mean(p ** 2) + torch.mean(n ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        embeds, gcnEmbedsLst, hyperEmbedsLst = self.forward()
        sslLoss = 0
        for i in range(self.gnn_layer):                      # (:192-197: layers 0 .. L - 1 of the two lists, layer 0 being the ego table of both)
            embeds1, embeds2 = gcnEmbedsLst[i].detach(), hyperEmbedsLst[i]
            sslLoss = sslLoss + self.ssl_loss(embeds1[:self.num_user], embeds2[:self.num_user], users) \
                + self.ssl_loss(embeds1[self.num_user:], embeds2[self.num_user:], pos_items)
        return self.bpr_loss(users, pos_items, neg_items, embeds) + self.ssl_alpha * sslLoss \
            + self.regularization_loss(users, pos_items, neg_items)

    def gene_ranklist(self, topk=50, to_cpu=True):
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
